#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations/sec (whole job) on BASELINE.json config #2 --
CartPole-v1, fully-connected net (the reference's trained checkpoint), 4096 envs x 50 sims per GPU.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one move of self-play search for every env on the rank: root inference + root expansion
with Dirichlet noise, 50 x (select -> recurrent_inference -> expand/backup), readout, action sampling
(synthetic fixed-weight rollouts, observations uniform(-0.05, 0.05) resident in HBM).
value = (ranks x envs x sims x steps) / max-over-ranks wall time; weak scaling (fixed envs per GPU);
each rank owns envs [rank*E, (rank+1)*E) with RNG seeds config.seed + global env index, and every
`--bcast-every` steps all ranks take rank 0's flat weight buffer by RCCL broadcast.

Prints ONE JSON line (rank 0).  Extra legs on rank 0:
  roofline      HIP-event-bracketed eager pass of the same workload (events cannot bracket kernels
                inside a replayed hipGraph): per-kernel mean duration, algorithmic bytes per launch
                (SURVEY.md section 8d formula with the measured mean select depth) -> achieved GB/s vs 8 TB/s
  cpu_baseline  N=1 only: the C oracle (a port of the reference's one-tree-at-a-time loop) on one host
                core over a bounded sample of the same observations
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
# oracle-port / reference ratio measured in the build container on the same inputs (1 thread, Xeon
# 2.1 GHz): C port 324e3 sims/s vs reference Python 1559 sims/s (tests/golden/g10_reference_speed.npz)
RHO_PORT_OVER_REFERENCE = 208.0


def pkg(sub):
    return importlib.import_module(f"muzero-hypermodel_amd.{sub}")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs", type=int, default=4096, help="envs (trees) per GPU")
    ap.add_argument("--bcast-every", type=int, default=50, help="weight broadcast period in steps (N>1)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--profile-steps", type=int, default=10, help="steps of the HIP-event pass (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--obs-sets", type=int, default=8)
    ap.add_argument("--mode", choices=["fused", "lockstep"], default="fused",
                    help="fused: whole move in one HIP launch (FC net in-kernel, trees in LDS); "
                         "lockstep: select -> PyTorch-ROCm inference -> expand_backup per simulation")
    ap.add_argument("--group", type=int, default=0, help="lanes per tree (0 = default for the mode)")
    ap.add_argument("--hidden-in-hbm", action="store_true", help="fused mode: keep hidden states out of LDS")
    ap.add_argument("--groups", type=int, default=2,
                    help="fused mode with --moves-per-batch 0: env groups per GPU on separate HIP streams")
    ap.add_argument("--moves-per-batch", type=int, default=50,
                    help="fused mode: moves queued back to back per host round trip (mzmcts_moves_*); "
                         "0 = one host round trip per move, pipelined over --groups env groups")
    return ap.parse_args()


def main():
    args = parse_args()
    from parity_helpers import load_golden
    actor_mod, cartpole, engine_mod = pkg("actor"), pkg("games.cartpole"), pkg("engine")
    rank, world, local_rank = actor_mod.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    config = cartpole.MuZeroConfig()
    E, S, A = args.envs, config.num_simulations, len(config.action_space)
    w = load_golden("cartpole_weights")
    weights = {k: torch.from_numpy(w[k]) for k in w.files}
    fused = args.mode == "fused"
    group = args.group if args.group else (16 if fused else 0)
    actor = actor_mod.SearchActor(config, weights, E, rank=rank, device=device, use_graph=not args.no_graph,
                                  group_width=group, fused_fc=fused)
    engine, model = actor.engine, actor.model
    engine.fused_hidden_in_lds = not args.hidden_in_hbm

    rs = np.random.RandomState(123 + rank)
    obs_sets = [torch.from_numpy(rs.uniform(-0.05, 0.05, (E, 1, 1, 4)).astype(np.float32)).to(device)
                for _ in range(args.obs_sets)]
    legal = np.tile(np.arange(A, dtype=np.int32), (E, 1))
    num_legal = np.full(E, A, dtype=np.int32)
    to_play = np.zeros(E, dtype=np.int32)
    temperature = np.ones(E, dtype=np.float64)

    def one_step(i):
        if world > 1 and args.bcast_every and i % args.bcast_every == 0:
            actor.refresh_weights(src=0)
        engine.search(model, obs_sets[i % len(obs_sets)], legal, to_play, True, num_legal=num_legal)
        engine.sample_actions(temperature)

    def search_only_step(i):
        """One move on the single full-size engine, no collectives (roofline leg, rank 0 only)."""
        engine.search(model, obs_sets[i % len(obs_sets)], legal, to_play, True, num_legal=num_legal)
        engine.sample_actions(temperature)

    pipe = None
    batched = fused and args.moves_per_batch > 0
    if batched:
        engine.set_fused_options("auto", publish_tree=False)     # MCTS.run's callers consume the root only
        flat_obs = [o.reshape(E, -1).contiguous() for o in obs_sets]

        def batch_sizes(start, count):
            i, out = start, []
            while count > 0:
                b = min(args.moves_per_batch, count)
                if world > 1 and args.bcast_every:
                    b = min(b, args.bcast_every - i % args.bcast_every)
                out.append((i, b))
                i += b
                count -= b
            return out

        def run_plan(plan, trailing_predraw):
            """Batches plan[i] = (first step, moves); the first one must already be drawn and uploaded.  The host
            draws a batch's exploration noise while the previous batch runs and collects actions / visit counts /
            root values of a batch when it is done; kernels of one batch follow each other without host round
            trips.  trailing_predraw = (moves) draws one more batch during the last one (left uploaded-ready)."""
            trace = os.environ.get("MZ_BENCH_TRACE")
            played = 0
            for n, (i, b) in enumerate(plan):
                if world > 1 and args.bcast_every and i % args.bcast_every == 0:
                    actor.refresh_weights(src=0)
                t = [time.perf_counter()]
                for k in range(b):
                    engine.moves_enqueue(flat_obs[(i + k) % len(flat_obs)])
                t.append(time.perf_counter())
                nxt = plan[n + 1][1] if n + 1 < len(plan) else trailing_predraw
                if nxt:
                    engine.moves_predraw_next(nxt, legal, to_play, temperature, True, num_legal=num_legal)
                t.append(time.perf_counter())
                out = engine.moves_collect(copy=bool(os.environ.get("MZ_BENCH_COPY")))   # default: views of the download ring
                played += int(out["moves_done"].sum())   # an env whose stream left the pre-drawn path sits out the rest
                t.append(time.perf_counter())
                if n + 1 < len(plan):
                    engine.moves_submit_next()
                t.append(time.perf_counter())
                if trace:
                    print(f"[bench] batch of {b}: enqueue {1e3 * (t[1] - t[0]):.2f} ms, predraw {1e3 * (t[2] - t[1]):.2f}, "
                          f"collect {1e3 * (t[3] - t[2]):.2f}, submit {1e3 * (t[4] - t[3]):.2f}; moves done min "
                          f"{int(out['moves_done'].min())}", file=sys.stderr, flush=True)
            return played
    elif fused and args.groups > 1:
        # n env groups on n streams: one group's host work overlaps the other groups' kernels
        n = args.groups
        pipe = engine_mod.PipelinedSearch(config, E, model, actor.flat, groups=n, device=device,
                                          seeds=actor_mod.shard_seeds(config.seed, rank, E), group_width=group)
        per = E // n
        g_obs = [[o.reshape(E, -1)[pipe.slice(g)].contiguous() for g in range(n)] for o in obs_sets]
        g_legal, g_nl, g_tp, g_temp = legal[:per], num_legal[:per], to_play[:per], temperature[:per]
        started = [False] * n

        def one_step(i):  # noqa: F811 -- one move for every group, pipelined across groups and steps
            if world > 1 and args.bcast_every and i % args.bcast_every == 0:
                torch.cuda.synchronize(device)          # weights are shared by all groups' kernels
                actor.refresh_weights(src=0)
            for g in range(n):
                if started[g]:
                    pipe.finish(g)
                    pipe.engines[g].sample_actions(g_temp)
                pipe.begin(g, g_obs[i % len(obs_sets)][g], g_legal, g_tp, True, num_legal=g_nl)
                started[g] = True

        def drain():
            for g in range(n):
                if started[g]:
                    pipe.finish(g)
                    pipe.engines[g].sample_actions(g_temp)
                    started[g] = False

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    if batched:
        warm, timed = batch_sizes(0, args.warmup), batch_sizes(args.warmup, args.steps)
        engine.moves_prepare(max(b for _, b in warm + timed), legal, to_play, temperature, True, num_legal=num_legal)
        engine.moves_collect()                           # (sizes the batch buffers once; nothing was queued)
        if warm:
            engine.moves_prepare(warm[0][1], legal, to_play, temperature, True, num_legal=num_legal)
            run_plan(warm, timed[0][1])                  # steady state: the first timed batch is drawn during warm-up
    else:
        for i in range(args.warmup):
            one_step(i)
    if pipe:
        drain()
    barrier()
    t0 = time.perf_counter()
    if batched:
        # every timed batch is uploaded, run and collected inside the timed region, and so is one noise draw per
        # batch (the last batch's draw produces rows nobody runs: it stands in for the first batch's, done above)
        if warm:
            engine.moves_submit_next()
        else:
            engine.moves_prepare(timed[0][1], legal, to_play, temperature, True, num_legal=num_legal)
        moves_played = run_plan(timed, timed[-1][1])
    else:
        for i in range(args.steps):
            one_step(i)
        moves_played = E * args.steps
    if pipe:
        drain()                                          # every queued move is finished inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if batched:
        engine.moves_discard_next()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    if world > 1:
        t = torch.tensor([moves_played], dtype=torch.int64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        moves_played = int(t.item())
    # only searches that really ran count: in move batches an env whose tie-break words differ from the pre-drawn
    # assumption sits out the rest of its batch, so moves_played <= world * E * steps
    sims_total = moves_played * S
    value = sims_total / elapsed

    result = {
        "metric": "mcts_simulations_per_sec", "value": value, "unit": "simulations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"cartpole_fc_{E}envs_x_{S}sims", "envs_per_gpu": E, "simulations": S,
                   "actions": A,
                   "network": "fullyconnected (reference checkpoint), fp32 inference "
                              + ("in the fused HIP kernel" if fused else "through PyTorch-ROCm"),
                   "mode": args.mode, "lanes_per_tree": engine.group_width(),
                   "env_groups_per_gpu": args.groups if (fused and not batched and args.groups > 1) else 1,
                   "moves_per_host_round_trip": args.moves_per_batch if batched else 1,
                   "fused_kernel": engine.fused_variant() if fused else None,
                   "launch": "one kernel per move" if fused else ("eager" if args.no_graph else "hipgraph"),
                   "parallelism": f"actors{world}",
                   "weight_broadcast_every_steps": args.bcast_every if world > 1 else None},
        "self_play_moves_per_sec": moves_played / elapsed,
        "moves_played": moves_played, "moves_scheduled": world * E * args.steps,
    }

    if rank == 0:
        if args.profile_steps > 0:
            result["roofline"], result["kernels"] = roofline_leg(engine, search_only_step, args.profile_steps, device)
        if world == 1 and args.cpu_seconds > 0:
            result["cpu_baseline"] = cpu_baseline_leg(config, w, args.cpu_seconds)
            result["speedup_vs_reference_equivalent"] = value / result["cpu_baseline"]["reference_equivalent_value"]
    barrier()
    if rank == 0:
        print(json.dumps(result))
    actor.close()
    if world > 1:
        torch.distributed.destroy_process_group()


def roofline_leg(engine, one_step, steps, device):
    """Eager pass with HIP events around every tree-kernel launch (same stream as the launches)."""
    engine.set_profiling(True)
    engine.get_profile(reset=True)
    for i in range(2):
        one_step(i)
    torch.cuda.synchronize(device)
    engine.get_profile(reset=True)
    for i in range(steps):
        one_step(i)
    torch.cuda.synchronize(device)
    prof = engine.get_profile(reset=True)
    engine.set_profiling(False)
    sims = max(prof["simulations"], 1)
    mean_depth = prof["select_depth_sum"] / sims
    bytes_sim = engine.algorithmic_bytes_per_simulation(mean_depth)
    kernels = {}
    per_kernel = [("select", "select_ms", "select_launches", bytes_sim["select"] * engine.E),
                  ("expand_backup", "expand_backup_ms", "expand_backup_launches", bytes_sim["expand_backup"] * engine.E),
                  # one fused launch = S simulations of every tree: select + expand/backup bytes of all of them
                  ("search_fused_fc", "fused_ms", "fused_launches", bytes_sim["total"] * engine.E * engine.S)]
    for name, ms_key, n_key, per_launch in per_kernel:
        if prof[n_key] == 0:
            continue
        avg_us = 1e3 * prof[ms_key] / prof[n_key]
        kernels[name] = {"avg_us": avg_us, "launches": prof[n_key], "algorithmic_bytes_per_launch": per_launch,
                         "achieved_GBs": per_launch / (avg_us * 1e-6) / 1e9 if avg_us > 0 else None}
    dominant = max(kernels, key=lambda k: kernels[k]["avg_us"])
    d = kernels[dominant]
    kernel_name = f"mz::{dominant}_kernel"
    if dominant == "search_fused_fc" and engine.fused_variant() == "narrow":
        kernel_name = "mz::search_fused_narrow_kernel"
    traffic, traffic_source = pmc_traffic(kernel_name, engine.E)
    roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": d["achieved_GBs"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["achieved_GBs"] / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_source,
                "avg_kernel_us": d["avg_us"], "mean_select_depth": mean_depth,
                "algorithmic_bytes_per_simulation": bytes_sim,
                "timing": "HIP events bound to the kernel dispatch (hipExtLaunchKernel) on the launch stream, "
                          "over a second pass of the same steps",
                "working_set_bytes": engine.device_bytes()}
    return roofline, kernels


def pmc_traffic(kernel, envs):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE need separate runs and cannot be collected from inside this process).  Raw counter
    bytes, uncorrected -- see DESIGN.md section 5."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_e{envs}.json")))
    if not files:
        return None, None
    data = json.load(open(files[-1]))["counters"]
    total = 0.0
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        hit = [v for k, v in data.get(counter, {}).items() if kernel in k]
        if not hit:
            return None, None
        total += hit[0]["mean_KB_per_launch"] * 1024.0
    return total, os.path.relpath(files[-1], ROOT)


def cpu_baseline_leg(config, w, seconds):
    """The oracle's C port of the reference loop, one host core, bounded sample of the same workload."""
    import mz_oracle
    weights = {k: w[k] for k in w.files}
    net = mz_oracle.FcNet.from_config(config, weights)
    cfg = mz_oracle.config_from_muzero(config, H=config.encoding_size)
    obs = np.random.RandomState(123).uniform(-0.05, 0.05, (4096, 1, 1, 4)).astype(np.float32)
    rng = mz_oracle.Rng(config.seed)
    mz_oracle.fc_selfplay_moves(cfg, net, rng, obs[:64], 1.0)  # warm
    t0 = time.perf_counter()
    sims, moves = 0, 0
    while time.perf_counter() - t0 < seconds:
        out = mz_oracle.fc_selfplay_moves(cfg, net, rng, obs[:1024], 1.0)
        sims += out["sims"]
        moves += 1024
    dt = time.perf_counter() - t0
    try:
        cpu_model = [line.split(":", 1)[1].strip() for line in open("/proc/cpuinfo") if line.startswith("model name")][0]
    except Exception:
        cpu_model = "unknown"
    return {"value": sims / dt, "unit": "simulations/s", "cores": 1, "kind": "port",
            "sample": f"{moves} moves x {config.num_simulations} sims ({dt:.1f} s), same weights and "
                      "observation distribution, one tree at a time",
            "cpu": cpu_model, "host_cores_available": os.cpu_count(),
            "rho_port_over_reference": RHO_PORT_OVER_REFERENCE,
            "reference_equivalent_value": sims / dt / RHO_PORT_OVER_REFERENCE}


if __name__ == "__main__":
    main()

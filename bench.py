#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations/sec (whole job) of the self-play search path.

    python bench.py [--gpus N] [--workload cartpole|tictactoe|connect4|atari84] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`; default = config #2, the one the metric is quoted on):
  cartpole    #2  CartPole-v1, fully-connected net (the reference's trained checkpoint), 4096 envs x 50 sims per GPU,
                  whole move in ONE HIP launch (network in-kernel, trees in LDS), moves queued in batches
  tictactoe   #3  residual net 1 block x 16 channels, 25 sims, masked roots, 4096 envs per GPU, lock-step
  connect4    #4  residual net 3 blocks x 64 channels, 200 sims, 1024 envs per GPU, lock-step
  atari84     #5  84x84x4 frames -> CNN down-sampler -> 2 blocks x 16 channels at 6x6, 50 sims, 1024 envs per GPU
(lock-step = per simulation: select kernel -> recurrent inference through PyTorch-ROCm + HIP network kernels ->
expand/backup kernel; the S-simulation loop replayed from one hipGraph.  Synthetic weights for the residual nets.)

One "step" = one move of self-play search for every env on the rank: root inference + root expansion with Dirichlet
noise, S x (select -> recurrent_inference -> expand/backup), readout, action sampling (synthetic
fixed-weight rollouts, observations resident in HBM).  value = simulations of all ranks / max-over-ranks wall time;
weak scaling (fixed envs per GPU); rank g owns envs [g*E, (g+1)*E) with RNG seeds config.seed + global env index, and
every `--bcast-every` steps all ranks take rank 0's flat weight buffer by one RCCL broadcast.

Launch: with N > 1 and no WORLD_SIZE in the environment this script starts its N ranks itself (fresh child
processes, decided before anything touches the GPU) and relays rank 0's JSON line; under torch.distributed.run it
joins the group it is given.  The timed region is K steps repeated until it lasts >= --min-seconds (a 20-step CartPole
run would otherwise be a 5 ms sample): `steps` stays K, `timed_steps` says how many were timed, `ms_per_step` is
per step.

Prints ONE JSON line (rank 0).  Extra legs on rank 0:
  roofline      HIP-event pass of the same workload: per-kernel mean duration, algorithmic bytes per launch (SURVEY.md
                section 8d formula with the measured mean select depth) vs 8 TB/s for the tree kernels, network FLOPs vs
                the fp32 matrix peak for the inference launches; the dominant one is `roofline`, all are in `kernels`
  cpu_baseline  N = 1 only: the oracle's port of the reference's one-tree-at-a-time loop on one host core and on all
                available host cores (one process per core), bounded samples of the same workload
"""
import argparse
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_MATRIX_PEAK_TFLOPS = 157.3  # v_mfma_f32_* / fp32 VALU peak (same guide)
F16_MATRIX_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA peak (same guide; the 2:1-sparsity figure is never used)
# rho = oracle port / ACTUAL reference (imported from /root/reference), same inputs, measured in the build container with 1
# process and with one process per core (tests/golden/time_reference.py -> tests/golden/g10_reference_speed_configs.json:
# Xeon 2.1 GHz, 8 vCPU).  The port's measurement on the GPU box's host divided by rho is a LABELLED extrapolation of what
# the reference's own Python would do there; `cpu_baseline.value` is the port itself.
def _load_rho():
    try:
        data = json.load(open(os.path.join(ROOT, "tests", "golden", "g10_reference_speed_configs.json")))
    except (OSError, ValueError):
        return {}, None
    return data.get("configs", {}), {k: data.get(k) for k in ("cpu", "cores", "procs", "seconds_per_leg")}


REFERENCE_TIMING, REFERENCE_TIMING_HOST = _load_rho()

# Envs (trees) per GPU and engines they are split into.  BASELINE.json fixes the env count for config #2 only (4096
# CartPole envs); for the lock-step configs it is the actor's choice, and one MI355X (288 GB) is best used with many more
# trees than 4096: the per-simulation launches (tower, heads, select, gather, expand_backup) are filled better and their
# fixed costs amortised -- TicTacToe 37 M simulations/s at 4096 envs, 63 M at 16384, 75 M at 65536; Connect4 4.0 / 4.2 /
# 4.5 M at 1024 / 2048 / 4096; the 84x84 config 9.9 / 17.0 / 20.6 M at 1024 / 4096 / 16384 (first measurements of round
# 2, one engine).
# groups: the envs of a GPU as that many engines on streams of their own (engine.PipelinedLockstep): one group's host work
# -- readout, action sampling, the next move's checks and uploads -- runs under the other groups' kernels, and the kernels
# of two streams fill each other's gaps: TicTacToe 65536 envs 100 -> 138 M simulations/s with two groups, Connect4 8192
# envs 4.94 -> 5.67 M, the 84x84 config 32768 envs 26.6 -> 27.9 M.
WORKLOADS = {
    "cartpole": dict(envs=4096, baseline_config=2, groups=2),
    # fused_step: expand_backup of a simulation and the descent of the next in one launch (MZ_FUSED_STEP, engine.py) --
    # measured on one box, three alternating pairs: 239.6 / 244.2 / 244.5 M without, 245.3 / 250.8 / 253.4 M with it;
    # config #5 the same either way, Connect4 6.82 M without against 6.71 M with it: on for TicTacToe only
    "tictactoe": dict(envs=65536, baseline_config=3, groups=2, fused_step=True),
    "connect4": dict(envs=8192, baseline_config=4, groups=2),
    # (one group: a move is 42 ms of GPU work, the host's turn between moves is noise, and two half-size groups run
    #  every kernel at half its batch -- measured 38.9 M against 36.9 M simulations/s; TicTacToe needs its two groups
    #  to cover the host: 243 M against 160 M, and Connect4 gains 3 % from them)
    "atari84": dict(envs=32768, baseline_config=5, groups=1),
}


def pkg(sub):
    return importlib.import_module(f"muzero-hypermodel_amd.{sub}")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cartpole")
    ap.add_argument("--envs", type=int, default=0, help="envs (trees) per GPU (0 = the workload's default)")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="the K timed steps are repeated until the timed region lasts at least this long")
    ap.add_argument("--bcast-every", type=int, default=50, help="weight broadcast period in steps (N>1)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--host-noise", action="store_true",
                    help="lock-step workloads: draw the exploration noise on the host mirrors of the RNG streams instead "
                         "of on the GPU (the same rows to the last bit; 5 % slower at 65536 TicTacToe envs)")
    ap.add_argument("--profile-steps", type=int, default=10, help="steps of the HIP-event pass (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="budget of each cpu_baseline leg, 1 core and all cores (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="processes of the all-cores leg (0 = every available core)")
    ap.add_argument("--obs-sets", type=int, default=8)
    ap.add_argument("--mode", choices=["fused", "lockstep"], default="fused",
                    help="cartpole only.  fused: whole move in one HIP launch (FC net in-kernel, trees in LDS); "
                         "lockstep: select -> PyTorch-ROCm inference -> expand_backup per simulation")
    ap.add_argument("--group", type=int, default=0, help="lanes per tree (0 = default for the mode)")
    ap.add_argument("--hidden-in-hbm", action="store_true", help="fused mode: keep hidden states out of LDS")
    ap.add_argument("--groups", type=int, default=0,
                    help="env groups per GPU on separate HIP streams, one group's host work under the others' kernels: "
                         "lock-step workloads, and fused mode with --moves-per-batch 0 (0 = the workload's default)")
    ap.add_argument("--moves-per-batch", type=int, default=50,
                    help="fused mode: moves queued back to back per host round trip (mzmcts_moves_*); "
                         "0 = one host round trip per move, pipelined over --groups env groups")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU: run launch, rendezvous (gloo), weight broadcast, timing reduction and the JSON relay "
                         "with the search replaced by a sleep -- a test of the N>1 plumbing, never a measurement")
    args = ap.parse_args(argv)
    if args.groups <= 0:
        args.groups = WORKLOADS[args.workload]["groups"]
    if WORKLOADS[args.workload].get("fused_step"):
        os.environ.setdefault("MZ_FUSED_STEP", "on")     # (read by every engine this process creates)
    return args


# ----------------------------------------------------------------------------------------------------------------
# self-launch (N > 1 without torch.distributed.run)
# ----------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    """Start `args.gpus` fresh rank processes of this script and relay rank 0's JSON line.  Runs before anything in
    this process has touched the GPU (torch.cuda.device_count() does not initialise it on this image); children
    are new processes, never an exec of one that holds the GPU.  Mirrors muzero.py:170-186 (N self-play workers with
    seeds config.seed + i), without Ray."""
    n = args.gpus
    env = dict(os.environ)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MZ_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not args.rehearse_cpu:
        visible = torch.cuda.device_count()
        if visible == 0:
            print("bench.py needs MI355X GPUs (none visible); --rehearse-cpu exercises the N>1 plumbing on CPU",
                  file=sys.stderr)
            return 2
        if visible < n:
            # fewer GPUs than ranks: every rank shares cuda:0 and the group runs over gloo (RCCL refuses two ranks
            # on one device).  The JSON line says so ("rehearsal"); it is never a measurement.
            env["MZ_REHEARSE_ON_ONE_GPU"] = "1"
    procs = []
    for rank in range(n):
        child_env = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=child_env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr, text=True))
    # rank 0's stdout is drained by a thread: a rank that hangs or dies at the rendezvous keeps the others (and rank 0's
    # pipe) open for ever, and a blocking read here would never reach the deadline below
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("MZ_BENCH_LAUNCH_DEADLINE_S", "3400"))
    codes = [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = any(c not in (None, 0) for c in codes)
        if failed or time.time() > deadline:
            # the first rank to fail (or the deadline) ends the job: the others would wait in a collective for ever
            grace = time.time() + (5.0 if failed else 0.0)
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=max(0.0, grace - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        p.wait()
                        codes[i] = -9
            break
        time.sleep(0.05)
    reader.join(timeout=5.0)
    line = "".join(chunks)
    for row in line.splitlines():                           # ONE JSON line on stdout; anything else a rank printed -> stderr
        print(row, file=sys.stdout if row.startswith("{") else sys.stderr, flush=True)
    bad = [c for c in codes if c != 0]
    return bad[0] if bad else 0


# ----------------------------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------------------------
def workload_config(name):
    if name == "atari84":
        return pkg("games.breakout").atari84_config()
    return pkg(f"games.{name}").MuZeroConfig()


def workload_weights(name, config):
    from parity_helpers import load_golden
    if name == "cartpole":
        w = load_golden("cartpole_weights")
        return {k: torch.from_numpy(w[k]) for k in w.files}, "reference checkpoint"
    from synth import synthetic_state_dict
    template = pkg("models").MuZeroNetwork(config).state_dict()
    return ({k: torch.from_numpy(v) for k, v in synthetic_state_dict(template, 0).items()},
            "synthetic (tests/golden/synth.py seed 0)")


def synthetic_positions(name, config, E, rs, device, sets):
    """Resident observation batches + root legal sets: CartPole uniform(-0.05, 0.05); board
    games random planes with at least one illegal root action per env; Atari-like uniform(0, 1) frames."""
    A, P = len(config.action_space), len(config.players)
    C, H, W = config.observation_shape
    obs = []
    for _ in range(sets):
        if name == "cartpole":
            o = rs.uniform(-0.05, 0.05, (E, C, H, W)).astype(np.float32)
        elif name == "atari84":
            o = rs.uniform(0, 1, (E, C, H, W)).astype(np.float32)
        else:
            o = rs.randint(0, 2, (E, C, H, W)).astype(np.float32)
            o[:, 2] = 1.0
        obs.append(torch.from_numpy(o).to(device))
    legal = np.zeros((E, A), np.int32)
    num_legal = np.zeros(E, np.int32)
    if P == 1:
        legal[:] = np.arange(A, dtype=np.int32)
        num_legal[:] = A
    else:
        for e in range(E):
            k = int(rs.randint(max(1, A // 2), A))
            legal[e, :k] = np.sort(rs.choice(A, size=k, replace=False))
            num_legal[e] = k
    to_play = rs.randint(0, P, E).astype(np.int32) if P > 1 else np.zeros(E, np.int32)
    return obs, legal, num_legal, to_play


def _conv_flops(c_in, c_out, k, h, w):
    return 2 * c_in * c_out * k * k * h * w


def _mlp_flops(sizes):
    return sum(2 * a * b for a, b in zip(sizes[:-1], sizes[1:]))


def recurrent_inference_flops(config):
    """FLOPs (2 per multiply-add) of one recurrent_inference of one sample (reference models.py:399-522, 128-195);
    matches SURVEY.md section 8's table, which was measured with torch's flop counter."""
    A, F = len(config.action_space), 2 * config.support_size + 1
    if config.network == "fullyconnected":
        enc = config.encoding_size
        return (_mlp_flops([enc + A] + list(config.fc_dynamics_layers) + [enc])
                + _mlp_flops([enc] + list(config.fc_reward_layers) + [F])
                + _mlp_flops([enc] + list(config.fc_value_layers) + [F])
                + _mlp_flops([enc] + list(config.fc_policy_layers) + [A]))
    _, h, w = config.observation_shape
    if config.downsample:
        h, w = math.ceil(h / 16), math.ceil(w / 16)
    c, b = config.channels, config.blocks
    plane = h * w
    flops = _conv_flops(c + 1, c, 3, h, w) + b * 2 * _conv_flops(c, c, 3, h, w)          # dynamics tower
    flops += b * 2 * _conv_flops(c, c, 3, h, w)                                          # prediction tower
    for reduced, layers, out in ((config.reduced_channels_reward, config.resnet_fc_reward_layers, F),
                                 (config.reduced_channels_value, config.resnet_fc_value_layers, F),
                                 (config.reduced_channels_policy, config.resnet_fc_policy_layers, A)):
        flops += _conv_flops(c, reduced, 1, h, w) + _mlp_flops([reduced * plane] + list(layers) + [out])
    return flops


class Workload:
    """One rank's shard of a benchmark workload: model replica, engine, resident synthetic inputs."""

    def __init__(self, args, rank, world, device):
        self.args, self.rank, self.world, self.device = args, rank, world, device
        self.name = args.workload
        self.config = workload_config(self.name)
        self.E = args.envs or WORKLOADS[self.name]["envs"]
        self.S, self.A = self.config.num_simulations, len(self.config.action_space)
        weights, self.weights_kind = workload_weights(self.name, self.config)
        self.fused = self.name == "cartpole" and args.mode == "fused"
        group = args.group if args.group else (16 if self.fused else 0)
        actor_mod = pkg("actor")
        self.actor = actor_mod.SearchActor(self.config, weights, self.E, rank=rank, device=device,
                                           use_graph=not args.no_graph, group_width=group, fused_fc=self.fused,
                                           device_noise=not self.fused and not args.host_noise)
        self.engine, self.model = self.actor.engine, self.actor.model
        self.engine.fused_hidden_in_lds = not args.hidden_in_hbm
        rs = np.random.RandomState(123 + rank)
        self.obs_sets, self.legal, self.num_legal, self.to_play = synthetic_positions(
            self.name, self.config, self.E, rs, device, args.obs_sets)
        self.temperature = np.ones(self.E, dtype=np.float64)
        self.batched = self.fused and args.moves_per_batch > 0
        self.pipe = None
        self.bcast = world > 1 and args.bcast_every > 0
        self.refreshes = 0
        if self.batched:
            self.engine.set_fused_options("auto", publish_tree=False)     # MCTS.run's callers consume the root only
            self.flat_obs = [o.reshape(self.E, -1).contiguous() for o in self.obs_sets]
        elif self.fused and args.groups > 1:
            self._init_pipeline(actor_mod, group)
        elif not self.fused and args.groups > 1:
            self._init_lockstep_pipeline(actor_mod)

    # ---- weight refresh (the one exchange step of the path) ---------------------------------------------------------
    def refresh(self):
        self.actor.refresh_weights(src=0)
        if self.pipe is not None and hasattr(self.pipe, "refresh"):
            self.pipe.refresh()                              # the other groups' network replicas follow
        self.refreshes += 1

    # ---- one move for every env, one host round trip per move -------------------------------------------------------
    def search_only_step(self, i):
        """One move on the full-size engine, no collectives (also the roofline leg's step)."""
        e = self.engine
        e.search(self.model, self.obs_sets[i % len(self.obs_sets)], self.legal, self.to_play, True,
                 num_legal=self.num_legal)
        e.sample_actions(self.temperature)

    def one_step(self, i):
        if self.bcast and i % self.args.bcast_every == 0:
            if self.pipe:
                torch.cuda.synchronize(self.device)          # weights are shared by all groups' kernels
            self.refresh()
        if self.pipe:
            self._pipe_step(i)
        else:
            self.search_only_step(i)

    # ---- fused CartPole, several env groups on separate streams -----------------------------------------------------
    def _init_pipeline(self, actor_mod, group):
        n = self.args.groups
        self.pipe = pkg("engine").PipelinedSearch(self.config, self.E, self.model, self.actor.flat, groups=n,
                                                  device=self.device,
                                                  seeds=actor_mod.shard_seeds(self.config.seed, self.rank, self.E),
                                                  group_width=group)
        per = self.E // n
        self.g_obs = [[o.reshape(self.E, -1)[self.pipe.slice(g)].contiguous() for g in range(n)] for o in self.obs_sets]
        self.g_in = (self.legal[:per], self.num_legal[:per], self.to_play[:per], self.temperature[:per])
        self.started = [False] * n

    # ---- lock-step workloads, several env groups on separate streams (one group's host work under the others' kernels)
    def _init_lockstep_pipeline(self, actor_mod):
        n = self.args.groups
        self.pipe = pkg("engine").PipelinedLockstep(self.config, self.E, self.model, groups=n, device=self.device,
                                                    seeds=actor_mod.shard_seeds(self.config.seed, self.rank, self.E),
                                                    use_graph=not self.args.no_graph,
                                                    device_noise=not self.args.host_noise)
        per = self.E // n
        self.g_obs = [[o[self.pipe.slice(g)].contiguous() for g in range(n)] for o in self.obs_sets]
        self.g_in = (self.legal[:per], self.num_legal[:per], self.to_play[:per], self.temperature[:per])
        self.started = [False] * n

    def _pipe_step(self, i):
        g_legal, g_nl, g_tp, g_temp = self.g_in
        for g in range(self.args.groups):
            if self.started[g]:
                self.pipe.finish(g)
                self.pipe.engines[g].sample_actions(g_temp)
            self.pipe.begin(g, self.g_obs[i % len(self.obs_sets)][g], g_legal, g_tp, True, num_legal=g_nl)
            self.started[g] = True

    def drain(self):
        if not self.pipe:
            return
        for g in range(self.args.groups):
            if self.started[g]:
                self.pipe.finish(g)
                self.pipe.engines[g].sample_actions(self.g_in[3])
                self.started[g] = False

    # ---- fused CartPole, moves queued in batches (no host round trip inside a batch) --------------------------------
    def batch_sizes(self, start, count):
        i, out = start, []
        while count > 0:
            b = min(self.args.moves_per_batch, count)
            if self.bcast:
                b = min(b, self.args.bcast_every - i % self.args.bcast_every)
            out.append((i, b))
            i += b
            count -= b
        return out

    def prepare(self, n_moves):
        self.engine.moves_prepare(n_moves, self.legal, self.to_play, self.temperature, True, num_legal=self.num_legal)

    def run_plan(self, plan, trailing_predraw):
        """Batches plan[i] = (first step, moves); the first one must already be drawn and uploaded.  The host draws a
        batch's exploration noise while the previous batch runs and collects actions / visit counts / root values of a
        batch when it is done; kernels of one batch follow each other without host round trips.  trailing_predraw =
        (moves) draws one more batch during the last one (left uploaded-ready)."""
        engine, trace = self.engine, os.environ.get("MZ_BENCH_TRACE")
        played = 0
        for n, (i, b) in enumerate(plan):
            if self.bcast and i % self.args.bcast_every == 0:
                self.refresh()
            t = [time.perf_counter()]
            for k in range(b):
                engine.moves_enqueue(self.flat_obs[(i + k) % len(self.flat_obs)])
            t.append(time.perf_counter())
            nxt = plan[n + 1][1] if n + 1 < len(plan) else trailing_predraw
            if nxt:
                engine.moves_predraw_next(nxt, self.legal, self.to_play, self.temperature, True, num_legal=self.num_legal)
            t.append(time.perf_counter())
            out = engine.moves_collect(copy=bool(os.environ.get("MZ_BENCH_COPY")))   # default: views of the download ring
            played += int(out["moves_done"].sum())   # an env whose stream left the pre-drawn path sits out the rest
            t.append(time.perf_counter())
            if n + 1 < len(plan):
                engine.moves_submit_next()
            t.append(time.perf_counter())
            self.pending_predraw = nxt if n + 1 == len(plan) else 0
            if trace:
                print(f"[bench] batch of {b}: enqueue {1e3 * (t[1] - t[0]):.2f} ms, predraw {1e3 * (t[2] - t[1]):.2f}, "
                      f"collect {1e3 * (t[3] - t[2]):.2f}, submit {1e3 * (t[4] - t[3]):.2f}; moves done min "
                      f"{int(out['moves_done'].min())}", file=sys.stderr, flush=True)
        return played

    def close(self):
        if self.pipe:
            self.pipe.close()
        self.actor.close()


def run_steps(wl, first, count, barrier, next_first=0):
    """`count` steps starting at step index `first`, bracketed by barrier + synchronize; returns (seconds, moves)."""
    if wl.batched:
        plan = wl.batch_sizes(first, count)
        drawn_ahead = getattr(wl, "pending_predraw", 0) == plan[0][1]
        if not drawn_ahead:                                  # the batch drawn ahead has another size: draw this one now
            wl.engine.moves_discard_next()
            wl.prepare(plan[0][1])
        barrier()
        t0 = time.perf_counter()
        # every timed batch is run and collected inside the timed region, and so is one noise draw per batch (the last
        # batch's draw produces rows nobody runs: it stands in for the first batch's, done before the timer started)
        if drawn_ahead:
            wl.engine.moves_submit_next()
        played = wl.run_plan(plan, next_first or plan[0][1])
        barrier()
        return time.perf_counter() - t0, played
    barrier()
    t0 = time.perf_counter()
    for i in range(first, first + count):
        wl.one_step(i)
    wl.drain()                                               # every queued move is finished inside the timed region
    barrier()
    return time.perf_counter() - t0, wl.E * count


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    if os.environ.get("MZ_BENCH_FAULT_RANK") is not None and os.environ.get("MZ_BENCH_FAULT_RANK") == os.environ.get("RANK"):
        sys.exit(3)                                          # fault injection for tests/test_bench_launcher.py: a rank that dies before the rendezvous
    if args.rehearse_cpu:
        return rehearse_cpu(args)
    cpu_helper = None
    if args.gpus == 1 and args.cpu_seconds > 0:
        # the cpu_baseline workers are forked from this helper, which is started before anything here touches the GPU
        cpu_helper = subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_selfplay.py"), "--serve"],
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    actor_mod = pkg("actor")
    rank, world, local_rank = actor_mod.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = torch.distributed
    rehearsal = bool(os.environ.get("MZ_REHEARSE_ON_ONE_GPU")) and world > 1

    wl = Workload(args, rank, world, device)
    E, S, A = wl.E, wl.S, wl.A

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def reduce_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather(x):
        if world == 1:
            return [x]
        out = [None] * world
        dist.all_gather_object(out, x)
        return out

    K, W = args.steps, args.warmup
    # ---- warm-up: W untimed steps, then an untimed probe of K steps that sizes the timed region ----------------------
    if wl.batched:
        plan_all = wl.batch_sizes(0, max(W, 1) + K)
        wl.prepare(max(b for _, b in plan_all + wl.batch_sizes(0, args.moves_per_batch)))
        wl.engine.moves_collect()                        # (sizes the batch buffers once; nothing was queued)
        warm = wl.batch_sizes(0, max(W, 1))
        wl.prepare(warm[0][1])
        wl.run_plan(warm, wl.batch_sizes(max(W, 1), K)[0][1])   # steady state: the next batch is drawn during warm-up
    else:
        for i in range(W):
            wl.one_step(i)
        wl.drain()
    next_first = wl.batch_sizes(max(W, 1) + K, 1 << 30)[0][1] if wl.batched else 0
    probe_s, _ = run_steps(wl, max(W, 1), K, barrier, next_first)
    probe_s = reduce_max(probe_s)
    repeats = max(1, int(math.ceil(args.min_seconds / max(probe_s, 1e-9))))
    # ---- timed region: exactly repeats x K steps ---------------------------------------------------------------------
    refreshes_before = wl.refreshes
    elapsed, moves_played = run_steps(wl, max(W, 1) + K, repeats * K, barrier)
    if wl.batched:
        wl.engine.moves_discard_next()
    per_rank = gather((elapsed, moves_played))
    elapsed = max(e for e, _ in per_rank)
    moves_played = sum(m for _, m in per_rank)
    timed_steps = repeats * K
    # only searches that really ran count: in move batches an env whose tie-break words differ from the pre-drawn
    # assumption sits out the rest of its batch, so moves_played <= world * E * timed_steps
    value = moves_played * S / elapsed

    # ---- the weight refresh by itself (not part of `value`; N > 1) ---------------------------------------------------
    bcast_us = None
    if world > 1:
        barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            wl.actor.flat.broadcast(src=0)
        barrier()
        bcast_us = reduce_max(1e6 * (time.perf_counter() - t0) / 20)
    # after the last refresh every network replica of every rank (one per env group) must hold rank 0's weights
    replica_sums = None
    if world > 1:
        wl.refresh()
        torch.cuda.synchronize(device)
        nets = [wl.model] + (list(getattr(wl.pipe, "models", [])[1:]) if wl.pipe is not None else [])
        mine = [float(sum(v.double().sum() for v in net.state_dict().values() if v.dtype == torch.float32)) for net in nets]
        replica_sums = gather(mine)

    fused, batched = wl.fused, wl.batched
    engine = wl.engine
    network = (f"fullyconnected ({wl.weights_kind}), fp32 inference "
               + ("in the fused HIP kernel" if fused else "through PyTorch-ROCm")) if wl.name == "cartpole" else \
        (f"resnet {wl.config.blocks} blocks x {wl.config.channels} channels ({wl.weights_kind}), fp32 inference through "
         "PyTorch-ROCm + HIP network kernels")
    result = {
        "metric": "mcts_simulations_per_sec", "value": value, "unit": "simulations/s",
        "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / timed_steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{wl.name}_{E}envs_x_{S}sims", "baseline_config": WORKLOADS[wl.name]["baseline_config"],
                   "envs_per_gpu": E, "simulations": S, "actions": A, "network": network,
                   "mode": "fused" if fused else "lockstep", "lanes_per_tree": engine.group_width(),
                   "env_groups_per_gpu": args.groups if (wl.pipe is not None) else 1,
                   "moves_per_host_round_trip": args.moves_per_batch if batched else 1,
                   "fused_kernel": engine.fused_variant() if fused else None,
                   "launch": "one kernel per move" if fused else ("eager" if args.no_graph else "hipgraph"),
                   "parallelism": f"actors{world}",
                   "exploration_noise": "drawn on the GPU (root_noise_kernel)" if engine._device_noise else
                                        "drawn on the host mirrors of the RNG streams",
                   "weight_broadcast_every_steps": args.bcast_every if world > 1 else None},
        "timed_steps": timed_steps, "timed_seconds": elapsed, "repeats_of_steps": repeats,
        "self_play_moves_per_sec": moves_played / elapsed,
        "moves_played": moves_played, "moves_scheduled": world * E * timed_steps,
    }
    if world > 1:
        backend = dist.get_backend()
        result.update({
            "per_rank_simulations_per_sec": [m * S / e for e, m in per_rank],
            "collective_backend": backend, "rccl_ranks": world if backend == "nccl" else 0,
            "weight_refreshes_in_timed_region": wl.refreshes - refreshes_before,
            "weight_broadcast_us": bcast_us, "weight_bytes": wl.actor.flat.nbytes(),
            "network_replicas_per_rank": len(replica_sums[0]),
            "weights_identical_on_every_replica_of_every_rank": len({x for row in replica_sums for x in row}) == 1,
            "self_launched": bool(os.environ.get("MZ_BENCH_SELF_LAUNCHED"))})
        if rehearsal:
            result["rehearsal"] = f"{world} ranks share cuda:0 over gloo (fewer GPUs than ranks): not a measurement"

    if rank == 0:
        if args.profile_steps > 0:
            # (a fused launch lasts 0.2 ms and its duration follows the positions searched: the pass covers every synthetic
            #  position set several times, so that the HIP-event mean is the mean of the timed region's launches)
            profile_steps = max(args.profile_steps, 4 * len(wl.obs_sets)) if wl.fused else args.profile_steps
            result["roofline"], result["kernels"] = roofline_leg(wl, profile_steps, device)
        if cpu_helper is not None:
            result["cpu_baseline"] = cpu_baseline_leg(cpu_helper, wl.name, wl.config, args.cpu_seconds, args.cpu_workers)
            result["speedup_vs_cpu_port_1core"] = value / result["cpu_baseline"]["value"]
            if result["cpu_baseline"].get("all_cores"):
                result["speedup_vs_cpu_port_all_cores"] = value / result["cpu_baseline"]["all_cores"]["value"]
    barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    wl.close()
    if cpu_helper is not None:
        cpu_helper.stdin.close()
        cpu_helper.wait(timeout=60)
    if world > 1:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------
# N > 1 plumbing on CPU (tests/test_bench_launcher.py): no GPU, no search -- never a measurement
# ----------------------------------------------------------------------------------------------------------------
def rehearse_cpu(args):
    dist = torch.distributed
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    actor_mod, weights_mod, models = pkg("actor"), pkg("weights"), pkg("models")
    config = workload_config(args.workload)
    weights, _ = workload_weights(args.workload, config)
    model = models.MuZeroNetwork(config)
    model.set_weights(weights)
    model.eval()
    flat = weights_mod.FlatWeights(model)
    if rank != 0:
        with torch.no_grad():
            flat.flat.add_(float(rank))                      # stale weights everywhere but on rank 0
    seeds = actor_mod.shard_seeds(config.seed, rank, 4)
    E = 4
    refreshes = 0
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if world > 1 and args.bcast_every and i % args.bcast_every == 0:
            flat.broadcast(src=0)
            refreshes += 1
        time.sleep(0.001)                                    # stands in for one move of search
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    checksum = float(flat.flat.double().sum())
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (elapsed, checksum, seeds))
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    else:
        gathered = [(elapsed, checksum, seeds)]
    if rank == 0:
        print(json.dumps({
            "metric": "launcher_rehearsal_not_a_measurement", "value": None, "unit": None, "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "collective_backend": "gloo", "rccl_ranks": 0, "weight_refreshes": refreshes,
            "weights_identical_after_refresh": len({c for _, c, _ in gathered}) == 1,
            "rank_seeds": [s for _, _, s in gathered], "envs_per_rank": E,
            "self_launched": bool(os.environ.get("MZ_BENCH_SELF_LAUNCHED")),
            "config": {"workload": f"{args.workload}_plumbing_only"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------
# roofline leg
# ----------------------------------------------------------------------------------------------------------------
def roofline_leg(wl, steps, device):
    """Eager pass with HIP events around every tree-kernel launch (bound to the dispatch, on the launch stream); for
    the lock-step workloads also torch events (same stream: the network kernels are launched on torch's current
    stream) around the network launches of one recurrent inference."""
    engine = wl.engine
    engine.set_profiling(True)
    engine.get_profile(reset=True)
    for i in range(2):
        wl.search_only_step(i)
    torch.cuda.synchronize(device)
    engine.get_profile(reset=True)
    for i in range(steps):
        wl.search_only_step(i)
    torch.cuda.synchronize(device)
    prof = engine.get_profile(reset=True)
    engine.set_profiling(False)
    sims = max(prof["simulations"], 1)
    mean_depth = prof["select_depth_sum"] / sims
    bytes_sim = engine.algorithmic_bytes_per_simulation(mean_depth)
    # SURVEY 8(d) charges the hidden-state gather (read the parent's state, write the contiguous batch: 2 * 4 * H bytes)
    # to the descent.  Where the tower kernel gathers its own input from the pool (mzmcts_board_tower_gathered) `select`
    # moves none of those bytes: they are charged to the kernel that moves them, the tower (read only: there is no batch)
    gather_bytes = 2 * 4 * engine.H
    towers_gather = (not wl.fused) and engine._pool_path(wl.model)
    if towers_gather:
        bytes_sim = dict(bytes_sim, select=bytes_sim["select"] - gather_bytes, total=bytes_sim["total"] - gather_bytes // 2,
                         gather_moved_by="board tower kernel (reads the parent state straight from the pool; 4 * H bytes)")
    kernels = {}
    per_kernel = [("select", "select_ms", "select_launches", bytes_sim["select"] * engine.E),
                  ("expand_backup", "expand_backup_ms", "expand_backup_launches", bytes_sim["expand_backup"] * engine.E),
                  # one fused launch = S simulations of every tree: select + expand/backup bytes of all of them
                  ("search_fused_fc", "fused_ms", "fused_launches", bytes_sim["total"] * engine.E * engine.S)]
    for name, ms_key, n_key, per_launch in per_kernel:
        if prof[n_key] == 0:
            continue
        avg_us = 1e3 * prof[ms_key] / prof[n_key]
        kernels[name] = {"bound": "hbm", "avg_us": avg_us, "launches": prof[n_key],
                         "algorithmic_bytes_per_launch": per_launch,
                         "achieved_GBs": per_launch / (avg_us * 1e-6) / 1e9 if avg_us > 0 else None,
                         "frac_of_hbm_peak": per_launch / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS if avg_us > 0 else None}
    if not wl.fused:
        kernels.update(network_leg(wl, device))
        if towers_gather and "board_tower" in kernels:
            kernels["board_tower"]["hbm_bytes_gathered_per_launch"] = 4 * engine.H * engine.E
    # the dominant KERNEL: the whole-call entry (several launches) is reported but is not a kernel
    single = {k: v for k, v in kernels.items() if k != "recurrent_inference" or "board_tower" not in kernels}
    dominant = max(single, key=lambda k: single[k]["avg_us"])
    d = kernels[dominant]
    if d["bound"] == "mfma":
        roofline = {"bound": "mfma", "kernel": d["what"], "achieved": d["achieved_TFLOPs"],
                    "peak": d["peak_TFLOPs"], "unit": "TFLOP/s", "frac": d["frac_of_matrix_peak"],
                    "traffic": None, "avg_kernel_us": d["avg_us"], "flops_per_launch": d["flops_per_launch"],
                    "precision": d.get("precision", "fp32"),
                    "fp32_equivalent_TFLOPs": d.get("fp32_equivalent_TFLOPs"),
                    "timing": "torch.cuda.Event pairs on the launch stream around the launch, eager, mean over repeats",
                    "mean_select_depth": mean_depth, "algorithmic_bytes_per_simulation": bytes_sim}
        return roofline, kernels
    kernel_name = f"mz::{dominant}_kernel"
    if dominant == "search_fused_fc" and engine.fused_variant() == "narrow":
        kernel_name = "mz::search_fused_narrow_kernel"
    traffic, traffic_source = pmc_traffic(kernel_name, engine.E)
    # the fused kernels keep their trees in LDS: what bounds them is the dependent instruction chain of one tree (one
    # wavefront per SIMD at 4096 envs), not HBM; `achieved` / `frac` stay the algorithmic-bytes figure SURVEY 8(d) defines
    # (also under `algorithmic_hbm`), next to what the counters say really crosses the HBM interface
    bound = "latency/issue" if dominant == "search_fused_fc" else "hbm"
    roofline = {"bound": bound, "kernel": kernel_name, "achieved": d["achieved_GBs"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["achieved_GBs"] / HBM_PEAK_GBS,
                "algorithmic_hbm": {"achieved_GBs": d["achieved_GBs"], "peak_GBs": HBM_PEAK_GBS,
                                    "frac": d["achieved_GBs"] / HBM_PEAK_GBS,
                                    "bytes_per_launch": d["algorithmic_bytes_per_launch"]},
                "traffic": traffic, "traffic_source": traffic_source,
                "avg_kernel_us": d["avg_us"], "mean_select_depth": mean_depth,
                "algorithmic_bytes_per_simulation": bytes_sim,
                "timing": "HIP events bound to the kernel dispatch (hipExtLaunchKernel) on the launch stream, "
                          "over a second pass of the same steps",
                "working_set_bytes": engine.device_bytes()}
    if traffic:
        # what really crosses the HBM interface (PMC counters) vs the algorithmic bytes `achieved` is defined on
        roofline["hbm_traffic_GBs"] = traffic / (d["avg_us"] * 1e-6) / 1e9
        roofline["frac_hbm_traffic"] = roofline["hbm_traffic_GBs"] / HBM_PEAK_GBS
        roofline["traffic_over_algorithmic"] = traffic / d["algorithmic_bytes_per_launch"]
    if dominant == "search_fused_fc":
        roofline["limiter"] = ("latency: the trees live in LDS (HBM sees only the published roots), one wavefront per "
                               "SIMD at 4096 envs, so a launch lasts one tree's dependent instruction chain; "
                               "`frac` says how far the workload is from an HBM-sized one, not how well HBM is used")
        issue = issue_profile(kernel_name)
        if issue:
            roofline["issue"] = issue
    return roofline, kernels


def tower_conv_flops(config):
    """FLOPs (2 per multiply-add) of the 3x3 convolutions of one recurrent inference of one sample: the dynamics stem
    and the residual blocks of the dynamics and prediction networks -- what mzmcts_board_tower runs."""
    _, h, w = config.observation_shape
    if config.downsample:
        h, w = math.ceil(h / 16), math.ceil(w / 16)
    c, b = config.channels, config.blocks
    return _conv_flops(c + 1, c, 3, h, w) + 4 * b * _conv_flops(c, c, 3, h, w)


def _timed(call, device, repeats):
    for _ in range(3):
        call()
    torch.cuda.synchronize(device)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(repeats):
        call()
    stop.record()
    torch.cuda.synchronize(device)
    return 1e3 * start.elapsed_time(stop) / repeats


def network_leg(wl, device, repeats=20):
    """The network side of one simulation for all envs, eager, torch events on the launch stream: the residual tower
    kernel by itself (the dominant launch: every 3x3 convolution of the dynamics and prediction networks) and the whole
    recurrent inference (tower + heads).  FLOPs vs the matrix peak of the path the tower runs on: exact fp32 MFMA for
    16-channel networks; for 64-channel networks every fp32 product is three fp16 MFMA products (csrc/board_conv.hip),
    so the executed FLOPs are 3 x the algorithmic ones and the peak is the dense f16 one."""
    engine, model = wl.engine, wl.model
    out = {}
    with torch.no_grad():
        if len(engine.state_shape) == 3 and hasattr(model, "recurrent_inference_from_planes"):
            slab = engine.pool[1].view(engine.E, *engine.state_shape)
            if engine._pool_path(model):
                # the towers gather their own input from the hidden-state pool (mzmcts_board_tower_gathered): what the
                # last search's last descent left in leaf_parent / batch_action
                c, h, w = engine.state_shape
                shape, gather = (engine.E, c + 1, h, w), engine.tower_gather()
                whole = lambda: model.recurrent_inference_from_pool(gather, engine.E, out_state=slab)  # noqa: E731
                tower = lambda: model._recurrent_tower(None, slab, gather=gather, shape=shape, device=slab.device)  # noqa: E731
            else:
                planes = engine.batch_planes                 # the dynamics input the last search's gather left behind
                whole = lambda: model.recurrent_inference_from_planes(planes, out_state=slab)  # noqa: E731
                tower = lambda: model._recurrent_tower(planes, slab)  # noqa: E731
            if tower() is not None:
                split = wl.config.channels == 64 and os.environ.get("MZ_BOARD_CONV_PRECISION", "split") != "fp32"
                us = _timed(tower, device, repeats)
                alg = tower_conv_flops(wl.config) * engine.E
                executed = alg * (3 if split else 1)
                peak = F16_MATRIX_PEAK_TFLOPS if split else FP32_MATRIX_PEAK_TFLOPS
                tf = executed / (us * 1e-6) / 1e12
                _, bh, bw = engine.state_shape
                cols = (not split and wl.config.channels == 16 and (bh, bw) in ((3, 3), (6, 6))
                        and os.environ.get("MZ_TOWER_COLS", "on") != "off")
                kernel = ("mz::board_tower_split_kernel" if split else
                          ("mz::board_tower_cols_kernel" if (bh, bw) == (3, 3) else "mz::board_tower_patch_kernel") if cols else
                          "mz::board_tower_kernel")
                if cols and (bh, bw) == (3, 3):
                    # the board-column kernel skips the products with padding zeros (taps outside a 3 x 3 board): 49 of the
                    # 81 (position, tap) pairs are multiplied; `achieved` stays the algorithmic count of a padded convolution
                    executed = alg * 49.0 / 81.0
                out["board_tower"] = {
                    "bound": "mfma", "what": kernel,
                    "avg_us": us, "launches": repeats, "flops_per_launch": executed, "algorithmic_flops_per_launch": alg,
                    "achieved_TFLOPs": tf, "peak_TFLOPs": peak, "frac_of_matrix_peak": tf / peak,
                    "executed_TFLOPs": executed / (us * 1e-6) / 1e12,
                    "precision": "two fp16 halves per operand, three products, fp32 accumulate" if split else "fp32 MFMA",
                    "fp32_equivalent_TFLOPs": alg / (us * 1e-6) / 1e12}
        else:
            hidden = engine.batch_hidden.view(engine.E, *engine.state_shape)
            whole = lambda: model.recurrent_inference(hidden, engine.batch_action, out_state=engine.pool[1].view(engine.E, *engine.state_shape))  # noqa: E731
        us = _timed(whole, device, repeats)
    flops = recurrent_inference_flops(wl.config) * engine.E
    tower = out.get("board_tower")
    # FLOPs the call really executes, against the peak of the pipe its dominant part runs on: a split tower executes three
    # f16 products per fp32 product (against the f16 peak); heads and everything else are fp32
    executed = flops + (tower["flops_per_launch"] - tower["algorithmic_flops_per_launch"] if tower else 0)
    peak = tower["peak_TFLOPs"] if tower else FP32_MATRIX_PEAK_TFLOPS
    out["recurrent_inference"] = {"bound": "mfma", "what": "recurrent_inference (tower + heads, all launches of one call)",
                                  "avg_us": us, "launches": repeats, "flops_per_launch": executed,
                                  "algorithmic_flops_per_launch": flops,
                                  "achieved_TFLOPs": executed / (us * 1e-6) / 1e12, "peak_TFLOPs": peak,
                                  "frac_of_matrix_peak": executed / (us * 1e-6) / 1e12 / peak,
                                  "fp32_equivalent_TFLOPs": flops / (us * 1e-6) / 1e12,
                                  "note": "FLOPs executed by the whole call (a split tower's three f16 products per fp32 "
                                          "product counted as executed) against the peak of the pipe the tower runs on"}
    return out


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


def issue_profile(kernel):
    """Instruction-issue picture of the fused kernel from the committed SQ-counter pass (profiles/r*_fused_sq.json,
    written by tools/pmc_summary.py --sq): instructions and cycles per simulation, cycles per instruction, LDS share."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fused_sq.json")))
    if not files:
        return None
    data = json.load(open(files[-1]))
    entry = data.get("kernels", {}).get(kernel) or next((v for k, v in data.get("kernels", {}).items() if kernel in k), None)
    if entry is None:
        return None
    return dict(entry, source=os.path.relpath(files[-1], ROOT))


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline leg (the oracle is the thing timed here, on the host cores; it is not on the product path)
# ----------------------------------------------------------------------------------------------------------------
def available_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:                                                     # a cgroup CPU quota is the real share of a container
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except Exception:
        pass
    return n


def cpu_baseline_leg(helper, name, config, seconds, workers=0):
    """The oracle's port of the reference loop on the host: one core, then one process per available core.  The
    worker processes belong to `helper` (oracle/cpu_selfplay.py --serve), started before this process opened the GPU."""
    try:
        cpu_model = [line.split(":", 1)[1].strip() for line in open("/proc/cpuinfo") if line.startswith("model name")][0]
    except Exception:
        cpu_model = "unknown"
    n = workers or available_cores()
    if name != "cartpole":
        n = min(n, 64)                                       # each worker holds a torch runtime (~0.4 GiB)
    helper.stdin.write(f"{name} {seconds} {n}\n")
    helper.stdin.flush()
    answer = json.loads(helper.stdout.readline())
    one, many, wall = answer["one"], answer["many"], answer["wall"]
    assert one, "cpu_baseline worker failed"
    value = one["sims"] / one["seconds"]
    kind_note = ("tree and FC network in C, one tree at a time" if name == "cartpole" else
                 "tree in C, residual network at batch 1 through torch on the CPU (1 thread), one tree at a time")
    out = {"value": value, "unit": "simulations/s", "cores": 1, "kind": "port",
           "sample": f"{one['moves']} moves x {config.num_simulations} sims ({one['seconds']:.1f} s), same weights and "
                     f"observation distribution; {kind_note}",
           "cpu": cpu_model, "host_cores_available": available_cores(), "os_cpu_count": os.cpu_count()}
    ref = REFERENCE_TIMING.get(name)
    if ref:
        procs = (REFERENCE_TIMING_HOST or {}).get("procs")
        out["reference_in_build_container"] = {
            "host": REFERENCE_TIMING_HOST, "reference_sims_per_s_1proc": ref["reference_sims_per_s_1proc"],
            f"reference_sims_per_s_{procs}proc": ref.get(f"reference_sims_per_s_{procs}proc"),
            "port_sims_per_s_1proc": ref["port_sims_per_s_1proc"], "rho_1proc": ref["rho_1proc"],
            f"rho_{procs}proc": ref.get(f"rho_{procs}proc"),
            "source": "tests/golden/g10_reference_speed_configs.json (tests/golden/time_reference.py runs the imported reference)"}
        out["rho_port_over_reference_build_container"] = ref["rho_1proc"]
        out["reference_equivalent_value_extrapolated"] = value / ref["rho_1proc"]
    if many:
        total = sum(o["sims"] / o["seconds"] for o in many)
        out["all_cores"] = {"value": total, "unit": "simulations/s", "cores": len(many), "kind": "port",
                            "sample": f"{len(many)} processes x {seconds:.0f} s, one per core, each as the 1-core leg "
                                      f"(seeds config.seed + worker index, muzero.py:175); wall {wall:.1f} s",
                            "per_core_value": total / len(many)}
        if ref:
            procs = (REFERENCE_TIMING_HOST or {}).get("procs")
            rho_many = ref.get(f"rho_{procs}proc") or ref["rho_1proc"]
            out["all_cores"]["reference_equivalent_value_extrapolated"] = total / rho_many
    return out


if __name__ == "__main__":
    main()

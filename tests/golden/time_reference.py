#!/usr/bin/env python3
"""time the ACTUAL reference (imported from /root/reference, build container only) and
the oracle's port on the same synthetic inputs, with 1 process and with one process per core, and record
rho = port / reference per config.  bench.py reads the result (tests/golden/g10_reference_speed_configs.json: data, not
code) to print "reference-equivalent" CPU figures next to what the port measures on the GPU box's host.

    python tests/golden/time_reference.py [--seconds 20] [--procs 8]

The reference path timed is self_play.MCTS(config).run(model, observation, legal_actions, to_play, True) followed by
SelfPlay.select_action(root, 1.0) -- one move -- on the inputs of oracle/cpu_selfplay.synthetic_inputs,
torch.set_num_threads(1), Ray replaced by the in-process stand-in of make_golden.py.  The port is
oracle/cpu_selfplay.run (tree in C; residual networks through this package's torch modules at batch 1)."""
import argparse
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def reference_worker(name, seconds, seed):
    import numpy
    import torch
    import make_golden as mg
    import cpu_selfplay
    models, self_play = mg.import_reference()
    torch.set_num_threads(1)
    config = mg.make_configs()[name]
    if name == "cartpole":
        model = mg.build_model(models, config, mg.load_cartpole_checkpoint())
    else:
        from synth import synthetic_state_dict
        model = models.MuZeroNetwork(config)
        model.set_weights({k: torch.from_numpy(v) for k, v in synthetic_state_dict(model.state_dict(), 0).items()})
        model.eval()
    obs, legal, num_legal, to_play = cpu_selfplay.synthetic_inputs(name, cpu_selfplay.workload_config(name), 64, 1 if name != "cartpole" else 123)
    numpy.random.seed(config.seed + seed)
    sims = moves = 0
    t0 = None
    i = 0
    with torch.no_grad():
        while True:
            if t0 is not None and time.perf_counter() - t0 >= seconds:
                break
            e = i % len(obs)
            root, _ = self_play.MCTS(config).run(model, obs[e], [int(a) for a in legal[e, :num_legal[e]]], int(to_play[e]), True)
            self_play.SelfPlay.select_action(root, 1.0)
            if t0 is None:
                t0 = time.perf_counter()                  # the first move is the warm-up
            else:
                sims += config.num_simulations
                moves += 1
            i += 1
    return dict(sims=sims, moves=moves, seconds=time.perf_counter() - t0)


def spawn(kind, name, seconds, n):
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", kind, name, str(seconds), str(i)],
                              stdout=subprocess.PIPE, text=True, env=env) for i in range(n)]
    outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    return sum(o["sims"] / o["seconds"] for o in outs), sum(o["moves"] for o in outs)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        kind, name, seconds, seed = sys.argv[2], sys.argv[3], float(sys.argv[4]), int(sys.argv[5])
        if kind == "reference":
            print(json.dumps(reference_worker(name, seconds, seed)))
        else:
            import cpu_selfplay
            print(json.dumps(cpu_selfplay.run(name, seconds, seed)))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    ap.add_argument("--configs", nargs="*", default=["cartpole", "tictactoe", "connect4", "atari84"])
    args = ap.parse_args()
    cpu = open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")
    out = {"cpu": cpu, "cores": os.cpu_count(), "seconds_per_leg": args.seconds, "procs": args.procs, "configs": {}}
    for name in args.configs:
        row = {}
        for kind in ("reference", "port"):
            for n in (1, args.procs):
                rate, moves = spawn(kind, name, args.seconds, n)
                row[f"{kind}_sims_per_s_{n}proc"] = rate
                row[f"{kind}_moves_{n}proc"] = moves
                print(f"{name:10s} {kind:9s} {n} proc: {rate:12.1f} sims/s ({moves} moves)", flush=True)
        row["rho_1proc"] = row["port_sims_per_s_1proc"] / row["reference_sims_per_s_1proc"]
        row[f"rho_{args.procs}proc"] = row[f"port_sims_per_s_{args.procs}proc"] / row[f"reference_sims_per_s_{args.procs}proc"]
        out["configs"][name] = row
        with open(os.path.join(HERE, "g10_reference_speed_configs.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out["configs"], indent=1))


if __name__ == "__main__":
    main()

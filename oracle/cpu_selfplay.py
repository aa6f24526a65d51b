#!/usr/bin/env python3
"""CPU baseline worker of bench.py's `cpu_baseline` leg (TEST INFRASTRUCTURE -- never on the product path).

One process = one host core running the reference's one-tree-at-a-time loop (self_play.py:261-362 + 223-246)
through the oracle's C restatement (oracle/mz_oracle.c) for a bounded number of seconds:

  cartpole                          tree AND the fully-connected network in C (oracle_fc_selfplay_moves)
  tictactoe / connect4 / atari84    tree in C, the residual network evaluated at batch 1 through torch on the CPU
                                    (one thread), as the reference does -- the package's models.py builds the same
                                    modules with the same state-dict keys and is pinned to the reference by
                                    fixture G3 (tests/test_models_cpu.py)

    python oracle/cpu_selfplay.py <workload> <seconds> [seed]      -> one JSON line {"sims", "moves", "seconds"}

    python oracle/cpu_selfplay.py --serve       -> reads "<workload> <seconds> <workers>" lines on stdin and answers
                                                   each with one JSON line {"one": {...}, "many": [{...}, ...], "wall": s}

bench.py starts the `--serve` form BEFORE it touches the GPU and asks it for the two legs (1 core, one process per
available core) after the GPU measurements: every worker is then forked from a process that never held the GPU (a
GPU box allows only a few processes with the device open, and a fork of the benchmark process would count).
"""
import importlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def workload_config(name):
    if name == "atari84":
        return importlib.import_module("muzero-hypermodel_amd.games.breakout").atari84_config()
    return importlib.import_module(f"muzero-hypermodel_amd.games.{name}").MuZeroConfig()


def synthetic_inputs(name, config, n, seed):
    """Observations / legal sets / to_play of `n` synthetic root positions."""
    rs = np.random.RandomState(seed)
    A, P = len(config.action_space), len(config.players)
    C, H, W = config.observation_shape
    if name == "cartpole":
        obs = rs.uniform(-0.05, 0.05, (n, C, H, W)).astype(np.float32)
    elif name == "atari84":
        obs = rs.uniform(0, 1, (n, C, H, W)).astype(np.float32)
    else:
        obs = rs.randint(0, 2, (n, C, H, W)).astype(np.float32)
        obs[:, 2] = 1.0
    legal = np.zeros((n, A), np.int32)
    num_legal = np.zeros(n, np.int32)
    for e in range(n):
        k = A if P == 1 else int(rs.randint(max(1, A // 2), A))        # board games: at least one illegal root action
        legal[e, :k] = np.sort(rs.choice(A, size=k, replace=False))
        num_legal[e] = k
    to_play = rs.randint(0, P, n).astype(np.int32)
    return obs, legal, num_legal, to_play


def run_cartpole(seconds, seed):
    import mz_oracle
    from parity_helpers import load_golden
    config = workload_config("cartpole")
    w = load_golden("cartpole_weights")
    net = mz_oracle.FcNet.from_config(config, {k: w[k] for k in w.files})
    cfg = mz_oracle.config_from_muzero(config, H=config.encoding_size)
    obs, _, _, _ = synthetic_inputs("cartpole", config, 1024, 123)
    rng = mz_oracle.Rng(config.seed + seed)
    mz_oracle.fc_selfplay_moves(cfg, net, rng, obs[:64], 1.0)          # warm
    t0 = time.perf_counter()
    sims = moves = 0
    while time.perf_counter() - t0 < seconds:
        out = mz_oracle.fc_selfplay_moves(cfg, net, rng, obs[:256], 1.0)
        sims += out["sims"]
        moves += 256
    return dict(sims=sims, moves=moves, seconds=time.perf_counter() - t0)


def run_residual(name, seconds, seed):
    import torch
    import mz_oracle
    from parity_helpers import synthetic_model
    torch.set_num_threads(1)
    models = importlib.import_module("muzero-hypermodel_amd.models")
    engine = importlib.import_module("muzero-hypermodel_amd.engine")
    config = workload_config(name)
    model, _ = synthetic_model(models, config, "cpu")
    A, F = len(config.action_space), 2 * config.support_size + 1
    shape = engine.hidden_state_shape(config)
    H = int(np.prod(shape))
    cfg = mz_oracle.config_from_muzero(config, H=H)
    obs, legal, num_legal, to_play = synthetic_inputs(name, config, 64, 1)
    as_array = np.ctypeslib.as_array

    def cb(user, hid, a, vl, rl, pl, nh):
        with torch.no_grad():
            state = torch.from_numpy(as_array(hid, (H,)).copy()).view(1, *shape)
            v, r, p, h = model.recurrent_inference(state, torch.tensor([[a]]))
        as_array(vl, (F,))[:] = v.numpy()[0]
        as_array(rl, (F,))[:] = r.numpy()[0]
        as_array(pl, (A,))[:] = p.numpy()[0]
        as_array(nh, (H,))[:] = h.numpy().reshape(-1)

    rng = mz_oracle.Rng(config.seed + seed)
    tree = mz_oracle.Tree(cfg)
    sims = moves = 0
    t0 = None
    i = 0
    while True:
        if t0 is not None and time.perf_counter() - t0 >= seconds:
            break
        e = i % len(obs)
        with torch.no_grad():
            v, r, p, h = model.initial_inference(torch.from_numpy(obs[e:e + 1]))
        root_reward = float(mz_oracle.support_to_scalar(r.numpy(), config.support_size)[0])
        tree.reset(rng, legal[e, :num_legal[e]], int(to_play[e]), root_reward, root_policy_logits=p.numpy()[0],
                   root_hidden=h.numpy().reshape(-1))
        tree.simulate(rng, callback=cb)
        mz_oracle.select_action(rng, tree.root_stats()["visits"], 1.0)
        if t0 is None:
            t0 = time.perf_counter()                                    # the first move is the warm-up
        else:
            sims += cfg.S
            moves += 1
        i += 1
    return dict(sims=sims, moves=moves, seconds=time.perf_counter() - t0)


def run(name, seconds, seed=0):
    return run_cartpole(seconds, seed) if name == "cartpole" else run_residual(name, seconds, seed)


def _spawn(name, seconds, n):
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="",
               ROCR_VISIBLE_DEVICES="")                     # CPU work only: never open the GPU
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), name, str(seconds), str(i)],
                              stdout=subprocess.PIPE, text=True, env=env) for i in range(n)]
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=seconds + 600)
        if p.returncode == 0 and out.strip():
            outs.append(json.loads(out.strip().splitlines()[-1]))
    return outs, time.perf_counter() - t0


def serve():
    for line in sys.stdin:
        parts = line.split()
        if not parts:
            continue
        name, seconds, workers = parts[0], float(parts[1]), int(parts[2])
        one, _ = _spawn(name, seconds, 1)
        many, wall = _spawn(name, seconds, workers) if workers > 1 else ([], 0.0)
        print(json.dumps({"one": one[0] if one else None, "many": many, "wall": wall}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--serve":
        serve()
    else:
        print(json.dumps(run(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0)))

#!/usr/bin/env python3
"""Throughput of the lock-step path (PyTorch-ROCm inference + HIP tree kernels, hipGraph replay) on the
other BASELINE.json configs: TicTacToe ResNet (25 sims, masked roots), Connect4 ResNet (200 sims),
Atari-like 84x84x4 CNN representation (50 sims).  Synthetic weights (tests/golden/synth.py), synthetic
observations, Dirichlet noise on.  One JSON line per config.

    python tools/bench_configs.py [tictactoe:4096 connect4:1024 atari84:1024] [--moves 20] [--warm 3] [--no-graph]

(--no-graph --warm 1 --moves 1 is the form profiled under rocprofv3 --pmc: eager launches, few of them.)
"""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from parity_helpers import synthetic_model
eng = importlib.import_module("muzero-hypermodel_amd.engine")
models = importlib.import_module("muzero-hypermodel_amd.models")

def config_of(name):
    if name == "atari84":
        return importlib.import_module("muzero-hypermodel_amd.games.breakout").atari84_config()
    return importlib.import_module(f"muzero-hypermodel_amd.games.{name}").MuZeroConfig()

if "--miopen-find" in sys.argv:
    torch.backends.cudnn.benchmark = True
args = [a for a in sys.argv[1:] if not a.startswith("--") and ":" in a] or ["tictactoe:4096", "connect4:1024", "atari84:1024"]
moves = int(sys.argv[sys.argv.index("--moves") + 1]) if "--moves" in sys.argv else 20
warm = int(sys.argv[sys.argv.index("--warm") + 1]) if "--warm" in sys.argv else 3
use_graph = "--no-graph" not in sys.argv
for spec in args:
    name, E = spec.split(":"); E = int(E)
    cfg = config_of(name)
    A, S = len(cfg.action_space), cfg.num_simulations
    model, _ = synthetic_model(models, cfg, "cuda")
    rs = np.random.RandomState(1)
    C, H, W = cfg.observation_shape
    if name == "atari84":
        obs = torch.from_numpy(rs.uniform(0, 1, (E, C, H, W)).astype(np.float32)).cuda()
    else:
        o = rs.randint(0, 2, (E, C, H, W)).astype(np.float32); o[:, 2] = 1.0
        obs = torch.from_numpy(o).cuda()
    # masked roots: every env has at least one illegal action (except single-player games)
    legal = np.zeros((E, A), np.int32); nl = np.zeros(E, np.int32)
    for e in range(E):
        n = A if len(cfg.players) == 1 else int(rs.randint(max(1, A // 2), A))
        acts = np.sort(rs.choice(A, size=n, replace=False)); legal[e, :n] = acts; nl[e] = n
    tp = rs.randint(0, len(cfg.players), E).astype(np.int32)
    engine = eng.BatchedMCTS(cfg, E, use_graph=use_graph)
    for _ in range(warm):
        engine.search(model, obs, legal, tp, True, num_legal=nl); engine.sample_actions(1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(moves):
        st = engine.search(model, obs, legal, tp, True, num_legal=nl); engine.sample_actions(1.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    d = float(st["depth_sum"].sum()) / (E * S)
    print(json.dumps({"config": name, "envs": E, "actions": A, "simulations": S, "hidden_floats": engine.H,
                      "sims_per_s": E * S * moves / dt, "moves_per_s": E * moves / dt, "ms_per_move_batch": 1e3 * dt / moves,
                      "mean_select_depth": d, "device_pool_GiB": engine.device_bytes() / 2**30}), flush=True)
    engine.close(); del engine, model; torch.cuda.empty_cache()

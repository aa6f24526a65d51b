#!/usr/bin/env python3
"""Diagnostic: wall time of the host-side phases of one fused move (bench workload)."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from parity_helpers import cartpole_model_and_weights
eng = importlib.import_module("muzero-hypermodel_amd.engine")
models = importlib.import_module("muzero-hypermodel_amd.models")
cfg = importlib.import_module("muzero-hypermodel_amd.games.cartpole").MuZeroConfig()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model, _ = cartpole_model_and_weights(models, cfg, "cuda")
e = eng.BatchedMCTS(cfg, E, group_width=16)
e.configure_fused_fc(model)
obs = torch.from_numpy(np.random.RandomState(0).uniform(-0.05, 0.05, (E, 4)).astype(np.float32)).cuda()
legal = np.tile(np.arange(2, dtype=np.int32), (E, 1)); nl = np.full(E, 2, np.int32); tp = np.zeros(E, np.int32)
temp = np.ones(E)
t = dict(begin=0.0, launch=0.0, readout=0.0, sample=0.0, stats=0.0)
N = 200
for it in range(N + 20):
    if it == 20:
        t = {k: 0.0 for k in t}
    t0 = time.perf_counter(); e.begin_search(legal, tp, True, num_legal=nl)
    t1 = time.perf_counter(); e._check(e._lib.mzmcts_search_fused_fc(e._h, obs.data_ptr(), 1, e._stream()))
    t2 = time.perf_counter(); e.readout()
    t3 = time.perf_counter(); e.sample_actions(temp)
    t4 = time.perf_counter(); e.search_statistics()
    t5 = time.perf_counter()
    t["begin"] += t1 - t0; t["launch"] += t2 - t1; t["readout"] += t3 - t2; t["sample"] += t4 - t3; t["stats"] += t5 - t4
print({k: round(1e6 * v / N, 1) for k, v in t.items()}, "us per move; threads env", os.environ.get("MZMCTS_HOST_THREADS"))

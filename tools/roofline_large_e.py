#!/usr/bin/env python3
"""HBM-scale run of the lock-step tree kernels ("a run with E scaled until node +
hidden pools exceed 512 MB").  Injected network outputs (no inference), CartPole search constants
(A=2, S=50, H=8), E trees; prints per-kernel mean duration (HIP events bound to the dispatch) and the
achieved algorithmic GB/s against the 8 TB/s HBM roofline, as one JSON line.

    python tools/roofline_large_e.py [log2_E=20] [A=2] [S=50] [H=8] [lanes per tree]
"""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from parity_helpers import make_search_config

STEP = "--fused-step" in sys.argv                     # expand_backup + next select in one launch
sys.argv = [a for a in sys.argv if a != "--fused-step"]
QUEUE = 0                                             # --queue N: trees per wavefront of select (0 = no queue)
if "--queue" in sys.argv:
    i = sys.argv.index("--queue"); QUEUE = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
log2e = int(sys.argv[1]) if len(sys.argv) > 1 else 20
A = int(sys.argv[2]) if len(sys.argv) > 2 else 2
S = int(sys.argv[3]) if len(sys.argv) > 3 else 50
H = int(sys.argv[4]) if len(sys.argv) > 4 else 8
GROUP = int(sys.argv[5]) if len(sys.argv) > 5 else 0     # lanes per tree (0 = default; 1 = one lane per tree, A = 2)
E = 1 << log2e
eng = importlib.import_module("muzero-hypermodel_amd.engine")
cfg = make_search_config(A, S, 1 if A == 2 else 2, 0.997, H=H)
t0 = time.time()
engine = eng.BatchedMCTS(cfg, E, group_width=GROUP)
engine.set_select_queue(QUEUE)
print(f"engine for E={E} built in {time.time()-t0:.1f}s, device pools {engine.device_bytes()/2**30:.2f} GiB", flush=True)
g = torch.Generator(device="cuda").manual_seed(0)
value = (torch.randn(E, generator=g, device="cuda", dtype=torch.float32) * 30).double()
reward = torch.randn(E, generator=g, device="cuda", dtype=torch.float32).double()
priors = torch.softmax(torch.randn(E, A, generator=g, device="cuda", dtype=torch.float32), dim=1).double()
root_reward = torch.zeros(E, device="cuda", dtype=torch.float64)
legal = np.tile(np.arange(A, dtype=np.int32), (E, 1)); nl = np.full(E, A, np.int32); tp = np.zeros(E, np.int32)
lib, h = engine._lib, engine._h

def move():
    engine.begin_search(legal, tp, True, num_legal=nl)
    engine._check(lib.mzmcts_expand_roots_injected(h, root_reward.data_ptr(), priors.data_ptr(), engine._stream()))
    for s in range(S):
        if s == 0 or not STEP:
            engine.select(gather=True)
        if STEP and s + 1 < S:
            engine._check(lib.mzmcts_expand_backup_select_injected(h, value.data_ptr(), reward.data_ptr(), priors.data_ptr(),
                                                                   engine.batch_hidden.data_ptr(), engine.batch_action.data_ptr(), engine._stream()))
        else:
            engine._check(lib.mzmcts_expand_backup_injected(h, value.data_ptr(), reward.data_ptr(), priors.data_ptr(), engine._stream()))
    return engine.readout()

move()
engine.set_profiling(True); engine.get_profile(reset=True)
moves = 2
for _ in range(moves):
    st = move()
torch.cuda.synchronize()
prof = engine.get_profile(reset=True)
d = prof["select_depth_sum"] / max(prof["simulations"], 1)
b = engine.algorithmic_bytes_per_simulation(d)
out = {"lanes_per_tree": engine.group_width() if GROUP != 1 else 1, "select_queue_trees": QUEUE, "E": E, "A": A, "S": S, "H": H, "mean_select_depth": d, "device_pool_GiB": engine.device_bytes() / 2**30, "kernels": {}}
for name, ms, n, per in (("select", "select_ms", "select_launches", b["select"]), ("expand_backup", "expand_backup_ms", "expand_backup_launches", b["expand_backup"])):
    us = 1e3 * prof[ms] / prof[n]
    out["kernels"][name] = {"avg_us": us, "launches": prof[n], "algorithmic_bytes_per_launch": per * E,
                            "achieved_GBs": per * E / (us * 1e-6) / 1e9, "frac_of_8TBs": per * E / (us * 1e-6) / 8e12}
if prof["step_launches"]:
    us = 1e3 * prof["step_ms"] / prof["step_launches"]
    per = b["select"] + b["expand_backup"]
    out["kernels"]["expand_backup_select"] = {"avg_us": us, "launches": prof["step_launches"], "algorithmic_bytes_per_launch": per * E,
                                              "achieved_GBs": per * E / (us * 1e-6) / 1e9, "frac_of_8TBs": per * E / (us * 1e-6) / 8e12}
# The same kernels against what the memory system sustains for their ACCESS PATTERN (stand-alone measurement,
# tools/record_size_ceiling.hip -> profiles/r02_record_size_ceiling.jsonl: ~50 G random 128-byte line reads / s, ~20 G random
# line read-modify-writes / s, whatever part of a line a request uses): requests per launch are estimated from the tree
# shape -- select: per tree the lines of its descent (0.65 per level with the first-expanded child co-located, measured on
# the CartPole traces) + hidden-state row + path / control lines; expand_backup: the distinct lines of the path it updates.
READ_LINES_PER_S, RMW_LINES_PER_S = 50e9, 20e9
sel, exb = out["kernels"]["select"], out["kernels"]["expand_backup"]
lines_per_level = 0.65 if A <= 2 else float(-(-32 * A // 128))
sel_requests = E * (d * lines_per_level + 3.0)
exb_requests = E * ((d + 1) * (0.65 if A <= 2 else 1.0))
sel["request_model"] = {"random_line_reads_per_launch": sel_requests, "ceiling_lines_per_s": READ_LINES_PER_S,
                        "time_at_ceiling_us": 1e6 * sel_requests / READ_LINES_PER_S,
                        "frac_of_request_rate_ceiling": (sel_requests / READ_LINES_PER_S) / (sel["avg_us"] * 1e-6)}
exb["request_model"] = {"random_line_read_modify_writes_per_launch": exb_requests, "ceiling_lines_per_s": RMW_LINES_PER_S,
                        "time_at_ceiling_us": 1e6 * exb_requests / RMW_LINES_PER_S,
                        "frac_of_request_rate_ceiling": (exb_requests / RMW_LINES_PER_S) / (exb["avg_us"] * 1e-6)}
out["tree_sims_per_s_kernels_only"] = E / ((out["kernels"]["select"]["avg_us"] + out["kernels"]["expand_backup"]["avg_us"]) * 1e-6)
print(json.dumps(out))
engine.close()

#!/usr/bin/env python3
"""Context for the HBM-scale numbers of the lock-step tree kernels: what does the MI355X deliver for
INDEPENDENT random 64-byte gathers (one child block per tree and level is exactly that access pattern, but
with a dependent chain on top)?  torch.index_select of 16-float rows from an 8 GiB table.

    python tools/random_gather_ceiling.py
"""
import json

import torch

rows, width = 1 << 27, 16            # 8 GiB of 64-byte rows
table = torch.empty((rows, width), dtype=torch.float32, device="cuda").normal_()
out = {}
for log2n in (20, 24):
    n = 1 << log2n
    idx = torch.randint(0, rows, (n,), device="cuda")
    torch.index_select(table, 0, idx)
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    reps = 20
    for _ in range(reps):
        got = torch.index_select(table, 0, idx)
    stop.record()
    torch.cuda.synchronize()
    sec = start.elapsed_time(stop) * 1e-3 / reps
    out[f"gather_2^{log2n}_rows"] = {"us": sec * 1e6, "read_GBs": n * 64 / sec / 1e9, "read_plus_write_GBs": 2 * n * 64 / sec / 1e9}
print(json.dumps(out))

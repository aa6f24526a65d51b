#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc run: reads every *counter_collection.csv under a directory and prints
one JSON object {kernel: {counter: sum, ..., "launches": n}}, kernels ordered by the first counter given.

    python tools/pmc_summary.py <rocprofv3 output dir> [--top 25] [--mfma] [--tail 0.3]

--tail F keeps only the last fraction F of the dispatches (by Dispatch_Id): the steady state of a run whose
first moves include MIOpen's kernel search.

--mfma adds MfmaUtil_percent = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) x 100 per kernel (the
derived counter of /opt/rocm/share/rocprofiler-sdk/counter_defs.yaml) when both counters were collected.

--sq SIMS adds the instruction-issue picture of a kernel whose every wavefront runs SIMS simulations (the fused
whole-move kernels): per wavefront and simulation the VALU / SALU / LDS / VMEM instructions issued, the wave's cycles
(SQ_WAVE_CYCLES and the SQ_WAIT_* / SQ_ACTIVE_* counters count quad-cycles: x4), cycles per instruction, and the
share of the wave's life spent waiting.  Needs SQ_WAVES, SQ_WAVE_CYCLES and the SQ_INSTS_* counters in the run."""
import csv
import glob
import json
import os
import sys
from collections import OrderedDict, defaultdict

root = sys.argv[1]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
tail = float(sys.argv[sys.argv.index("--tail") + 1]) if "--tail" in sys.argv else 1.0
sums = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
order = []
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as fh:
        rows = list(csv.DictReader(fh))
        if tail < 1.0 and rows and "Dispatch_Id" in rows[0]:
            last = max(int(r["Dispatch_Id"]) for r in rows)
            rows = [r for r in rows if int(r["Dispatch_Id"]) > last * (1.0 - tail)]
        for row in rows:
            kernel, counter = row["Kernel_Name"], row["Counter_Name"]
            if counter not in order:
                order.append(counter)
            sums[kernel][counter] += float(row["Counter_Value"])
            launches[kernel].add((path, row.get("Dispatch_Id", row.get("Correlation_Id"))))
if not order:
    sys.exit(f"no counter_collection.csv under {root}")
lead = "SQ_VALU_MFMA_BUSY_CYCLES" if "SQ_VALU_MFMA_BUSY_CYCLES" in order else order[0]
out = OrderedDict()
for kernel in sorted(sums, key=lambda k: -sums[k].get(lead, 0.0))[:top]:
    entry = OrderedDict(launches=len(launches[kernel]))
    for counter in order:
        entry[counter] = sums[kernel].get(counter, 0.0)
    if "--mfma" in sys.argv and entry.get("GRBM_GUI_ACTIVE"):
        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs; the derived counter takes its maximum
        entry["MfmaUtil_percent"] = 100.0 * entry.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (entry["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "--sq" in sys.argv and entry.get("SQ_WAVES") and entry.get("SQ_WAVE_CYCLES"):
        sims = float(sys.argv[sys.argv.index("--sq") + 1])
        waves = entry["SQ_WAVES"]
        per = lambda c: entry.get(c, 0.0) / waves / sims          # noqa: E731  per wavefront and simulation
        insts = {k: per(k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                     "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM") if k in entry}
        cycles = 4.0 * per("SQ_WAVE_CYCLES")
        sq = OrderedDict(simulations_per_wave=sims, waves_per_launch=waves / entry["launches"],
                         cycles_per_simulation=cycles,
                         instructions_per_simulation=OrderedDict((k[len("SQ_INSTS_"):].lower(), v) for k, v in insts.items()),
                         instructions_per_simulation_total=sum(insts.values()),
                         cycles_per_instruction=cycles / max(sum(insts.values()), 1e-9))
        for counter, label in (("SQ_WAIT_ANY", "waiting_share"), ("SQ_WAIT_INST_ANY", "issue_stall_share"),
                               ("SQ_WAIT_INST_LDS", "lds_issue_stall_share"), ("SQ_ACTIVE_INST_ANY", "issuing_share"),
                               ("SQ_ACTIVE_INST_VALU", "valu_active_share"), ("SQ_ACTIVE_INST_LDS", "lds_active_share")):
            if counter in entry:
                sq[label] = entry[counter] / entry["SQ_WAVE_CYCLES"]
        entry["per_wave_and_simulation"] = sq
    out[kernel[:160]] = entry
total = {c: sum(v.get(c, 0.0) for v in sums.values()) for c in order}
print(json.dumps({"tail_fraction": tail, "counters": order, "kernels_in_run": len(sums), "totals": total, "kernels": out}, indent=1))

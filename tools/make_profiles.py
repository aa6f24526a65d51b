#!/usr/bin/env python3
"""Copy / assemble the measurement summaries a GPU call left in gpurun_out/ (scratch) into profiles/ (tracked),
named per round:   python tools/make_profiles.py [r02]"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"

def have(name):
    return os.path.exists(os.path.join(G, name))

def copy(src, dst):
    if have(src):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, f"{tag}_{dst}"))
        print("  ", dst)

for w in ("cartpole", "tictactoe", "connect4", "atari84"):
    copy(f"bench_{w}.json", f"bench_{w}.json")
    copy(f"{w}_kernel_stats.csv", f"bench_{w}_kernel_stats.csv")
copy("bench_gpus2_rehearsal.json", "bench_gpus2_rehearsal.json")
copy("parity_report.json", "parity_report.json")
copy("record_size_ceiling.jsonl", "record_size_ceiling.jsonl")
copy("conv_bench.jsonl", "conv_bench.jsonl")
copy("conv_pmc.json", "conv_pmc.json")
copy("roofline_large_e.json", "roofline_large_e.json")
if have("large_e_FETCH_SIZE.json") and have("large_e_WRITE_SIZE.json") and have("roofline_large_e.json"):
    roof = json.load(open(os.path.join(G, "roofline_large_e.json")))
    out = {"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python3 tools/roofline_large_e.py 20 (separate "
                      "passes: tools/profile_r02.sh large)", "unit": "MB per launch (raw counters, uncorrected)", "kernels": {}}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = json.load(open(os.path.join(G, f"large_e_{c}.json")))
        for k, v in d["kernels"].items():
            for short in ("select", "expand_backup"):
                if f"mz::{short}_kernel" in k:
                    out["kernels"].setdefault(short, {})[c + "_MB"] = v[c] / v["launches"] / 1024.0
    for short, e in out["kernels"].items():
        alg = roof["kernels"][short]["algorithmic_bytes_per_launch"] / 1e6
        e["algorithmic_MB"] = alg
        e["traffic_over_algorithmic"] = (e.get("FETCH_SIZE_MB", 0) * 1.048576 + e.get("WRITE_SIZE_MB", 0) * 1.048576) / alg
        e["avg_us"] = roof["kernels"][short]["avg_us"]
        e["frac_of_8TBs"] = roof["kernels"][short]["frac_of_8TBs"]
    json.dump(out, open(os.path.join(P, f"{tag}_lockstep_pmc_large_e.json"), "w"), indent=1)
    print("   lockstep_pmc_large_e.json")
for w in ("connect4", "tictactoe", "atari84"):
    copy(f"{w}_mfma_pmc.json", f"{w}_mfma_pmc.json")

# SQ-counter picture of the fused kernel: the two passes (instruction counts; waits / activity) merged per kernel
if have("fused_sq_insts.json") and have("fused_sq_waits.json"):
    a = json.load(open(os.path.join(G, "fused_sq_insts.json")))
    b = json.load(open(os.path.join(G, "fused_sq_waits.json")))
    out = {"command": "rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python3 bench.py --steps 20 --warmup 5 --min-seconds 0 "
                      "--cpu-seconds 0 --profile-steps 0   (two passes: tools/profile_r02.sh fusedsq); tools/pmc_summary.py --sq 50",
           "note": "per wavefront (4 trees of 16 lanes) and simulation; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles (x4 in "
                   "cycles_per_simulation, shares are ratios of quad-cycles); one wavefront per SIMD at 4096 envs",
           "kernels": {}}
    for name, ea in a["kernels"].items():
        if "fused" not in name:
            continue
        eb = b["kernels"].get(name, {})
        short = name.split("(")[0].replace("void ", "")
        merged = dict(ea.get("per_wave_and_simulation", {}))
        for k, v in eb.get("per_wave_and_simulation", {}).items():
            if k.endswith("_share"):
                merged[k] = v
        merged["raw_counters"] = {k: v for d in (ea, eb) for k, v in d.items() if k.startswith("SQ_") or k == "launches"}
        out["kernels"][short] = merged
    json.dump(out, open(os.path.join(P, f"{tag}_fused_sq.json"), "w"), indent=1)
    print("   fused_sq.json")

# HBM-side traffic per launch (FETCH_SIZE / WRITE_SIZE passes) in the form bench.py's roofline leg reads
if have("traffic_FETCH_SIZE.json") and have("traffic_WRITE_SIZE.json"):
    out = {"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python3 bench.py --steps 20 --warmup 5 --min-seconds 0 "
                      "--cpu-seconds 0 --profile-steps 0 (separate passes: tools/profile_r02.sh traffic)",
           "workload": "cartpole_fc_4096envs_x_50sims",
           "unit": "KB per launch (raw counter as rocprofv3 reports it; uncorrected, see DESIGN.md section 5)", "counters": {}}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = json.load(open(os.path.join(G, f"traffic_{c}.json")))
        out["counters"][c] = {k.split("(")[0].replace("void ", ""): {"launches": v["launches"], "mean_KB_per_launch": v[c] / v["launches"]}
                              for k, v in d["kernels"].items() if k.startswith("void mz::") or k.startswith("mz::")}
    json.dump(out, open(os.path.join(P, f"{tag}_pmc_traffic_e4096.json"), "w"), indent=1)
    print("   pmc_traffic_e4096.json")

#!/bin/bash
# rocprofv3 runs behind profiles/r01_{connect4,tictactoe}_lockstep_kernel_stats.csv and r01_connect4_mfma_pmc.json.
# Run on the GPU box from the repo root:   bash tools/profile_lockstep.sh [stats|mfma]
# (the program goes directly after `--`; counters are collected in their own run, with --kernel-trace only)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
what=${1:-stats}
if [ "$what" = "stats" ]; then
    for spec in connect4:1024 tictactoe:4096; do
        name=${spec%%:*}
        timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/ks_$name -o $name --output-format csv -- \
            python3 "$R/tools/bench_configs.py" $spec --no-graph --warm 2 --moves 3 > "$R/gpurun_out/ks_$name.log" 2>&1
        cp "$(find /tmp/ks_$name -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/${name}_kernel_stats.csv"
    done
else
    timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace \
        -d /tmp/mfma_c4 -o c4 --output-format csv -- \
        python3 "$R/tools/bench_configs.py" connect4:1024 --no-graph --warm 2 --moves 1 > "$R/gpurun_out/mfma_run.log" 2>&1
    python3 "$R/tools/pmc_summary.py" /tmp/mfma_c4 --mfma --top 12 --tail 0.3 > "$R/gpurun_out/mfma_connect4.json"
fi

#!/bin/bash
# One gpurun call after the fused-kernel work of round 3: whole GPU suite, the CartPole bench line, its kernel stats and
# the SQ-counter passes of the fused kernel (profiles/r03_fused_sq.json, r03_bench_cartpole*).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/gpu_suite.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/gpu_suite.log
tail -4 gpurun_out/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile_r02.sh fusedsq && python tools/make_profiles.py r03 > /dev/null
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_cartpole -o cartpole --output-format csv -- \
    python3 "$R/bench.py" --workload cartpole --steps 20 --warmup 2 --min-seconds 0 --cpu-seconds 0 > "$R/gpurun_out/ks_cartpole.log" 2>&1 \
 && cp "$(find /tmp/ks_cartpole -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/cartpole_kernel_stats.csv") || exit 1
cd "$R" && python bench.py > gpurun_out/bench_cartpole.json 2> gpurun_out/bench_cartpole.err && cut -c1-220 gpurun_out/bench_cartpole.json
cp profiles/r03_fused_sq.json gpurun_out/r03_fused_sq.json
head -c 1500 gpurun_out/r03_fused_sq.json

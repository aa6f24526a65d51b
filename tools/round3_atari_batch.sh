#!/bin/bash
# One gpurun call: config #5 after mzmcts_downsample_cnn (bench line, kernel stats, root-inference timing) and the CartPole line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" && mkdir -p gpurun_out
python bench.py --workload atari84 > gpurun_out/bench_atari84.json 2> gpurun_out/bench_atari84.err && cut -c1-200 gpurun_out/bench_atari84.json
python bench.py > gpurun_out/bench_cartpole.json 2> gpurun_out/bench_cartpole.err && cut -c1-200 gpurun_out/bench_cartpole.json
python tools/root_inference_time.py atari84 32768 > gpurun_out/root_inference_atari84.json 2> /dev/null
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_atari84 -o atari84 --output-format csv -- \
    python3 "$R/bench.py" --workload atari84 --steps 4 --warmup 2 --min-seconds 0 --cpu-seconds 0 > "$R/gpurun_out/ks_atari84.log" 2>&1 \
 && cp "$(find /tmp/ks_atari84 -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/atari84_kernel_stats.csv") || exit 1
head -6 "$R/gpurun_out/atari84_kernel_stats.csv" | cut -c1-160

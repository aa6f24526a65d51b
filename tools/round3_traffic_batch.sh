#!/bin/bash
# One gpurun call: FETCH_SIZE / WRITE_SIZE passes of the fused CartPole bench (profiles/r03_pmc_traffic_e4096.json) and the
# TicTacToe kernel statistics after the root tower.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" && mkdir -p gpurun_out
bash tools/profile_r02.sh traffic || exit 1
python tools/make_profiles.py r03 > /dev/null; cp profiles/r03_pmc_traffic_e4096.json gpurun_out/r03_pmc_traffic_e4096.json
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_ttt -o tictactoe --output-format csv -- \
    python3 "$R/bench.py" --workload tictactoe --steps 8 --warmup 2 --min-seconds 0 --cpu-seconds 0 > "$R/gpurun_out/ks_tictactoe.log" 2>&1 \
 && cp "$(find /tmp/ks_ttt -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/tictactoe_kernel_stats.csv") || exit 1
head -5 gpurun_out/tictactoe_kernel_stats.csv | cut -c1-150; head -c 900 gpurun_out/r03_pmc_traffic_e4096.json

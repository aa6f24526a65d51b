#!/bin/bash
# One gpurun call: the whole GPU suite, then the four bench lines (profiles/r03_bench_*.json).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" && mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/gpu_suite.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/gpu_suite.log
tail -3 gpurun_out/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
for w in cartpole tictactoe atari84 connect4; do
  python bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err && echo "$w: $(cut -c1-150 gpurun_out/bench_$w.json)"
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log

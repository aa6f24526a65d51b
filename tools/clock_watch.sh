#!/bin/bash
# Diagnostic: the shader clock and socket power the chip holds while bench.py's timed region runs (rocm-smi samples).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" && mkdir -p gpurun_out
python bench.py --cpu-seconds 0 --min-seconds 6 ${1:+--workload $1} > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
pid=$!
sleep 2
for i in $(seq 1 30); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|Package Power" | sed -e "s/.*sclk clock level//" -e "s/.*Power (W)//" | tr '\n' ' '; echo
  sleep 0.4
done > gpurun_out/clock_samples.txt
wait $pid
cut -c1-160 gpurun_out/clock_bench.json
cat gpurun_out/clock_samples.txt

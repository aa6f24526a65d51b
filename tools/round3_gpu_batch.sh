#!/bin/bash
# One gpurun call's worth of round-3 measurements (box acquisition is charged per call: batch them).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/gpu_suite.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_suite.log
python bench.py > gpurun_out/bench_cartpole.json 2> gpurun_out/bench_cartpole.err
for w in tictactoe connect4 atari84; do
  python bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
# N = 2 rehearsals on one GPU (gloo): the N > 1 code path of every lock-step workload, two pipelined groups per rank
python bench.py --gpus 2 --workload tictactoe --envs 8192 --steps 20 --warmup 4 --bcast-every 5 --cpu-seconds 0 --profile-steps 0 > gpurun_out/bench_gpus2_tictactoe.json 2> gpurun_out/bench_gpus2_tictactoe.err
python bench.py --gpus 2 --workload connect4 --envs 1024 --steps 6 --warmup 2 --bcast-every 2 --cpu-seconds 0 --profile-steps 0 --min-seconds 0.2 > gpurun_out/bench_gpus2_connect4.json 2> gpurun_out/bench_gpus2_connect4.err
python bench.py --gpus 2 --steps 100 --bcast-every 50 --cpu-seconds 0 --profile-steps 0 > gpurun_out/bench_gpus2_cartpole.json 2> gpurun_out/bench_gpus2_cartpole.err
tail -3 gpurun_out/gpu_suite.log
for f in gpurun_out/bench_*.json; do echo "$f: $(cut -c1-160 $f)"; done
# whole self-play loop (search + env step + history filing) on device envs
python tools/selfplay_rate.py --game tictactoe --envs 65536 --moves 400 --kinds device-batch,device-pipelined-batch,device-pipelined > gpurun_out/selfplay_rate.jsonl 2> gpurun_out/selfplay_rate.err
python tools/selfplay_rate.py --game tictactoe --envs 65536 --moves 400 --kinds device-pipelined-batch --no-prefetch >> gpurun_out/selfplay_rate.jsonl 2>> gpurun_out/selfplay_rate.err
python tools/selfplay_rate.py --game connect4 --envs 8192 --moves 40 --batch 10 --kinds device-batch,device-pipelined-batch,device-pipelined >> gpurun_out/selfplay_rate.jsonl 2>> gpurun_out/selfplay_rate.err
python tools/selfplay_rate.py --game cartpole --envs 4096 --moves 200 --weights checkpoint --kinds device-batch,device >> gpurun_out/selfplay_rate.jsonl 2>> gpurun_out/selfplay_rate.err
cut -c1-200 gpurun_out/selfplay_rate.jsonl

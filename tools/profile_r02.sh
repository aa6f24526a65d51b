#!/bin/bash
# Round-2 measurement recipes.  Run on the GPU box from the repo root:   bash tools/profile_r02.sh <what>
#   bench4     the four driver-runnable bench lines + the --gpus 2 self-launch rehearsal  -> gpurun_out/bench_<w>.json
#   stats      rocprofv3 --kernel-trace --stats of each workload's bench command        -> gpurun_out/<w>_kernel_stats.csv
#   fusedsq    SQ-counter passes of the fused CartPole kernel (2 passes x 8 counters)    -> gpurun_out/fused_sq.json
#   mfma       matrix-pipe counters of the Connect4 network kernels                      -> gpurun_out/connect4_mfma_pmc.json
#   traffic    FETCH_SIZE / WRITE_SIZE passes of the fused CartPole bench                -> gpurun_out/pmc_traffic_e4096.json
# (the program goes directly after `--`; counters are collected in their own runs, with --kernel-trace only)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
what=${1:-bench4}
PY=python3
case "$what" in
bench4)
    cd "$R"
    for w in cartpole tictactoe connect4 atari84; do
        steps=100; [ $w = connect4 ] && steps=10; [ $w = atari84 ] && steps=20; [ $w = tictactoe ] && steps=40
        timeout -k 10 900 $PY bench.py --workload $w --steps $steps --warmup 3 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
        tail -c 400 gpurun_out/bench_$w.json; echo
    done
    timeout -k 10 300 $PY bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_gpus2_rehearsal.json 2> gpurun_out/bench_gpus2_rehearsal.err
    tail -c 600 gpurun_out/bench_gpus2_rehearsal.json; echo
    ;;
stats)
    cd /tmp && export TMPDIR=/tmp
    # (MIOpen searches its solvers once per convolution shape and keeps the result in the user's find db: an
    #  unprofiled run first, so that the search's candidate kernels stay out of the statistics)
    timeout -k 10 300 $PY "$R/bench.py" --workload atari84 --steps 1 --warmup 1 --min-seconds 0 --cpu-seconds 0 --profile-steps 0 > /dev/null 2>&1
    for w in cartpole tictactoe connect4 atari84; do
        steps=20; [ $w = connect4 ] && steps=2; [ $w = atari84 ] && steps=4; [ $w = tictactoe ] && steps=8
        timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_$w -o $w --output-format csv -- \
            $PY "$R/bench.py" --workload $w --steps $steps --warmup 2 --min-seconds 0 --cpu-seconds 0 > "$R/gpurun_out/ks_$w.log" 2>&1
        cp "$(find /tmp/ks_$w -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/${w}_kernel_stats.csv"
    done
    ;;
fusedsq)
    cd /tmp && export TMPDIR=/tmp
    timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM \
        --kernel-trace -d /tmp/sq_a -o a --output-format csv -- \
        $PY "$R/bench.py" --steps 20 --warmup 5 --min-seconds 0 --cpu-seconds 0 --profile-steps 0 > "$R/gpurun_out/sq_a.log" 2>&1
    timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
        --kernel-trace -d /tmp/sq_b -o b --output-format csv -- \
        $PY "$R/bench.py" --steps 20 --warmup 5 --min-seconds 0 --cpu-seconds 0 --profile-steps 0 > "$R/gpurun_out/sq_b.log" 2>&1
    $PY "$R/tools/pmc_summary.py" /tmp/sq_a --sq 50 --top 4 > "$R/gpurun_out/fused_sq_insts.json"
    $PY "$R/tools/pmc_summary.py" /tmp/sq_b --sq 50 --top 4 > "$R/gpurun_out/fused_sq_waits.json"
    ;;
mfma)
    # MFMA evidence for the network path of the lock-step configs: matrix-pipe busy cycles / instruction mix per kernel
    cd /tmp && export TMPDIR=/tmp
    for w in connect4 tictactoe atari84; do
        [ $w = atari84 ] && timeout -k 10 300 $PY "$R/bench.py" --workload atari84 --steps 1 --warmup 1 --min-seconds 0 --cpu-seconds 0 --profile-steps 0 > /dev/null 2>&1
        timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
            --kernel-trace -d /tmp/mfma_$w -o $w --output-format csv -- \
            $PY "$R/bench.py" --workload $w --groups 1 --steps 1 --warmup 1 --min-seconds 0 --cpu-seconds 0 --profile-steps 0 --no-graph > "$R/gpurun_out/mfma_run_$w.log" 2>&1
        $PY "$R/tools/pmc_summary.py" /tmp/mfma_$w --mfma --top 8 --tail 0.5 > "$R/gpurun_out/${w}_mfma_pmc.json"
    done
    ;;
large)
    # lock-step tree kernels at HBM scale (2^20 trees): per-kernel time, then FETCH_SIZE / WRITE_SIZE in their own passes
    cd "$R"
    timeout -k 10 500 $PY tools/roofline_large_e.py 20 > gpurun_out/roofline_large_e.log 2>&1
    tail -1 gpurun_out/roofline_large_e.log > gpurun_out/roofline_large_e.json
    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace -d /tmp/le_$c -o t --output-format csv -- \
            $PY "$R/tools/roofline_large_e.py" 20 > "$R/gpurun_out/le_$c.log" 2>&1
        $PY "$R/tools/pmc_summary.py" /tmp/le_$c --top 3 > "$R/gpurun_out/large_e_$c.json"
    done
    ;;
conv)
    cd "$R"
    $PY tools/conv_bench.py > gpurun_out/conv_bench.jsonl 2> gpurun_out/conv_bench.err
    tools/_bin/record_size_ceiling 8 > gpurun_out/record_size_ceiling.jsonl
    ;;
all)
    for part in bench4 stats mfma fusedsq traffic large conv; do bash "$R/tools/profile_r02.sh" $part || exit 1; done
    ;;
traffic)
    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace -d /tmp/tr_$c -o t --output-format csv -- \
            $PY "$R/bench.py" --steps 20 --warmup 5 --min-seconds 0 --cpu-seconds 0 --profile-steps 0 > "$R/gpurun_out/tr_$c.log" 2>&1
        $PY "$R/tools/pmc_summary.py" /tmp/tr_$c --top 6 > "$R/gpurun_out/traffic_$c.json"
    done
    ;;
esac

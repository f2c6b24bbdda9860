#!/bin/bash
# round 6, GPU call 3: the 5-rank bench with phase progress, counters of the evaluator in the pipeline, kernel stats of the bench, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== bench.py --gpus 5 on one device, progress on stderr"
CCSP_BENCH_ONE_DEVICE=1 timeout -k 10 700 python3 bench.py --gpus 5 --games 1024 --sims 400 --steps 4 --warmup 1 --spread-plies 14 --min-seconds 0 --fused-plies 4 --cpu-seconds 0 --config5-games 40 --config5-sims 100 --config5-timeout 500 > gpurun_out/r6c_bench5.json 2> gpurun_out/r6c_bench5.err
echo "rc=$?"; grep "config 5\|Traceback\|Error" gpurun_out/r6c_bench5.err | tail -25
bash tools/pmc_pipeline.sh r6c 2>&1 | tail -12
bash tools/gpu_session.sh r6c stats bench
echo "== session 3 done"

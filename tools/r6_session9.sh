#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_api.py -x -q -k bench 2>&1 | tail -4 | tee gpurun_out/r6i_bench_tests.txt
bash tools/gpu_session.sh r6i bench
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | grep '^{' > gpurun_out/r6i_bench_driver_like_20_steps.json
echo "== session 9 done"

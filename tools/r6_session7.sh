#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== movegen parity (persistent workgroups + register-built line patterns)"
timeout -k 10 300 python3 -m pytest tests/test_gpu_rules.py tests/test_gpu_greedy.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout -k 10 200 python3 tools/bench_movegen.py 2>&1 | tail -2; done
timeout -k 10 200 python3 tools/stamps_movegen.py 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r6g_movegen.txt
echo "== session 7 done"

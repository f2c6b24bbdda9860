#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for p in 2 3 4; do
  echo "== parts $p"
  timeout -k 10 300 python3 tools/bench_free.py --plies 16 --spread 60 --budgets 8 --parts $p --harvest 2 2>&1 | grep -v amdgpu.ids | tail -2
done 2>&1 | tee gpurun_out/r6j_parts.txt
echo "== session 10 done"

#!/bin/bash
# round 6, GPU call 5: movegen V4 variants, then the whole GPU suite on the tree as it stands
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
P=$PWD/chinesecheckersagent_amd
{
for v in mgat mgss mgatss; do
  echo "== movegen parity on $v"
  CCSP_LIB=$P/libccsp_exp_$v.so timeout -k 10 300 python3 -m pytest tests/test_gpu_rules.py -x -q 2>&1 | tail -2
done
for v in "" mgat mgss mgatss ""; do
  echo "== bench_movegen ${v:-product (origin first)}"
  if [ -n "$v" ]; then export CCSP_LIB=$P/libccsp_exp_$v.so; else unset CCSP_LIB; fi
  timeout -k 10 200 python3 tools/bench_movegen.py 2>&1 | tail -2
done
unset CCSP_LIB
} > gpurun_out/r6e_movegen.txt 2>&1
cat gpurun_out/r6e_movegen.txt
bash tools/gpu_session.sh r6e tests
echo "== session 5 done"

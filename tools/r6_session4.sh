#!/bin/bash
# round 6, GPU call 4: the N-rank fit's step on one device, movegen V3, then what call 3 did not reach (pmc in the pipeline, kernel stats, the bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
P=$PWD/chinesecheckersagent_amd
for n in 1 2 3 5; do timeout -k 10 330 python3 tools/ddp_step_probe.py $n 10 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -4; done | tee gpurun_out/r6d_ddp_step.txt
{
echo "== movegen parity on the origin-first variant"
CCSP_LIB=$P/libccsp_exp_mgof.so timeout -k 10 300 python3 -m pytest tests/test_gpu_rules.py -x -q 2>&1 | tail -3
for v in mglds "" mgof; do
  echo "== bench_movegen ${v:-product (register stacks)}"
  if [ -n "$v" ]; then export CCSP_LIB=$P/libccsp_exp_$v.so; else unset CCSP_LIB; fi
  timeout -k 10 200 python3 tools/bench_movegen.py 2>&1 | tail -2
  timeout -k 10 200 python3 tools/bench_movegen.py 2>&1 | tail -2
done
unset CCSP_LIB
} > gpurun_out/r6d_movegen.txt 2>&1
tail -16 gpurun_out/r6d_movegen.txt
bash tools/pmc_pipeline.sh r6d 2>&1 | tail -12
bash tools/gpu_session.sh r6d stats bench
echo "== session 4 done"

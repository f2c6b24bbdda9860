#!/bin/bash
# round 6, GPU call 2: movegen variants (parity + speed), A/B of the experiment builds on the headline loop, the spread sweep for the discard rate
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
P=$PWD/chinesecheckersagent_amd
echo "== bench.py --gpus 5 on one device (the test's command), stderr kept"
CCSP_BENCH_ONE_DEVICE=1 timeout -k 10 600 python3 bench.py --gpus 5 --games 1024 --sims 400 --steps 4 --warmup 1 --spread-plies 14 --min-seconds 0 --fused-plies 4 --cpu-seconds 0 --config5-games 40 --config5-sims 100 --config5-timeout 400 > gpurun_out/r6b_bench5.json 2> gpurun_out/r6b_bench5.err
echo "rc=$?"; grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/r6b_bench5.err | tail -25
python3 -c "
import json
d = json.loads([l for l in open('gpurun_out/r6b_bench5.json') if l.startswith('{')][0])
print('degraded', d['degraded'], d.get('degraded_reason'), 'config5', d.get('config5'))"
{
echo "== movegen parity (tests/test_gpu_rules.py) on the register-stack variant"
CCSP_LIB=$P/libccsp_exp_mgreg.so timeout -k 10 300 python3 -m pytest tests/test_gpu_rules.py -x -q 2>&1 | tail -3
echo "== movegen parity on the product build"
timeout -k 10 300 python3 -m pytest tests/test_gpu_rules.py -x -q 2>&1 | tail -3
for v in mgold "" mgreg; do
  echo "== bench_movegen ${v:-product (V1)}"
  if [ -n "$v" ]; then export CCSP_LIB=$P/libccsp_exp_$v.so; else unset CCSP_LIB; fi
  timeout -k 10 200 python3 tools/bench_movegen.py 2>&1 | tail -2
  timeout -k 10 200 python3 tools/bench_movegen.py 2>&1 | tail -2
done
unset CCSP_LIB
} > gpurun_out/r6b_movegen.txt 2>&1
tail -30 gpurun_out/r6b_movegen.txt
{
echo "== A/B on one box: bench.py --no-extras --steps 48 (shipped library first and last)"
bash tools/ab_bench.sh - nt1 nt2 mg2x netd80 -
} > gpurun_out/r6b_ab.txt 2>&1
tail -12 gpurun_out/r6b_ab.txt
{
echo "== discard rate of a 20-step window against the untimed spread (long-run share of whole runs: 0.192)"
for n in 72 120 160 200; do AB_STEPS=20 AB_FLAGS="--spread-plies $n" bash tools/ab_bench.sh - | sed "s/^/spread $n: /"; done
} > gpurun_out/r6b_spread.txt 2>&1
tail -8 gpurun_out/r6b_spread.txt
echo "== session 2 done"

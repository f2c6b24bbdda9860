#!/bin/bash
# Round-5 steady-state experiments on the free-running path (tools/bench_free.py): limits of a call, tree-wave issue priority (pre-built
# variants of the library: CCSP_LIB), boundary cadence.   usage (GPU box): bash tools/exp_round5.sh <tag> [what...]
# (`prio` / `lib:<file>` need the variant built BEFORE the call, in the build container -- the .so travels with the snapshot:
#  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -shared -DCCSP_ADVANCE_PRIO=1 -o chinesecheckersagent_amd/libccsp_exp_prio1.so chinesecheckersagent_amd/csrc/*.hip)
tag=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
log=gpurun_out/${tag}_exp.log
run() { echo "## $*" >> $log; timeout -k 10 ${STEP_TIMEOUT:-400} "$@" 2>&1 | grep -v amdgpu.ids >> $log; local rc=${PIPESTATUS[0]}; [ $rc -ge 124 ] && { echo "killed: stop" >> $log; exit $rc; }; return 0; }
for what in "$@"; do
  case $what in
    base)   run python3 tools/bench_free.py --budgets 8 ;;
    dbgsmall) run python3 tools/bench_free.py --budgets 8 --debug --games 256 --plies 2 --spread 4 ;;
    dbgmid) run python3 tools/bench_free.py --budgets 8 --debug --games 2048 --plies 2 --spread 4 ;;
    dbg)    run python3 tools/bench_free.py --budgets 8 --debug --plies 4 --spread 24 ;;
    limits) run python3 tools/bench_free.py --budgets 8 --time-caps 40,50,60 --deadlines 60,80,100 ;;
    prio)   for v in 0 1 3; do CCSP_LIB=$PWD/chinesecheckersagent_amd/libccsp_exp_prio$v.so run python3 tools/bench_free.py --budgets 8; done ;;
    every)  for k in 3 4 8; do run python3 tools/bench_free.py --budgets 8 --boundary-every $k; done ;;
    lib:*)  CCSP_LIB=$PWD/chinesecheckersagent_amd/${what#lib:} run python3 tools/bench_free.py --budgets 8 ;;
    small)  run python3 tools/bench_free.py --budgets 8 --games 256 --sims 800 --lockstep --plies 6 --spread 12
            run python3 tools/bench_free.py --budgets 8 --games 1024 --sims 400 --lockstep ;;
  esac
done
echo "## done" >> $log
tail -n 60 $log

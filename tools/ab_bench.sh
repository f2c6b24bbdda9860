#!/bin/bash
# Development aid: the headline loop of bench.py (config 3, no extras) with pre-built variants of the library, on ONE box.
# usage (GPU box): bash tools/ab_bench.sh <variant|-> ...      ("-" = the shipped library; variants: chinesecheckersagent_amd/libccsp_exp_<variant>.so)
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  if [ "$v" = "-" ]; then unset CCSP_LIB; else export CCSP_LIB=$PWD/chinesecheckersagent_amd/libccsp_exp_$v.so; fi
  timeout -k 10 300 python3 bench.py --no-extras --steps ${AB_STEPS:-48} --warmup 5 ${AB_FLAGS:-} 2>/dev/null | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-10s %.3f M exp/s  %.1f games/s  evaluator: %.1f us wall per launch, %.1f its own duration in the pipeline, %.1f alone; tree kernels %.1f us; idle rows %.3f; discards %.3f' % ('$v', d['value'] / 1e6, d['config']['games_per_s'], r['wall_ms_per_launch'] * 1e3, r['avg_launch_ms'] * 1e3, r['avg_launch_ms_isolated'] * 1e3, r['tree_kernels_ms_in_the_same_rounds'] * 1e3, d['config']['idle_row_share'], d['config']['discard_rate']))"
  rc=${PIPESTATUS[0]}; [ $rc -ge 124 ] && { echo "killed: stop"; exit $rc; }
done

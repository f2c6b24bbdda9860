#!/bin/bash
# round 6, GPU call 6: movegen phase stamps + counters of the rules kernels, the driver-like 20-step line, the 300-step soak
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python3 tools/stamps_movegen.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6f_movegen_stamps.txt
bash tools/pmc_round.sh r6f rules 2>&1 | tail -8
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | grep '^{' > gpurun_out/r6f_bench_driver_like_20_steps.json; echo "driver-like rc=$?"
timeout -k 10 400 python3 bench.py --steps 300 --no-extras 2>/dev/null | grep '^{' > gpurun_out/r6f_soak.json; echo "soak rc=$?"
python3 - <<'PY'
import json
for f in ('gpurun_out/r6f_bench_driver_like_20_steps.json', 'gpurun_out/r6f_soak.json'):
    d = json.load(open(f)); c = d['config']; r = d['roofline']
    print(f, 'steps', d['steps'], '%.2f M exp/s' % (d['value'] / 1e6), 'games/s %.1f won %.1f' % (c['games_per_s'], c['games_won_per_s']), 'discards %.3f' % c['discard_rate'],
          'idle %.3f' % c['idle_row_share'], 'frac %.3f by_step %.3f useful %.3f isolated %.3f' % (r['frac'], r['frac_by_step'], r['useful_frac_by_step'], r['frac_isolated']), 'errors', d['errors'])
PY
echo "== session 6 done"

#!/bin/bash
# The small-batch regime under rocprofv3 --kernel-trace (round 5, verdict item 6): per simulation step of a batch that does not fill the
# GPU -- evaluator launch, tree kernel(s), the gaps between them.   usage (GPU box): bash tools/trace_small.sh <tag> <games> <sims> [bench_free args]
tag=$1; games=$2; sims=$3; shift 3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/ps_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/ps_$tag -- python3 tools/bench_free.py --games $games --sims $sims --plies 3 --spread 8 --budgets 8 "$@" > gpurun_out/${tag}_small.log 2>&1
python3 - "$(find /tmp/ps_$tag -name '*kernel_trace.csv' | head -1)" > gpurun_out/${tag}_small_trace.txt <<'PY'
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    k = ('net' if 'net_forward' in n else 'adv' if 'advance_kernel' in n else 'bnd' if 'boundary_kernel' in n else 'ebs' if 'expand_backup_select' in n
         else 'eb' if 'expand_backup_kernel' in n else 'sel' if 'select_kernel' in n else 'ply' if ('ply_' in n or 'root_expand' in n) else 'other')
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k))
rows.sort()
rows = rows[len(rows) * 2 // 3:]                   # the last third: steady state, graphs captured
t0, t1 = rows[0][0], rows[-1][1]
d = collections.defaultdict(list)
for s, e, k in rows: d[k].append((e - s) / 1e3)
print('window %.2f ms, %d kernels' % ((t1 - t0) / 1e6, len(rows)))
for k, v in sorted(d.items()):
    v.sort(); print('%-5s n %6d  sum %8.2f ms (%4.1f %% of the window)  avg %.1f us  p50 %.1f  p90 %.1f' % (k, len(v), sum(v) / 1e3, 100 * sum(v) * 1e3 / (t1 - t0), sum(v) / len(v), v[len(v) // 2], v[int(len(v) * .9)]))
busy = 0; end = rows[0][0]
gaps = []
for s, e, k in rows:
    if s > end: gaps.append((s - end) / 1e3)
    busy += max(0, e - max(s, end)); end = max(end, e)
gaps.sort()
print('device busy %.1f %% of the window; %d gaps between kernels: avg %.1f us  p50 %.1f  p90 %.1f  sum %.2f ms' % (100.0 * busy / (t1 - t0), len(gaps), sum(gaps) / max(len(gaps), 1), gaps[len(gaps) // 2] if gaps else 0, gaps[int(len(gaps) * .9)] if gaps else 0, sum(gaps) / 1e3))
nets = [(s, e) for s, e, k in rows if k == 'net']
if len(nets) > 10:
    per = [(nets[i + 1][0] - nets[i][0]) / 1e3 for i in range(len(nets) - 1)]
    per = sorted(x for x in per if x < 2000)
    print('evaluator launch to evaluator launch: avg %.1f us  p50 %.1f  p90 %.1f' % (sum(per) / len(per), per[len(per) // 2], per[int(len(per) * .9)]))
print('a sample of the timeline (start us, duration us, kernel):')
for s, e, k in rows[len(rows) // 2:len(rows) // 2 + 24]: print('%10.1f %8.1f %s' % ((s - t0) / 1e3, (e - s) / 1e3, k))
PY
tail -2 gpurun_out/${tag}_small.log | cut -c1-500; cat gpurun_out/${tag}_small_trace.txt

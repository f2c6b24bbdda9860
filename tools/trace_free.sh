#!/bin/bash
# kernel durations of the free-running path under rocprofv3 (development aid): bash tools/trace_free.sh <tag> <bench_free args...>
tag=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/pf_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$tag -- python3 tools/bench_free.py "$@" > gpurun_out/${tag}_free.log 2>&1
f=$(find /tmp/pf_$tag -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 - "$(find /tmp/pf_$tag -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    k = 'net' if 'net_forward' in n else ('adv' if 'advance_kernel' in n else ('bnd' if 'boundary_kernel' in n else ('ebs' if 'expand_backup_select' in n else None)))
    if k: rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k, r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
rows = rows[len(rows) // 2:]                       # steady state: the second half of the trace
import collections
d = collections.defaultdict(list)
for s, e, k, q in rows: d[k].append((e - s) / 1e3)
for k, v in d.items():
    v.sort(); print(k, 'n', len(v), 'avg %.1f us  p50 %.1f  p90 %.1f  max %.1f' % (sum(v) / len(v), v[len(v) // 2], v[int(len(v) * .9)], v[-1]))
t0, t1 = rows[0][0], rows[-1][1]
busy_net = sum(e - s for s, e, k, q in rows if k == 'net')
print('window %.1f ms; net busy %.1f %% (sum of net durations / window)' % ((t1 - t0) / 1e6, 100.0 * busy_net / (t1 - t0)))
adv = [(e - s) / 1e3 for s, e, k, q in rows if k == 'adv']
print('adv durations in launch order (us):', ' '.join('%.0f' % x for x in adv[3000:3200]))
bnd = [(e - s) / 1e3 for s, e, k, q in rows if k == 'bnd']
print('bnd durations in launch order (us):', ' '.join('%.0f' % x for x in bnd[3000:3200]))
# per half-batch (the two parts' launches alternate in the serialised trace): mean of every 10 consecutive advance launches
for half in (0, 1):
    seq = adv[half::2][:3000]
    print('adv half %d, means of 10 launches (us):' % half, ' '.join('%.0f' % (sum(seq[i:i + 10]) / len(seq[i:i + 10])) for i in range(0, len(seq), 10)))
# a sample of the timeline
for s, e, k, q in rows[2000:2040]: print('%8.1f %8.1f %s q=%s' % ((s - t0) / 1e3, (e - s) / 1e3, k, q))
PY
tail -3 gpurun_out/${tag}_free.log | cut -c1-400

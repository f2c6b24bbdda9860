#!/bin/bash
# One gpurun call = a list of steps, each under its own timeout; a step that is killed (exit >= 124) ends the session
# (no further GPU step after a hang), an ordinary failure does not.   usage: bash tools/gpu_session.sh <tag> step...
# steps: tests | bench | pmc[:what] | netexp:<flags,comma-separated> | stamps_net[:flags] | stamps | cmd:<shell command>
tag=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { # name seconds command...
  local name=$1 secs=$2; shift 2
  echo "=== $name ($(date +%T))"
  timeout -k 10 "$secs" "$@" > gpurun_out/${tag}_$name.log 2>&1
  local rc=$?
  echo "=== $name rc=$rc"; tail -n 12 gpurun_out/${tag}_$name.log
  if [ $rc -ge 124 ]; then echo "step $name was killed: session ends"; exit $rc; fi
  return 0
}
for step in "$@"; do
  case $step in
    tests) run tests 900 python3 -m pytest tests -m gpu -x -q ;;
    tests_all) run tests 900 python3 -m pytest tests -m gpu -q ;;
    bench) run bench 600 python3 bench.py; grep '^{' gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json ;;
    stats) rm -rf /tmp/st_$tag; run stats 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -- python3 bench.py --steps 20 --warmup 5 --spread-plies 16 --no-extras
           grep '^{' gpurun_out/${tag}_stats.log > gpurun_out/${tag}_bench_under_rocprof.json
           f=$(find /tmp/st_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv
           f=$(find /tmp/st_$tag -name "*kernel_trace.csv" | head -1)      # every launch of the multi-ply kernel with its own duration
           [ -n "$f" ] && python3 - "$f" > gpurun_out/${tag}_launches.csv <<'PY'
import csv, sys
print('kernel,start_ns,duration_ms')
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'fused_plies_kernel' in n:
        print('fused_plies_kernel,%s,%.3f' % (r['Start_Timestamp'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0)))
rows.sort()
# net_forward_kernel.  (1) THE ISOLATED BURST: bench.py's net_kernel_alone -- 100 + 400 launches of the request form (template argument
# `true`) back to back on one stream after the run has been closed: the longest run of CONSECUTIVE trace entries that are all such
# launches (nothing else starts in between: nothing else is on the device).  (2) the launches of the timed pipelines: the request-form
# launches of the full grid that are NOT in that run.
is_req = lambda n: 'net_forward_kernel' in n and 'true' in n.split('net_forward_kernel')[-1]
best, cur = (0, 0), None
for i, (s, e, n, g) in enumerate(rows + [(0, 0, '', 0)]):
    if is_req(n):
        cur = i if cur is None else cur
    else:
        if cur is not None and i - cur > best[1] - best[0]: best = (cur, i)
        cur = None
burst = rows[best[0]:best[1]]
if len(burst) >= 300:
    d = [e - s for s, e, n, g in burst][-400:]
    gaps = [burst[i + 1][0] - burst[i][1] for i in range(len(burst) - 1)][-399:]
    print('net_forward_kernel,isolated_burst_%d_consecutive_launches_last_%d_avg_ms,%.6f' % (len(burst), len(d), sum(d) / len(d) / 1e6))
    print('net_forward_kernel,isolated_burst_gap_between_launches_avg_us,%.3f' % (sum(gaps) / len(gaps) / 1e3))
    print('net_forward_kernel,isolated_burst_start_to_start_avg_ms,%.6f' % ((burst[-1][0] - burst[-400][0]) / 399 / 1e6))
inb = set(range(best[0], best[1]))
gmax = max([g for s, e, n, g in rows if is_req(n)] or [0])
pipe = [(s, e) for i, (s, e, n, g) in enumerate(rows) if is_req(n) and g == gmax and i not in inb]
if len(pipe) > 900:
    d = [e - s for s, e in pipe]
    print('net_forward_kernel,in_pipeline_%d_launches_avg_ms,%.6f' % (len(d), sum(d) / len(d) / 1e6))
    gaps = [pipe[i + 1][0] - pipe[i][1] for i in range(len(pipe) - 1)]
    gaps = [x for x in gaps if -200000 < x < 200000]   # (neighbours of the same run of plies)
    print('net_forward_kernel,in_pipeline_gap_between_consecutive_launches_avg_us,%.3f' % (sum(gaps) / len(gaps) / 1e3))
    print('net_forward_kernel,in_pipeline_gap_median_us,%.3f' % (sorted(gaps)[len(gaps) // 2] / 1e3))
    st = [pipe[i + 1][0] - pipe[i][0] for i in range(len(pipe) - 1)]
    st = sorted(x for x in st if 0 < x < 400000)
    print('net_forward_kernel,in_pipeline_start_to_start_median_ms,%.6f' % (st[len(st) // 2] / 1e6))
    over = [min(pipe[i][1], pipe[i + 1][1]) - pipe[i + 1][0] for i in range(len(pipe) - 1)]
    over = [x for x in over if -200000 < x < 200000]
    print('net_forward_kernel,in_pipeline_overlap_with_the_next_launch_avg_us,%.3f' % (sum(max(x, 0) for x in over) / len(over) / 1e3))
for key in ('advance_kernel', 'boundary_kernel'):
    d = [e - s for s, e, n, g in rows if key in n and g >= 1024]
    if d: print('%s,%d_launches_of_a_half_batch_avg_ms,%.6f' % (key, len(d), sum(d) / len(d) / 1e6))
PY
           ;;
    pmc*) w=${step#pmc}; w=${w#:}; run pmc 1100 bash tools/pmc_round.sh $tag ${w:-all} ;;
    netexp*) f=${step#netexp}; f=${f#:}; run netexp_$(echo "$f" | tr -c 'A-Za-z0-9\n' '_') 300 python3 tools/bench_net.py ${f//,/ } ;;
    stamps_net*) f=${step#stamps_net}; f=${f#:}; run stamps_net_$(echo "$f" | tr -c 'A-Za-z0-9\n' '_') 300 python3 tools/stamps_net.py ${f//,/ } ;;
    stamps) run stamps 300 python3 tools/stamps.py ;;
    cmd:*) run cmd 600 bash -c "${step#cmd:}" ;;
  esac
done
echo "=== session done"

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
    stats) rm -rf /tmp/st_$tag; run stats 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -- python3 bench.py --steps 20 --warmup 5 --spread-plies 16 --fused-plies 32 --cpu-seconds 0 --no-config5
           grep '^{' gpurun_out/${tag}_stats.log > gpurun_out/${tag}_bench_under_rocprof.json
           f=$(find /tmp/st_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv
           f=$(find /tmp/st_$tag -name "*kernel_trace.csv" | head -1)      # every launch of the multi-ply kernel with its own duration
           [ -n "$f" ] && python3 - "$f" > gpurun_out/${tag}_launches.csv <<'PY'
import csv, sys
print('kernel,start_ns,duration_ms')
net = []
for r in csv.DictReader(open(sys.argv[1])):
    if 'fused_plies_kernel' in r['Kernel_Name']:
        print('fused_plies_kernel,%s,%.3f' % (r['Start_Timestamp'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
    if 'net_forward_kernel' in r['Kernel_Name']:
        net.append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0)))
# net_forward_kernel: the launches of the timed region sit in hipGraphs beside the other half-batch's tree kernels; the LAST 400
# launches of the trace are bench.py's back-to-back burst (the figure `roofline.avg_launch_ms` reports)
net.sort()
gmax = max([g for _, _, g in net] or [0])
big = [d for _, d, g in net if g == gmax]
full = [(st, d) for st, d, g in net if g == gmax]
if len(full) > 900:
    pp = full[:-500]                                   # the launches inside the timed pipelines, in start order
    gaps = [pp[i + 1][0] - (pp[i][0] + pp[i][1]) for i in range(len(pp) - 1)]
    gaps = [x for x in gaps if -200000 < x < 200000]   # (neighbours of the same run of plies)
    print('net_forward_kernel,in_pipeline_gap_between_consecutive_launches_avg_us,%.3f' % (sum(gaps) / len(gaps) / 1e3))
    print('net_forward_kernel,in_pipeline_gap_median_us,%.3f' % (sorted(gaps)[len(gaps) // 2] / 1e3))
if len(big) > 900:
    burst, pipe = big[-400:], big[:-500]
    print('net_forward_kernel,back_to_back_burst_last_400_avg_ms,%.6f' % (sum(burst) / len(burst) / 1e6))
    print('net_forward_kernel,in_pipeline_%d_launches_avg_ms,%.6f' % (len(pipe), sum(pipe) / len(pipe) / 1e6))
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

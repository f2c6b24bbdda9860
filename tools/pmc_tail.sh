#!/bin/bash
# counters of the LAST launches of one kernel of the free-running path in steady state (development aid)
# usage (GPU box): bash tools/pmc_tail.sh <tag> <kernel substring> <steps> <counter> [<counter> ...]
tag=$1; kern=$2; steps=$3; shift 3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/pt_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/pt_$tag -- python3 tools/kernels_once.py free $steps > /tmp/pt_$tag.log 2>&1
echo "rocprofv3 rc=$?"; tail -2 /tmp/pt_$tag.log | cut -c1-300
python3 - "$(find /tmp/pt_$tag -name '*counter_collection.csv' | head -1)" "$kern" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        rows[r['Counter_Name']].append((int(r['Start_Timestamp']), float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for c, v in rows.items():
    v.sort(); last = v[-400:]
    print('%-24s last %d launches: mean %.4g per launch = %.1f per wave (2048 waves); launch %.1f us under pmc' % (c, len(last), sum(x[1] for x in last) / len(last), sum(x[1] for x in last) / len(last) / 2048, sum(x[2] for x in last) / len(last) / 1e3))
PY

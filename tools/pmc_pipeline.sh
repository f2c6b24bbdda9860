#!/bin/bash
# Counter passes of net_forward_kernel AS IT RUNS IN THE FREE-RUNNING PIPELINE beside the same kernel alone, in ONE process per pass:
# tools/kernels_once.py `pipe` = 1300 rounds of [evaluator -> advance | boundary] on two streams (plain launches: no hipGraph under --pmc),
# then 60 back-to-back evaluator launches on the run's own request records with nothing else on the device.
# usage (GPU box): bash tools/pmc_pipeline.sh <tag>   -> gpurun_out/<tag>_pmc_pipeline.json
tag=${1:-rX}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
sets=("TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
      "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE")
i=0
for set in "${sets[@]}"; do
  i=$((i+1)); d=/tmp/pmcp_${tag}_$i; rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $d -- python3 tools/kernels_once.py pipe 900 > $d.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $i failed (rc $rc)"; tail -3 $d.log; [ $rc -ge 124 ] && { echo "a pass was killed: no further passes"; break; }; fi
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" /tmp/pmcp_${tag}_pass$i.csv
  echo "pass $i ($set): $(wc -l < /tmp/pmcp_${tag}_pass$i.csv 2>/dev/null) rows"
done
python3 - "$tag" <<'PY'
import csv, sys, glob, json, collections
tag = sys.argv[1]
out = {'net_forward_kernel<8,8,REQ>': {'in_pipeline': {}, 'alone': {}}, 'advance_kernel': {}, 'boundary_kernel': {}, 'note':
       'per launch, mean; in_pipeline = the last third of the 2048-position request-form launches issued between advance / boundary launches on two '
       'streams (plain launches), alone = the last 40 of a back-to-back burst of the same launches after the run; launch_us = the launch\'s duration '
       'under the counter pass; overlap_us = how far a pipeline launch overlapped the previous kernel of the OTHER stream (0 = the pass serialised them)'}
for f in sorted(glob.glob('/tmp/pmcp_%s_pass*.csv' % tag)):
    rows = []
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '')
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k, r['Counter_Name'], float(r['Counter_Value']), int(r['Grid_Size']), int(r.get('Dispatch_Id') or 0)))
    rows.sort(key=lambda x: (x[6], x[0]))
    disp = collections.OrderedDict()
    for s, e, k, c, v, g, did in rows:
        disp.setdefault(did, dict(s=s, e=e, k=k, g=g, c={}))['c'][c] = v
    seq = list(disp.values())
    is_net = lambda d: 'net_forward_kernel' in d['k'] and 'true' in d['k'].split('net_forward_kernel')[-1] and d['g'] >= 256 * 512
    # the burst: the trailing run of consecutive evaluator launches
    j = len(seq)
    while j > 0 and is_net(seq[j - 1]):
        j -= 1
    burst, before = seq[j:], seq[:j]
    pipe = [d for d in before if is_net(d)]
    pipe = pipe[len(pipe) * 2 // 3:]
    alone = burst[-40:]
    def put(dst, ds):
        if not ds: return
        for c in ds[0]['c']:
            dst[c] = sum(d['c'][c] for d in ds) / len(ds)
            dst.setdefault('launch_us', {})[c] = sum(d['e'] - d['s'] for d in ds) / len(ds) / 1e3
            dst.setdefault('launches', {})[c] = len(ds)
    put(out['net_forward_kernel<8,8,REQ>']['in_pipeline'], pipe)
    put(out['net_forward_kernel<8,8,REQ>']['alone'], alone)
    for key in ('advance_kernel', 'boundary_kernel'):
        ds = [d for d in before if key in d['k'] and d['g'] >= 1024 * 64]
        put(out[key], ds[len(ds) * 2 // 3:])
    # did the pass serialise the two streams?  overlap of each pipeline evaluator launch with whatever started before it and was still running
    ov = []
    idx = {id(d): i for i, d in enumerate(before)}
    for d in pipe:
        i = idx[id(d)]
        prev_end = max((p['e'] for p in before[max(0, i - 6):i]), default=d['s'])
        ov.append(max(0, prev_end - d['s']))
    if ov:
        out['net_forward_kernel<8,8,REQ>']['in_pipeline'].setdefault('overlap_us', {})[f.split('pass')[-1]] = sum(ov) / len(ov) / 1e3
json.dump(out, open('gpurun_out/%s_pmc_pipeline.json' % tag, 'w'), indent=1)
n = out['net_forward_kernel<8,8,REQ>']
for k in ('in_pipeline', 'alone'):
    print(k, {c: ('%.5g' % v) for c, v in n[k].items() if isinstance(v, float)}, 'launch_us', {c: round(v, 1) for c, v in n[k].get('launch_us', {}).items()})
print('overlap_us', n['in_pipeline'].get('overlap_us'))
PY

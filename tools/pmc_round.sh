#!/bin/bash
# Counter passes for every kernel of the library (separate rocprofv3 runs: --pmc with --kernel-trace only).
# usage (GPU box): bash tools/pmc_round.sh <tag> [what]    -> gpurun_out/<tag>_pmc.json (+ the raw per-pass CSV rows)
tag=${1:-rX}; what=${2:-all}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
sets=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES"
      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
      "TCC_HIT_sum TCC_MISS_sum"
      "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU")      # lane utilisation: thread-cycles / (instruction-cycles x 64)
i=0
for set in "${sets[@]}"; do
  i=$((i+1)); d=/tmp/pmc_${tag}_$i; rm -rf $d
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $d -- python3 tools/kernels_once.py $what > $d.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $i failed (rc $rc)"; tail -3 $d.log; [ $rc -ge 124 ] && { echo "a pass was killed: no further passes"; break; }; fi
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_pmc_pass$i.csv
  echo "pass $i ($set): $(wc -l < gpurun_out/${tag}_pmc_pass$i.csv 2>/dev/null) rows"
done
python3 - "$tag" <<'PY'
import csv, sys, collections, glob, json
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
meta = {}
for f in sorted(glob.glob('gpurun_out/%s_pmc_pass*.csv' % tag)):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[1].split(')')[-1] if False else row['Kernel_Name']
        k = k.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
        dur[(k, row['Counter_Name'])].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
        meta[k] = dict(vgpr=int(row['VGPR_Count']), sgpr=int(row['SGPR_Count']), lds=int(row['LDS_Block_Size']), wg=int(row['Workgroup_Size']), grid=int(row['Grid_Size']))
out = {}
for k, cs in acc.items():
    if k.startswith('at::') or k.startswith('__amd') or 'elementwise' in k or 'reduce_kernel' in k:
        continue
    o = dict(meta[k])
    for c, vs in cs.items():
        vs2 = vs[len(vs) // 2:] if len(vs) > 3 else vs          # later launches: warm
        o[c] = sum(vs2) / len(vs2)
        ds = dur[(k, c)][len(vs) // 2:] if len(vs) > 3 else dur[(k, c)]
        o.setdefault('launch_ns_under_pmc', {})[c] = sum(ds) / len(ds)
        o.setdefault('launches', {})[c] = len(vs)
    out[k] = o
json.dump(out, open('gpurun_out/%s_pmc.json' % tag, 'w'), indent=1)
for k, o in out.items():
    print(k, {c: ('%.4g' % v) for c, v in o.items() if isinstance(v, float)})
PY
# the aggregate the profiles/ files are made from (largest-grid launches only), then the raw per-pass rows go: they exceed what gpurun copies back
python3 tools/pmc_aggregate.py "$tag" gpurun_out/${tag}_pmc_agg.json > gpurun_out/${tag}_pmc_agg.txt 2>&1
rm -f gpurun_out/${tag}_pmc_pass*.csv

"""Aggregate the rocprofv3 --pmc passes of tools/pmc_round.sh (gpurun_out/<tag>_pmc_pass*.csv) per kernel: the launches
with the LARGEST grid of each kernel (the benchmark-sized ones; set-up code launches the same kernels on small arrays),
mean counter value per launch.  usage: python tools/pmc_aggregate.py <tag> [out.json]"""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
rows = collections.defaultdict(list)          # (kernel, counter) -> [(grid, value, ns, meta)]
for f in sorted(glob.glob('gpurun_out/%s_pmc_pass*.csv' % tag)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        if k.startswith('at::') or k.startswith('__amd') or 'elementwise' in k or 'reduce_kernel' in k:
            continue
        meta = dict(vgpr=int(r['VGPR_Count']), sgpr=int(r['SGPR_Count']), lds_bytes=int(r['LDS_Block_Size']), workgroup=int(r['Workgroup_Size']))
        rows[(k, r['Counter_Name'])].append((int(r['Grid_Size']), float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), meta))
out = {}
for (k, c), vs in sorted(rows.items()):
    g = max(v[0] for v in vs)
    big = [v for v in vs if v[0] == g]
    if len(big) > 2:
        big = big[1:]                          # the first launch of a size is cold
    if len(big) > 300:
        big = big[len(big) * 2 // 3:]          # a long run (tools/kernels_once.py `free`): its last third = the steady state
    o = out.setdefault(k, dict(big[0][3], grid_threads=g))
    o[c] = sum(v[1] for v in big) / len(big)
    o.setdefault('launches', {})[c] = len(big)
    o.setdefault('launch_us_under_pmc', {})[c] = sum(v[2] for v in big) / len(big) / 1e3
json.dump(out, open(sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/%s_pmc.json' % tag, 'w'), indent=1)
for k, o in out.items():
    print(k, 'grid', o['grid_threads'], {c: '%.4g' % v for c, v in o.items() if isinstance(v, float)})

"""Development aid: from a rocprofv3 --kernel-trace CSV of the stepped path, how much of the wall time the evaluator launches cover and
what sits between consecutive ones.   usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    k = 'net' if 'net_forward_kernel' in n else ('tree' if ('select_kernel' in n or 'expand_backup' in n) else None)
    if k:
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k))
rows.sort()
nets = [r for r in rows if r[2] == 'net']
# the densest stretch: the last 2000 evaluator launches
nets = nets[-2000:]
t0, t1 = nets[0][0], nets[-1][1]
cov = sum(e - s for s, e, _ in nets)
gaps = [nets[i + 1][0] - nets[i][1] for i in range(len(nets) - 1)]
gaps.sort()
tree = [r for r in rows if r[2] == 'tree' and t0 <= r[0] <= t1]
print('evaluator launches %d over %.3f ms: covered %.1f %%, mean duration %.1f us' % (len(nets), (t1 - t0) / 1e6, 100.0 * cov / (t1 - t0), cov / len(nets) / 1e3))
print('gap between consecutive evaluator launches (start - previous end), us: min %.2f  median %.2f  p90 %.2f  max %.2f  (negative = overlap)' %
      (gaps[0] / 1e3, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3))
print('tree launches in the stretch %d, mean duration %.2f us' % (len(tree), sum(e - s for s, e, _ in tree) / max(len(tree), 1) / 1e3))

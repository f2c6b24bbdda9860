"""Development aid: VGPR / SGPR / scratch / static LDS of every kernel of libccsp.so as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage with the flags of chinesecheckersagent_amd/build.py).   usage: python tools/isa_resources.py > profiles/rN_isa_resources.txt"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chinesecheckersagent_amd import build as B

print('kernel resources of libccsp.so (hipcc %s + per-file flags of build.py; -Rpass-analysis=kernel-resource-usage)' % ' '.join(B.FLAGS))
print('%-64s %5s %5s %8s %8s' % ('kernel', 'vgpr', 'sgpr', 'scratch', 'lds'))
for f in B.SOURCES:
    cmd = ['hipcc'] + B.FLAGS + B.EXTRA.get(f, []) + ['-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(B.CSRC, f), '-o', '/dev/null']
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    print('\n# ' + f)
    cur = {}
    for line in out.splitlines():
        m = re.search(r'remark: +(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\S+)', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            cur = {'name': v}
        else:
            cur[k] = v
        if k.startswith('LDS Size'):
            name = subprocess.run(['c++filt', cur['name']], capture_output=True, text=True).stdout.strip()
            name = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
            print('%-64s %5s %5s %8s %8s' % (name, cur.get('VGPRs'), cur.get('TotalSGPRs'), cur.get('ScratchSize [bytes/lane]'), cur.get('LDS Size [bytes/block]')))
    if f == 'ccsp_net.hip':
        print('   (net_forward_kernel: dynamic LDS, sizeof(Smem<Cfg>): see the static_asserts in ccsp_net.hip)')

"""Build libccsp.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m chinesecheckersagent_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the PUCT arithmetic and the
draw-substitution samplers must round like the reference's float64 Python/NumPy code (no fused
multiply-add); see DESIGN.md.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libccsp.so')
SOURCES = ['ccsp_rules_kernels.hip', 'ccsp_engine.hip', 'ccsp_net.hip', 'ccsp_host.hip']
HEADERS = ['ccsp_rules.h', 'ccsp_common.h', os.path.join('..', '..', 'include', 'ccsp.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-std=c++17']
# per-file flags.  ccsp_net.hip: the machine scheduler's max-ILP strategy interleaves the MFMA chains of a layer's tiles with the LDS / L2
# reads of the next k-block more tightly -- A/B on one box, round 5: <8,8> 117.1 -> 116.2 us per 2048 positions, <4,4> 73.2 -> 72.4,
# <1,8> unchanged, <2,8> 44.2 -> 45.5 (the 257..512-position shape pays); scheduling only: the arithmetic and its order are the same
EXTRA = {'ccsp_net.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp']}
OBJDIR = os.path.join(HERE, 'build')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS) or os.path.getmtime(os.path.abspath(__file__)) > t


def build(force=False, verbose=False):
    """every source to its own object (in parallel: four hipcc processes), then one link"""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    os.makedirs(OBJDIR, exist_ok=True)
    procs = []
    for f in SOURCES:
        obj = os.path.join(OBJDIR, f.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + EXTRA.get(f, []) + ['-c', os.path.join(CSRC, f), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((subprocess.Popen(cmd), cmd, obj))
    objs = []
    for p, cmd, obj in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        objs.append(obj)
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))

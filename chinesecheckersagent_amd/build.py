"""Build libccsp.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m chinesecheckersagent_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the PUCT arithmetic and the
draw-substitution samplers must round like the reference's float64 Python/NumPy code (no fused
multiply-add); see DESIGN.md.
"""
import fcntl
import os
import shutil
import subprocess
import sys
import tempfile

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
OBJDIR = os.path.join(HERE, 'build')       # the objects of the last good build (tools/build_variant.sh links its variants against them)


def needs_build():
    """a library older than one of its sources (or missing).  This file's own mtime does not count: a box that received a pre-built
    library without preserved mtimes must not try to rebuild it."""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """every source to its own object (in parallel: four hipcc processes), then one link.  Ranks that import at the same moment serialise
    on a file lock; the compile runs in a directory of this process's own and the library is put in place by ONE atomic rename, so a
    reader never sees a half-written file and a failed compile leaves the old library (and the old objects) alone."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    if shutil.which(hipcc) is None:
        if os.path.exists(LIB):                             # (a box without the toolchain: what travelled is what runs)
            sys.stderr.write('chinesecheckersagent_amd.build: %s looks older than its sources but hipcc is not here: using it as it is\n' % LIB)
            return LIB
        raise RuntimeError('libccsp.so is missing and hipcc is not on PATH: build it where the ROCm toolchain is '
                           '(python -m chinesecheckersagent_amd.build)')
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(OBJDIR, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():                 # another process built it while this one waited
            return LIB
        work = tempfile.mkdtemp(prefix='ccsp-build-', dir=OBJDIR)
        try:
            procs = []
            for f in SOURCES:
                obj = os.path.join(work, f.replace('.hip', '.o'))
                cmd = [hipcc] + FLAGS + EXTRA.get(f, []) + ['-c', os.path.join(CSRC, f), '-o', obj]
                if verbose:
                    print(' '.join(cmd))
                procs.append((subprocess.Popen(cmd), cmd, obj))
            failed = None
            for p, cmd, obj in procs:                       # every child is waited for before anything is raised
                if p.wait() != 0 and failed is None:
                    failed = (p.returncode, cmd)
                    for q, _, _ in procs:
                        if q.poll() is None:
                            q.terminate()
            if failed:
                raise subprocess.CalledProcessError(failed[0], failed[1])
            out = os.path.join(work, 'libccsp.so')
            cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', out] + [obj for _, _, obj in procs]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
            for _, _, obj in procs:
                os.replace(obj, os.path.join(OBJDIR, os.path.basename(obj)))
            os.replace(out, LIB)
        finally:
            shutil.rmtree(work, ignore_errors=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))

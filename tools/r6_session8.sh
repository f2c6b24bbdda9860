#!/bin/bash
# round 6, final GPU call: the whole GPU suite, smoke(), the bench line, a whole-run soak (tail included) of the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_session.sh r6h tests
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r6h_smoke.txt
bash tools/gpu_session.sh r6h bench
timeout -k 10 300 python3 tools/soak_generate.py 12000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6h_soak_generate.txt
echo "== session 8 done"

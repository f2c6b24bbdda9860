#!/bin/bash
# One counter pass: usage (GPU box): bash tools/pmc_once.sh <tag> <what: rules|net|fused> <counter> [<counter> ...]  -> gpurun_out/<tag>_pmc_once.txt
tag=$1; what=$2; shift 2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/po_$tag
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/po_$tag -- python3 tools/kernels_once.py $what > /tmp/po_$tag.log 2>&1
echo "rocprofv3 rc=$?"; tail -3 /tmp/po_$tag.log
f=$(find /tmp/po_$tag -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_pmc_pass1.csv && python3 tools/pmc_aggregate.py $tag gpurun_out/${tag}_pmc_once.json | cut -c1-600

#!/bin/bash
# PMC passes of scripts/kernel_table.py (every kernel family at the C4 shape); usage: prof_kernel_table.sh <tag>
set -o pipefail
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/scripts/kernel_table.py > $OUT/fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/fetch.log; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/scripts/kernel_table.py > $OUT/write.log 2>&1 || { echo write failed; tail -5 $OUT/write.log; }
du -sh $OUT

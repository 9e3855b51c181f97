#!/bin/bash
# rocprofv3 passes for one GCN layer (aggregation + MFMA GEMM): kernel trace, then MFMA counters.
set -o pipefail
TAG=${1:-r01}
SCRIPT=${2:-scripts/layer_profile.py}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/layer_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
grep -i "mfma" $OUT/counters.txt | head -40 > $OUT/mfma_counters.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/$SCRIPT > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/mfma1 -- python3 $REPO/$SCRIPT > $OUT/mfma1.log 2>&1 || { echo mfma1 failed; tail -5 $OUT/mfma1.log; }
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 --output-format csv -d $OUT/mfma2 -- python3 $REPO/$SCRIPT > $OUT/mfma2.log 2>&1 || { echo mfma2 failed; tail -5 $OUT/mfma2.log; }
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- python3 $REPO/$SCRIPT > $OUT/grbm.log 2>&1 || { echo grbm failed; tail -5 $OUT/grbm.log; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/$SCRIPT > $OUT/fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/fetch.log; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/$SCRIPT > $OUT/write.log 2>&1 || { echo write failed; tail -5 $OUT/write.log; }
rm -f $OUT/counters.txt
find $OUT -name "*.csv" | head; du -sh $OUT

#!/bin/bash
# MFMA-utilisation counters of scripts/kernel_table.py (every kernel family at the C4 shape): the bf16x3 kernels of
# rounds 2-3 (agg_dense_pc_kernel F = 256 / 512, dense_x3_pc_kernel, dense_wgrad_pc_kernel, dense_fused_kernel).
# Counters in their own --pmc passes (no trace domains beside them).  usage: prof_mfma_table.sh <tag>
set -o pipefail
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_mfma_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -- python3 $REPO/scripts/kernel_table.py > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $OUT/$name.log; }
done
du -sh $OUT

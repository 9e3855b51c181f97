#!/bin/bash
# rocprofv3 kernel trace of one model's training step (scripts/model_profile.py); usage: prof_model.sh KIND [NODES] [DIM]
set -o pipefail
KIND=${1:-gcn}; export KIND; export NODES=${2:-10000000}; export DIM=${3:-256}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/model_$KIND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $REPO/scripts/model_profile.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
find $OUT/t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/stats.csv
# whole steps only: the window between the (STEPS-2)th and STEPS-th launch of the loss's forward kernel (the last 2 of STEPS steps, default 5)
find $OUT/t -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $REPO/scripts/step_window.py {} softmax_ce_rows_kernel $((${STEPS:-5} - 2)) ${STEPS:-5} > $OUT/steady.csv
# the time-ordered launches of the first 400 ms after the graph build, to tell first-call effects from steady state
find $OUT/t -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $REPO/scripts/trace_head.py {} > $OUT/first_launches.txt
find $OUT/t -type f -delete
tail -2 $OUT/log.txt

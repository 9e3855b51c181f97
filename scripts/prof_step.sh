#!/bin/bash
# kernel trace of the training step with a fresh ego batch per step (bench.py --mode step); usage: prof_step.sh <tag> [idgcn|idgin] [centres]
set -o pipefail
TAG=${1:-r04}
MODEL=${2:-idgcn}
CEN=${3:-4096}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_step_${TAG}_$MODEL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --mode step --step-model $MODEL --centres $CEN --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/trace
head -25 $OUT/kernel_stats.csv | cut -c1-200

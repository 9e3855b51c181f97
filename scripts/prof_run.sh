#!/bin/bash
# Profiling passes for the headline workload on the GPU box (rocprofv3; counters in their own runs).
# usage: scripts/prof_run.sh <tag>   -> gpurun_out/prof_<tag>/{trace,fetch,write}
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/fetch.log; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1 || { echo write failed; tail -5 $OUT/write.log; }
find $OUT -name "*.csv" | head -30
du -sh $OUT

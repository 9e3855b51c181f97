#!/bin/bash
# per-kernel times of the ego-net expansion (scripts/ego_probe.py) -> gpurun_out/prof_ego/
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_ego
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/scripts/ego_probe.py > $OUT/run.log 2>&1 || { echo failed; tail -5 $OUT/run.log; }
find $OUT -name "*kernel_stats.csv" | head

#!/bin/bash
# kernel-trace + PMC passes of scripts/agg_forms_ab.py (sum / max / two-branch, tile and plan-based); usage: prof_agg_forms.sh <tag>
set -o pipefail
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_af_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/scripts/agg_forms_ab.py > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/scripts/agg_forms_ab.py > $OUT/fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/fetch.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/scripts/agg_forms_ab.py > $OUT/write.log 2>&1 || { echo write failed; tail -5 $OUT/write.log; exit 1; }
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
python3 $REPO/scripts/summarize_kernel_pmc.py $OUT $OUT/pmc.json > /dev/null
rm -rf $OUT/trace $OUT/fetch $OUT/write
du -sh $OUT

#!/bin/bash
# PMC passes over scripts/dbg/pmc_probe.py; usage: pmc_probe.sh <tag>
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" \
         "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $REPO/scripts/dbg/pmc_probe.py > $OUT/p$i.log 2>&1 || { echo pass $i failed; tail -3 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(p)))
    big = [r for r in rows if "agg_rows_kernel" in r["Kernel_Name"] or "agg_dense_kernel" in r["Kernel_Name"]]
    byd = collections.OrderedDict()
    for r in big:
        byd.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"][:40]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print(p.split("/")[-3] if "/" in p else p)
    for (did, name), c in byd.items():
        print("  ", did, name, {k: f"{v:.4g}" for k, v in c.items()})
PY

#!/usr/bin/env python3
"""profiles/<tag>_layer.json from scripts/prof_layer.sh: the aggregation kernel next to the MFMA GEMM that
follows it (one GCN layer, C2 size), with the MFMA counters north_star asks for."""
import csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, d = 1_000_000, 256
stats = {}
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
pmc = {}
for kind in ("mfma1", "mfma2", "grbm"):
    for f in glob.glob(os.path.join(src, kind, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            pmc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
def avg(kname, counter):
    for (k, c), v in pmc.items():
        if kname in k and c == counter:
            return sum(v) / len(v)
    return None
gemm = next((k for k in stats if "dense_fused_kernel" in k), None) or next(k for k in stats if k.startswith("Cijk_"))
gkey = "dense_fused_kernel" if "dense_fused_kernel" in gemm else "Cijk_"
agg = next(k for k in stats if "agg_rows_kernel" in k)
gemm_ns = stats[gemm][1]
flops = avg(gkey, "SQ_INSTS_VALU_MFMA_MOPS_F32") * 512
busy, gui = avg(gkey, "SQ_VALU_MFMA_BUSY_CYCLES"), avg(gkey, "GRBM_GUI_ACTIVE")
rec = {
    "workload": f"one GCN layer, aggregate then transform, BA({n},5)+loops, d={d}, fp32",
    "aggregation_kernel": {"name": agg, "avg_ns": stats[agg][1], "mfma_instructions": avg("agg_rows", "SQ_INSTS_VALU_MFMA_F32")},
    "gemm_kernel": {"name": gemm, "avg_ns": gemm_ns, "mfma_flops_counter": flops,
                    "tflops": flops / gemm_ns / 1e3, "frac_of_157.3_TF_f32_matrix_peak": flops / gemm_ns / 1e3 / 157.3,
                    "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_sum_over_8_XCDs": gui,
                    "MfmaUtil_percent": busy / ((gui / 8) * 1024) * 100,
                    "note": "MfmaUtil = MFMA busy cycles / (GUI-active cycles per XCD x 1024 SIMDs), the rocprofv3 derived-counter formula; "
                            "counters collected in separate --pmc passes"},
}
json.dump(rec, open(os.path.join(ROOT, "profiles", f"{tag}_layer.json" if len(sys.argv) < 4 else sys.argv[3]), "w"), indent=1)
print(json.dumps(rec, indent=1))

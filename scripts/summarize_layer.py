#!/usr/bin/env python3
"""profiles/<tag>_layer.json from scripts/prof_layer.sh: the aggregation kernel next to the MFMA GEMM that
follows it (one GCN layer, C2 size), with the MFMA counters north_star asks for."""
import csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, d = int(os.environ.get("NODES", "1000000")), 256
stats = {}
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
pmc = {}
for kind in ("mfma1", "mfma2", "grbm", "fetch", "write"):
    for f in glob.glob(os.path.join(src, kind, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            pmc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
def avg(kname, counter):
    for (k, c), v in pmc.items():
        if kname in k and c == counter:
            return sum(v) / len(v)
    return None
fused = next((k for k in stats if "agg_dense_kernel" in k), None)
if fused is not None:
    # one-kernel aggregate -> transform: HBM bytes and MFMA counters of the same kernel
    ns = stats[fused][1]
    flops = avg("agg_dense_kernel", "SQ_INSTS_VALU_MFMA_MOPS_F32") * 512
    busy, gui = avg("agg_dense_kernel", "SQ_VALU_MFMA_BUSY_CYCLES"), avg("agg_dense_kernel", "GRBM_GUI_ACTIVE")
    fetch, write = avg("agg_dense_kernel", "FETCH_SIZE"), avg("agg_dense_kernel", "WRITE_SIZE")
    nnz = int(os.environ.get("NNZ", "0"))
    rec = {
        "workload": f"one GCN layer as ONE kernel (aggregate -> transform), BA({n},5)+loops, F=d={d}, fp32",
        "kernel": {"name": fused, "calls": stats[fused][0], "avg_ns": ns,
                   "mfma_flops_counter": flops, "tflops": flops / ns / 1e3,
                   "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_sum_over_8_XCDs": gui,
                   "MfmaUtil_percent": busy / ((gui / 8) * 1024) * 100 if busy and gui else None,
                   "hbm_fetch_bytes": None if fetch is None else fetch * 1024 * 2,
                   "hbm_write_bytes": None if write is None else write * 1024,
                   "note": "FETCH_SIZE in KiB x 2 (gfx950 correction), WRITE_SIZE in KiB, as in the MI355X guide; every counter "
                           "from its own --pmc pass; MfmaUtil = busy / (GUI-active per XCD x 1024)"},
        "other_kernels": {k: {"calls": v[0], "avg_ns": v[1]} for k, v in stats.items() if k.startswith("void mp::") and k != fused},
    }
    if rec["kernel"]["hbm_fetch_bytes"]:
        tot = rec["kernel"]["hbm_fetch_bytes"] + (rec["kernel"]["hbm_write_bytes"] or 0)
        rec["kernel"]["hbm_TBps_from_counters"] = tot / ns / 1e3
    json.dump(rec, open(os.path.join(ROOT, "profiles", sys.argv[3] if len(sys.argv) > 3 else f"{tag}_layer_fused.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))
    sys.exit(0)
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

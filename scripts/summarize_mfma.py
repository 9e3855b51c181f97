#!/usr/bin/env python3
"""Per-kernel MFMA utilisation of scripts/kernel_table.py from the --pmc passes of scripts/prof_mfma_table.sh.

    MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   (rocprofv3's derived-counter formula: the
               busy counter sums over the chip's 1024 SIMDs — 32 cycles per 32x32x16 bf16 MFMA —, the collected
               GUI_ACTIVE value sums the 8 XCDs' clocks: duration x ~2.0 GHz x 8; the median launch of each is used)
    bf16 matrix rate = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 flop / profiled duration   (one MOP = 512 flop, as the F32
               counter's relation to 2 M F d showed in round 1: profiles/r01_layer.json)

    python scripts/summarize_mfma.py gpurun_out/prof_mfma_r04 profiles/r04_kernel_mfma.json
"""
import csv, glob, json, os, statistics, sys
src, dst = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(src, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "mp::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(
            (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = {}
for k, d in sorted(acc.items()):
    med = {c: statistics.median(v for v, _ in vals) for c, vals in d.items()}
    dur = statistics.median(t for vals in d.values() for _, t in vals)
    busy, gui = med.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), med.get("GRBM_GUI_ACTIVE", 0.0)
    mops = med.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    mops32 = med.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
    if busy == 0 and mops == 0 and mops32 == 0:
        continue
    e = {"launches": len(next(iter(d.values()))), "profiled_ms_median": round(dur / 1e6, 3),
         "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES": med.get("SQ_BUSY_CYCLES"), "GRBM_GUI_ACTIVE": gui,
         "SQ_INSTS_VALU_MFMA_MOPS_BF16": mops, "SQ_INSTS_VALU_MFMA_MOPS_F32": mops32}
    if gui:
        e["MfmaUtil_pct"] = round(100.0 * busy / (gui / 8.0 * 1024), 2)
        e["shader_clock_GHz_under_profiler"] = round(gui / 8.0 / dur, 3)
    if mops:
        e["bf16_matrix_TFLOPs_under_profiler"] = round(mops * 512 / dur / 1e3, 1)
    if mops32:
        e["f32_matrix_TFLOPs_under_profiler"] = round(mops32 * 512 / dur / 1e3, 1)
    out[k] = e
json.dump({"note": "rocprofv3 --pmc passes of scripts/kernel_table.py (separate runs per counter group); medians per launch; "
                   "MfmaUtil = MFMA busy cycles / (GUI-active cycles / 8 XCDs x 1024 SIMDs); one MFMA MOP = 512 flop; peak = 2.5 PFLOP/s bf16 (157 TFLOP/s f32) at 2.4 GHz",
           "kernels": out}, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])

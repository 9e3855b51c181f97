#!/usr/bin/env python3
"""Per-kernel HBM traffic of scripts/kernel_table.py from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
median bytes per launch of every engine kernel, corrected as MI355X_MICROARCH.md §HBM prescribes (KiB units; FETCH_SIZE
x 2 for wide coalesced reads on gfx950; WRITE_SIZE exact for 16-B-per-lane streaming stores).

    python scripts/summarize_kernel_pmc.py gpurun_out/prof_kt_r02 profiles/r02_kernel_pmc.json
"""
import csv, glob, json, os, statistics, sys
src, dst = sys.argv[1], sys.argv[2]
acc = {}
for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(src, kind, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "mp::" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc.setdefault(k, {}).setdefault(counter, []).append(
                (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = {}
for k, d in sorted(acc.items()):
    e = {}
    f = statistics.median(v for v, _ in d.get("FETCH_SIZE", [(0.0, 0)])) * 1024 * 2
    w = statistics.median(v for v, _ in d.get("WRITE_SIZE", [(0.0, 0)])) * 1024
    dur = statistics.median(t for _, t in d.get("FETCH_SIZE", d.get("WRITE_SIZE")))
    if f + w < 5e8:          # small / once-per-graph kernels: not worth a line
        continue
    out[k] = {"launches": len(d.get("FETCH_SIZE", [])), "fetch_GB_corrected": round(f / 1e9, 2), "write_GB": round(w / 1e9, 2),
              "hbm_GB": round((f + w) / 1e9, 2), "profiled_ms_median": round(dur / 1e6, 3),
              "hbm_GBps_under_profiler": round((f + w) / dur, 0)}
json.dump({"note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB units, separate --pmc passes of "
                   "scripts/kernel_table.py; median per launch; profiled passes run ~1-3 % slower than un-profiled ones",
           "kernels": out}, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])

#!/usr/bin/env python3
"""Condense the rocprofv3 passes of scripts/prof_run.sh into the committed evidence under profiles/.

    python scripts/summarize_prof.py gpurun_out/prof_r01 r01 <workload-tag>

Writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats summary, engine kernels only),
profiles/<tag>_pmc.json (FETCH_SIZE / WRITE_SIZE per launch of the aggregation kernel) and
profiles/pmc_traffic.json (what bench.py reports as roofline.traffic).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950
FETCH_SIZE reads exactly 1/2 of a wide (16 B/lane) coalesced read stream, so it is doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores.  The two counters come from separate passes.
"""
import csv
import glob
import json
import os
import statistics
import sys

src, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

rows = []
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "mp::" in r["Name"]:
            rows.append(r)
with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)

def is_agg(name):
    """the launches of the aggregation itself: the tile kernel in its aggregation-only form (last template argument
    AGG_ONLY = true; mp_agg_rows_tiles_f32) or the plan-based kernel"""
    if "agg_rows_kernel" in name:
        return True
    if "agg_dense_pc_kernel<" in name:
        args = name.split("agg_dense_pc_kernel<", 1)[1].split(">", 1)[0].split(",")
        # <W, WEIGHTED, U, KH, NCB, PF, NT_OUT, BF16X3, TR, NP, NC, HAS_S, AGG_ONLY[, MAXR]>: the sum / mean aggregation
        return len(args) > 12 and args[12].strip() == "true" and (len(args) < 14 or args[13].strip() == "false")
    return False


timed = None
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_trace.csv")):
    recs = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # bench.py's timed region ends before its bandwidth probes; the side measurement of the whole layer
    # (more aggregation launches) comes after them
    probe_at = next((int(r["Start_Timestamp"]) for r in recs if "read_probe_kernel" in r["Kernel_Name"]), None)
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in recs
         if is_agg(r["Kernel_Name"]) and (probe_at is None or int(r["Start_Timestamp"]) < probe_at)]
    # bench.py runs `warmup` + `steps` launches back to back, then up to 10 cold launches (a memset in between):
    # with --steps 10 --warmup 3 the timed region is launches 3..12 of the trace
    if len(d) >= 13:
        d = d[:13]
    if len(d) >= 10:
        timed = {"launches_in_trace_before_the_probes": len(d), "last_10_launches_avg_ns": sum(d[-10:]) / 10,
                 "min_ns": min(d),
                 "note": "the last 10 launches before the flush memsets / bandwidth probes are bench.py's timed region "
                         "(--steps 10); earlier ones are warm-up; later ones are the cold launches, the backward and the "
                         "`layer` side measurements"}
pmc = {}
for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(src, kind, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "mp::agg_" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            pmc.setdefault(k, {}).setdefault(counter, []).append(
                (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
summary = {}
for k, d in pmc.items():
    e = {}
    for counter, vals in d.items():
        e[counter + "_KiB_median"] = statistics.median(v for v, _ in vals)
        e[counter + "_launches"] = len(vals)
        e[counter + "_pass_duration_ns_median"] = statistics.median(t for _, t in vals)
    summary[k] = e
cands = [k for k in summary if is_agg(k)]
main = max(cands, key=lambda k: summary[k].get("FETCH_SIZE_launches", 0))      # the one the timed region launches
fetch = summary[main]["FETCH_SIZE_KiB_median"] * 1024 * 2     # gfx950: wide coalesced reads are tallied at 1/2
write = summary[main]["WRITE_SIZE_KiB_median"] * 1024
rec = {"workload": workload, "kernel": main, "fetch_bytes_corrected": fetch, "write_bytes": write,
       "hbm_bytes_per_launch": fetch + write,
       "note": "FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md §HBM), KiB units, separate --pmc passes; "
               "counters sit on the L2's fabric side, so Infinity-Cache hits are included",
       "kernel_trace_timed_region": timed,
       "per_kernel": summary}
json.dump(rec, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
json.dump({"workload": workload, "hbm_bytes_per_launch": fetch + write, "source": f"profiles/{tag}_pmc.json"},
          open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1)[:1200])

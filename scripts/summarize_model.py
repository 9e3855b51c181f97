#!/usr/bin/env python3
"""Condense scripts/prof_model.sh's stats.csv into profiles/<tag>_<kind>_step_kernels.csv.

    python scripts/summarize_model.py gpurun_out/model_gin r02 gin

Reads steady.csv (scripts/step_window.py: the last two of model_profile.py's five steps, bounded by launches of the loss
kernel), so the graph build, the placement calibration and first-call effects are outside the window."""
import csv
import os
import sys

src, tag, kind = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(os.path.join(src, "steady.csv"))))
steps = int(rows[0]["Steps"])
total = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
wall = float(rows[0]["WindowNs"]) / steps / 1e6
worst = max(rows, key=lambda r: float(r["MaxNs"]) / max(float(r["MinNs"]), 1.0) if float(r["MaxNs"]) > 2e6 else 0)
path = os.path.join(ROOT, "profiles", f"{tag}_{kind}_step_kernels.csv")
with open(path, "w", newline="") as f:
    f.write(f'"# rocprofv3 --kernel-trace of scripts/model_profile.py KIND={kind} (N=1e7, d=256): the last {steps} '
            f'training steps of the run (scripts/step_window.py), {total:.1f} ms of kernels per step in {wall:.1f} ms of wall time '
            f'per step under the profiler; largest max/min spread of a >2 ms kernel: '
            f'{float(worst["MaxNs"]) / float(worst["MinNs"]):.2f}x"\n')
    w = csv.writer(f)
    w.writerow(["kernel", f"calls_in_{steps}_steps", "ms_per_step", "percent"])
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"] if len(r["Name"]) <= 140 else r["Name"][:137] + "..."
        ms = float(r["TotalDurationNs"]) / steps / 1e6
        if ms < 0.05:
            continue
        w.writerow([name, r["Calls"], f"{ms:.2f}", f"{100 * ms / total:.2f}"])
print(path, f"{total:.1f} ms/step")

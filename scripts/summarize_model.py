#!/usr/bin/env python3
"""Condense scripts/prof_model.sh's stats.csv into profiles/<tag>_<kind>_step_kernels.csv.

    python scripts/summarize_model.py gpurun_out/model_gin r02 gin

ms_per_step = total duration / 3 (model_profile.py runs three training steps and one graph build); the copies of
the one-off placement calibration (mp::arena_copy_kernel) are left out."""
import csv
import os
import sys

src, tag, kind = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [r for r in csv.DictReader(open(os.path.join(src, "stats.csv"))) if "arena_copy_kernel" not in r["Name"]]
total = sum(float(r["TotalDurationNs"]) for r in rows) / 3e6
worst = max(rows, key=lambda r: float(r["MaxNs"]) / max(float(r["MinNs"]), 1.0) if float(r["MaxNs"]) > 2e6 else 0)
path = os.path.join(ROOT, "profiles", f"{tag}_{kind}_step_kernels.csv")
with open(path, "w", newline="") as f:
    f.write(f'"# rocprofv3 --kernel-trace --stats of scripts/model_profile.py KIND={kind} (3 training steps at N=1e7, '
            f'd=256, incl. one graph build; the one-off placement calibration copies are left out); total '
            f'{total:.1f} ms of kernels per step; largest max/min spread of a >2 ms kernel: '
            f'{float(worst["MaxNs"]) / float(worst["MinNs"]):.2f}x"\n')
    w = csv.writer(f)
    w.writerow(["kernel", "calls_in_3_steps", "ms_per_step", "percent"])
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"] if len(r["Name"]) <= 140 else r["Name"][:137] + "..."
        ms = float(r["TotalDurationNs"]) / 3e6
        if ms < 0.05:
            continue
        w.writerow([name, r["Calls"], f"{ms:.2f}", f"{100 * ms / total:.2f}"])
print(path, f"{total:.1f} ms/step")

"""Per-kernel totals of whole training steps from a rocprofv3 kernel_trace.csv: the window between two launches of a
marker kernel that runs once per step (the loss's forward kernel), so graph build, placement calibration and first-call
effects stay out.  Prints CSV: name, calls, total_ns, avg_ns, min_ns, max_ns, steps."""
import csv, sys
path, marker = sys.argv[1], sys.argv[2]
first, last = int(sys.argv[3]), int(sys.argv[4])          # marker occurrences (1-based) bounding the window
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if marker in r["Kernel_Name"]]
if len(marks) < last:
    sys.exit(f"only {len(marks)} launches of {marker}")
t0, t1, steps = marks[first - 1], marks[last - 1], last - first
agg = {}
for r in rows:
    s = int(r["Start_Timestamp"])
    if t0 <= s < t1:
        d = int(r["End_Timestamp"]) - s
        a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Steps", "WindowNs"])
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([k, a[0], a[1], a[1] / a[0], a[2], a[3], steps, t1 - t0])

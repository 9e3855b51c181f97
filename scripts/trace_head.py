"""Time-ordered list of the launches >= 1 ms from a rocprofv3 kernel_trace.csv (name, duration, start offset)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if dur >= 1.0:
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:10.2f} ms  {dur:9.3f} ms  {r["Kernel_Name"][:110]}')

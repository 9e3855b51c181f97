"""ms per step of every record in a scripts/bench_train.py JSON-lines file"""
import json, sys
for l in open(sys.argv[1]):
    try:
        r = json.loads(l)
    except Exception:
        continue
    print(r["what"], round(r["ms_per_step"], 1), "ms")

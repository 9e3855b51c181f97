"""Study for the round-3 placement design: outputs allocated by torch, checked against the tensors the launch reads
with timed copies, and re-allocated (holding the rejected candidate) when they conflict.  Prints, for the headline
aggregation (X 10 GB torch-allocated), the copy-probe cost and the aggregation time of successive torch allocations."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import graphgym_amd as ga
from graphgym_amd import _lib, graphgen, ops, placement

dev = torch.device("cuda:0")
n, d = 10_000_000, int(os.environ.get("D", "256"))
ei = graphgen.ba_edge_index(n, 5, seed=12345, device=dev)
g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
del ei
g.plan()
torch.cuda.empty_cache()
os.environ["MP_PLACEMENT"] = "off"
x = torch.empty((n, d), device=dev).uniform_(-1, 1)


def agg_ms(y, reps=3):
    ops._raw_spmm(g, x, _lib.SUM, out=y)
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops._raw_spmm(g, x, _lib.SUM, out=y)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def pair_ms(a, b, chunk=512 << 20, k=3):
    na, nb = a.numel() * 4, b.numel() * 4
    tot, cnt = 0.0, 0
    for i in range(k):
        for j in range(k):
            src = a.data_ptr() + ((na - chunk) * i // (k - 1)) // 256 * 256
            dst = b.data_ptr() + ((nb - chunk) * j // (k - 1)) // 256 * 256
            tot += min(placement._probe(src, dst, chunk, 1) for _ in range(2))
            cnt += 1
    return tot / cnt


held = []
rows = []
for i in range(int(os.environ.get("TRIES", "8"))):
    y = torch.empty((n, d), device=dev)
    t0 = time.perf_counter()
    p = pair_ms(x, y)
    tp = time.perf_counter() - t0
    a = agg_ms(y)
    t0 = time.perf_counter()
    gp, _ = placement.pair_cost_ms((x,), y)
    tg = time.perf_counter() - t0
    rows.append({"try": i, "x_ptr_gib": x.data_ptr() / 2 ** 30, "y_ptr_gib": y.data_ptr() / 2 ** 30,
                 "probe_ms_512MiB": p, "probe_wall_s": tp, "gather_probe_ms": gp, "gather_probe_wall_s": tg,
                 "aggregate_ms": a})
    print(json.dumps(rows[-1]), flush=True)
    held.append(y)
pm = min(r["probe_ms_512MiB"] for r in rows)
gm = min(r["gather_probe_ms"] for r in rows)
am = min(r["aggregate_ms"] for r in rows)
for r in rows:
    print(f"try {r['try']}: y at {r['y_ptr_gib']:8.2f} GiB  copy probe +{(r['probe_ms_512MiB'] / pm - 1) * 100:5.1f} %  "
          f"gather probe +{(r['gather_probe_ms'] / gm - 1) * 100:5.1f} % ({r['gather_probe_wall_s'] * 1e3:.1f} ms wall)   "
          f"aggregate {r['aggregate_ms']:.3f} ms (+{(r['aggregate_ms'] / am - 1) * 100:4.1f} %)")
# the same with every candidate released and re-allocated (what a second step of a training loop sees)
del held, y
y = torch.empty((n, d), device=dev)
print("after release: first allocation again at", y.data_ptr() / 2 ** 30, "GiB:", agg_ms(y), "ms")

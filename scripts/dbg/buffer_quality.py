"""Is a buffer's speed for the aggregation an INTRINSIC property (how the driver backed it) or a pairwise one?
Allocates K torch buffers of 10 GB (held), measures for each the gather probe reading it (writing a fixed small scratch)
and the aggregation with it as X (fixed Y) and as Y (fixed X)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops, placement
dev = torch.device("cuda:0")
n, d = 10_000_000, 256
g = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n, 5, 12345, device=dev), n, add_self_loops=True).gcn_norm("row")
g.plan()
if os.environ.get("EMPTY_CACHE", "1") == "1":
    torch.cuda.empty_cache()
K = int(os.environ.get("K", "10"))
scratch = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
bufs = [torch.empty((n, d), device=dev) for _ in range(K)]
for b in bufs:
    b.uniform_(-1, 1)

def agg(x, y):
    for _ in range(2): ops._raw_spmm(g, x, 0, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops._raw_spmm(g, x, 0, out=y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3

nb = n * d * 4
rows = []
for i, b in enumerate(bufs):
    q = min(placement._probe_gather(b.data_ptr(), nb // 1024 * 1024, scratch.data_ptr(), 256 << 20, 1) for _ in range(3))
    rows.append({"i": i, "ptr_gib": round(b.data_ptr() / 2 ** 30, 2), "gather_read_ms": round(q, 4)})
for i in range(K):
    rows[i]["as_x_ms"] = round(agg(bufs[i], bufs[(i + K // 2) % K]), 3)     # a far-away partner
    rows[i]["as_y_of_x0_ms"] = round(agg(bufs[0], bufs[i]), 3) if i else None
    rows[i]["as_y_of_x1_ms"] = round(agg(bufs[1], bufs[i]), 3) if i != 1 else None
for r in rows:
    print(json.dumps(r), flush=True)

"""Where the round-1 form of the ID-GCN layer (two-branch aggregation + dual transform) spends its time."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graphgym_amd import graphgen, ops, placement
from graphgym_amd.graph import CSRGraph
dev = torch.device("cuda:0")
n, d = 10_000_000, 256
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
g = CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm()
del ei
ids = torch.arange(0, n, 100, device=dev)
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters, 3)
x = placement.empty_or_torch((n, d), dev); x.uniform_(-1, 1)
W = torch.randn(d, d, device=dev) * 0.05
Wid = torch.randn(d, d, device=dev) * 0.05
b = torch.randn(d, device=dev)
r = {}
with torch.no_grad():
    r["two_branch"] = timeit(lambda: ops.idgnn_aggregate(g, ids, x))
    P, Q = ops.idgnn_aggregate(g, ids, x)
    r["dual_transform"] = timeit(lambda: ops.dense_fused(P, W, Q, Wid, b, relu=True))
    r["single_transform"] = timeit(lambda: ops.dense_fused(P, W, None, None, b, relu=True))
    del P, Q
    def both():
        P, Q = ops.idgnn_aggregate(g, ids, x)
        return ops.dense_fused(P, W, Q, Wid, b, relu=True)
    r["both"] = timeit(both)
    os.environ["MP_AGG_TILES"] = "0"
    r["both_plan"] = timeit(both)
    r["two_branch_plan"] = timeit(lambda: ops.idgnn_aggregate(g, ids, x))
r["placement"] = {k: v for k, v in placement.stats().items() if not isinstance(v, (list, dict))} if hasattr(placement, "stats") else None
print(json.dumps(r))

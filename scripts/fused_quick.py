"""Quick A/B of the one-kernel layer at the C4 shape (developer tool): aggregation alone, exact-f32 and bf16x3 products."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops, placement
dev = torch.device("cuda:0")
n, d = int(os.environ.get("NODES", "10000000")), int(os.environ.get("DIM", "256"))
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev, permute_seed=1 if os.environ.get('PERMUTE') else None)
g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
del ei
g.plan()
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
x = placement.empty_or_torch((n, d), dev); x.uniform_(-1, 1)
y = placement.empty_or_torch((n, d), dev, reads=(x,), tries=9, accept=-1.0)
W = torch.randn(d, d, device=dev) * 0.05
b = torch.randn(d, device=dev)
res = {"tag": os.environ.get("TAG", "") or os.path.basename(os.environ.get("MP_ENGINE_LIB", "")), "agg_ms": timeit(lambda: ops._raw_spmm(g, x, 0, out=y))}
res["agg_dense_f32_ms"] = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=False))
res["agg_dense_bf16x3_ms"] = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=True))
P = placement.empty_or_torch((n, d), dev, reads=(x,))
res["agg_dense_bf16x3_keepP_ms"] = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=True, want_P=True))
print(json.dumps(res), flush=True)

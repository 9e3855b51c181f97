"""(Round-3 study; needs a debug build: the launches of agg_rows_kernel / agg_dense_kernel were given `atoi(getenv(...))`
bytes of dynamic LDS — a two-line patch that is not in the tree.)  How much of the one-kernel layer's gap to the plain aggregation is occupancy?  The plain kernel is launched with
dynamic LDS it does not use, which caps the workgroups per CU (160 KiB / bytes): MP_DEBUG_SPMM_LDS in the environment."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops
dev = torch.device("cuda:0")
n, d = 10_000_000, 256
g = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n, 5, 12345, device=dev), n, add_self_loops=True).gcn_norm("row")
g.plan()
torch.cuda.empty_cache()
bufs = [torch.empty((n, d), device=dev) for _ in range(9)]
x, y = bufs[0].uniform_(-1, 1), bufs[8]            # nine allocations apart: the fast band (scripts/dbg/buffer_quality.py)
from graphgym_amd import placement
def t():
    for _ in range(2): ops._raw_spmm(g, x, 0, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops._raw_spmm(g, x, 0, out=y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
r = {"lds_bytes": int(os.environ.get("MP_DEBUG_SPMM_LDS", "0")), "agg_ms": t(), "place": getattr(y, "_mp_place", None)}
W = torch.randn(d, d, device=dev) * 0.05
b = torch.randn(d, device=dev)
def tf():
    for _ in range(2): ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
r["fused_lds"] = int(os.environ.get("MP_DEBUG_FUSED_LDS", "0"))
r["fused_ms"] = tf()
print(json.dumps(r))

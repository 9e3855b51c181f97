"""max and the two-branch aggregation on the tile structure against the plan-based kernel (MP_AGG_TILES=0), same
process, same graph and buffers: N = 10^7 Barabasi-Albert rows (the bench graph), GCN-normalised, 1 % identity nodes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import graphgen, ops, placement
from graphgym_amd.graph import CSRGraph
dev = torch.device("cuda:0")
n = int(os.environ.get("NODES", "10000000"))
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
g = CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm()
del ei
nnz = g.nnz
ids = torch.arange(0, n, 100, device=dev)


def timeit(fn, iters=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for d in (256, 512):
    x = placement.empty_or_torch((n, d), dev)
    x.uniform_(-1, 1)
    y = placement.empty_or_torch((n, d), dev, reads=(x,))
    agg = (nnz * (d * 4 + 8) + n * (d * 4 + 4)) / 1e9
    row = {"d": d, "nnz": nnz, "algorithmic_GB": round(agg, 2), "two_branch_GB": round(agg + n * d * 4 / 1e9, 2)}
    with torch.no_grad():
        for tiles in ("1", "0", "1", "0"):
            os.environ["MP_AGG_TILES"] = tiles
            k = "tiles" if tiles == "1" else "plan"
            for name, fn in (("sum", lambda: ops._raw_spmm(g, x, 0, out=y)), ("max", lambda: ops._raw_spmm(g, x, 2, out=y)),
                             ("two_branch", lambda: ops.idgnn_aggregate(g, ids, x))):
                t = timeit(fn)
                key = f"{name}_{k}_ms"
                row[key] = round(min(row.get(key, 1e9), t), 3)
                torch.cuda.empty_cache()
    br = g.id_branch(ids)
    row["identity_rows"] = br.n_rows
    row["identity_entries"] = int(br.slot.numel())
    for name, b in (("sum", agg), ("max", agg), ("two_branch", agg + n * d * 4 / 1e9)):
        row[f"{name}_tiles_frac_of_8TBps"] = round(b / row[f"{name}_tiles_ms"] / 8.0, 3)
    print(json.dumps(row), flush=True)
    del x, y
    torch.cuda.empty_cache()

"""Does the aggregation of row chunk i+1 overlap the MFMA transform of chunk i when the two run on
separate HIP streams?  GCN layer forward at the C4 size: relu((A_hat X) W + b)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import graphgen, ops
from graphgym_amd.graph import CSRGraph
dev = torch.device("cuda:0")
n, d = int(os.environ.get("NODES", "10000000")), int(os.environ.get("DIM", "256"))
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
g = CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm()
del ei
x = torch.rand(n, d, device=dev) * 2 - 1
W = torch.randn(d, d, device=dev) * 0.05
b = torch.randn(d, device=dev) * 0.1
P = torch.empty(n, d, device=dev)
out = torch.empty(n, d, device=dev)
g.plan()

def timeit(fn, iters=10, warm=3):
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

def sequential():
    ops._raw_spmm(g, x, 0, out=P)
    ops._dense_into(out, P, W, b, True)

res = {"n": n, "d": d}
res["agg_ms"] = timeit(lambda: ops._raw_spmm(g, x, 0, out=P))
res["gemm_ms"] = timeit(lambda: ops._dense_into(out, P, W, b, True))
res["sequential_ms"] = timeit(sequential)
ref = out.clone()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for chunks in (2, 4, 8, 16, 32):
    bounds = [n * i // chunks for i in range(chunks + 1)]
    parts = [g.row_slice(bounds[i], bounds[i + 1]) for i in range(chunks)]
    for p in parts:
        p.plan()
    evs = [torch.cuda.Event() for _ in range(chunks)]
    def piped():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        for i, p in enumerate(parts):
            with torch.cuda.stream(sa):
                ops._raw_spmm(p, x, 0, out=P[bounds[i]:bounds[i + 1]])
                evs[i].record(sa)
            with torch.cuda.stream(sb):
                sb.wait_event(evs[i])
                ops._dense_into(out[bounds[i]:bounds[i + 1]], P[bounds[i]:bounds[i + 1]], W, b, True)
        cur.wait_stream(sa); cur.wait_stream(sb)
    out.zero_()
    res[f"piped_{chunks}_ms"] = timeit(piped)
    res[f"piped_{chunks}_equal"] = bool(torch.equal(out, ref))
    def chunked_seq():
        for i, p in enumerate(parts):
            ops._raw_spmm(p, x, 0, out=P[bounds[i]:bounds[i + 1]])
            ops._dense_into(out[bounds[i]:bounds[i + 1]], P[bounds[i]:bounds[i + 1]], W, b, True)
    res[f"chunkseq_{chunks}_ms"] = timeit(chunked_seq)
print(json.dumps(res))

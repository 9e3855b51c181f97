"""GPU sweep (developer tool): A/B of hot-kernel variants interleaved in one process, then
feature widths / reduces / two-branch / transposed (backward) operator at the headline size."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import ops, graphgen, _lib

dev = torch.device("cuda:0")
L = _lib.lib()

def timeit(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def balg(n, nnz, d, weighted, extra_rows=0):
    return nnz * d * 4 + n * d * 4 * (1 + extra_rows) + nnz * 4 + (nnz * 4 if weighted else 0) + (n + 1) * 4

def main(n):
    ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
    g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
    gu = ga.CSRGraph.from_edge_index(ei, n)       # unweighted, no loops (GIN / SAGE)
    del ei
    d = 256
    x = torch.rand(n, d, device=dev) * 2 - 1
    y = torch.empty(n, d, device=dev)
    out = {}
    # ---- variants, interleaved rounds
    variants = [(8, 1), (8, 0), (8, 9), (16, 1), (8, 5)]
    times = {v: [] for v in variants}
    for rnd in range(4):
        for v in variants:
            assert L.mp_spmm_kernel_config(*v) == 0
            ops._raw_spmm(g, x, 0, out=y)
            torch.cuda.synchronize()
            times[v].append(timeit(lambda: ops._raw_spmm(g, x, 0, out=y), 5))
    L.mp_spmm_kernel_config(8, 1)
    for v in variants:
        t = sorted(times[v])
        print(f"variant U={v[0]} VAR={v[1]}: median {t[len(t)//2]:.3f} ms  min {t[0]:.3f}  -> {balg(n,g.nnz,d,True)/t[len(t)//2]/1e6:.0f} GB/s", flush=True)
    if len(sys.argv) > 2 and sys.argv[2] == 'variants':
        return
    # ---- permuted node order (hubs scattered)
    ei = graphgen.ba_edge_index(n, 5, 12345, device=dev, permute_seed=1)
    gp = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
    del ei
    ops._raw_spmm(gp, x, 0, out=y)
    t = timeit(lambda: ops._raw_spmm(gp, x, 0, out=y), 10)
    print(f"permuted order d=256 sum weighted: {t:.3f} ms  {gp.nnz/t/1e6:.2f} Gedges/s  {balg(n,gp.nnz,d,True)/t/1e6:.0f} GB/s", flush=True)
    del gp
    # ---- transposed operator (backward) of the normalised graph
    t0 = time.time(); gt = g.transpose(); torch.cuda.synchronize(); print(f"transpose build {time.time()-t0:.2f}s", flush=True)
    ops._raw_spmm(gt, x, 0, out=y)
    t = timeit(lambda: ops._raw_spmm(gt, x, 0, out=y), 10)
    print(f"transposed (backward) d=256: {t:.3f} ms  {gt.nnz/t/1e6:.2f} Gedges/s  {balg(n,gt.nnz,d,True)/t/1e6:.0f} GB/s", flush=True)
    g._t = None; del gt
    del x, y
    torch.cuda.empty_cache()
    # ---- widths x reduces
    for d in (64, 128, 256, 512):
        x = torch.rand(n, d, device=dev) * 2 - 1
        y = torch.empty(n, d, device=dev)
        for name, gg, red, kw in [("gcn-sum-w", g, 0, {}), ("sum", gu, 0, {}), ("mean", gu, 1, {}), ("max", gu, 2, {}),
                                  ("gin-sum+self", gu, 0, dict(S=x, self_scale=1.0))]:
            ops._raw_spmm(gg, x, red, out=y, **kw)
            t = timeit(lambda: ops._raw_spmm(gg, x, red, out=y, **kw), 6)
            b = balg(n, gg.nnz, d, gg.val is not None, 1 if kw else 0)
            print(f"d={d:4d} {name:13s}: {t:8.3f} ms  {gg.nnz/t/1e6:6.2f} Gedges/s  alg {b/t/1e6:6.0f} GB/s ({b/t/1e6/80:.1f}%)", flush=True)
        if d in (128, 256):
            ids = torch.arange(0, n, 64, device=dev)
            colm = g.mark_ids(ids)
            xg = x
            P, Q = ops.idgnn_aggregate(g, ids, xg, col_marked=colm)
            t = timeit(lambda: ops.idgnn_aggregate(g, ids, xg, col_marked=colm), 5)
            b = balg(n, g.nnz, d, True, 1)
            print(f"d={d:4d} two-branch   : {t:8.3f} ms  {g.nnz/t/1e6:6.2f} Gedges/s  alg {b/t/1e6:6.0f} GB/s", flush=True)
            del P, Q
        del x, y
        torch.cuda.empty_cache()

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000)

"""Developer check on a GPU box: parity of the aggregation kernels against a plain
torch formulation on the same device + timing sweep.  Not part of the test suite."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import ops, graphgen, _lib

dev = torch.device("cuda:0")
torch.manual_seed(0)

def ref_spmm(ei, w, x, N, reduce):
    src, dst = ei[0], ei[1]
    msg = x[src]
    if w is not None:
        msg = msg * w[:, None]
    if reduce == "sum":
        return torch.zeros(N, x.size(1), device=x.device).index_add_(0, dst, msg)
    if reduce == "mean":
        s = torch.zeros(N, x.size(1), device=x.device).index_add_(0, dst, msg)
        c = torch.zeros(N, device=x.device).index_add_(0, dst, torch.ones_like(dst, dtype=torch.float32))
        return s / c.clamp(min=1)[:, None]
    out = torch.full((N, x.size(1)), float("-inf"), device=x.device)
    out = out.scatter_reduce(0, dst[:, None].expand_as(msg), msg, "amax", include_self=True)
    return torch.where(torch.isinf(out), torch.zeros_like(out), out)

def check_small():
    for N, E, d in [(1, 0, 4), (7, 20, 3), (100, 1000, 64), (1000, 5000, 128), (5000, 100000, 256),
                    (3000, 60000, 512), (2000, 30000, 100), (500, 4000, 1), (50, 40000, 256)]:
        ei = torch.randint(0, N, (2, E), device=dev)
        w = torch.rand(E, device=dev)
        x = torch.randn(N, d, device=dev)
        for weighted in (False, True):
            g = ga.CSRGraph.from_edge_index(ei, N, w if weighted else None)
            assert g.nnz == E
            for red in ("sum", "mean", "max"):
                y = ops.spmm(g, x, red)
                r = ref_spmm(ei, w if weighted else None, x, N, red)
                err = (y - r).abs().max().item() if N * d else 0.0
                tol = 1e-5 * max(1.0, r.abs().max().item() if r.numel() else 1.0)
                flag = "ok" if err <= 10 * tol else "FAIL"
                print(f"N={N} E={E} d={d} w={weighted} {red}: maxerr={err:.3e} {flag}", flush=True)
                assert flag == "ok"
    print("small parity ok", flush=True)

def bench(n, d, iters=10):
    t0 = time.time()
    ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
    torch.cuda.synchronize(); t1 = time.time()
    g = ga.CSRGraph.from_edge_index(ei, n, None, add_self_loops=True).gcn_norm("row")
    torch.cuda.synchronize(); t2 = time.time()
    plan, counts = g.plan()
    torch.cuda.synchronize(); t3 = time.time()
    print(f"n={n}: gen {t1-t0:.2f}s  csr+norm {t2-t1:.2f}s plan {t3-t2:.3f}s nnz={g.nnz} counts={list(counts)}", flush=True)
    x = torch.rand(n, d, device=dev) * 2 - 1
    y = ops.spmm(g, x, "sum")
    if n <= 2_000_000:
        eiw = torch.stack([g.col.long(), g.row_ids().long()])
        r = ref_spmm(eiw, g.val, x, n, "sum")
        print("  parity maxerr", (y - r).abs().max().item(), flush=True)
    for cfg in [(320, 4, 1024, 256), (192, 4, 1024, 256), (512, 4, 1024, 256), (1024, 8, 2048, 512), (128, 2, 1024, 256), (320, 16, 1024, 256)]:
        _lib.lib().mp_spmm_plan_config(*cfg)
        g._plan = None
        g.plan()
        for _ in range(2):
            ops._raw_spmm(g, x, 0)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            ops._raw_spmm(g, x, 0)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        balg = g.nnz * d * 4 + n * d * 4 + g.nnz * 8 + (n + 1) * 4
        print(f"  cfg={cfg} d={d}: {ms:.3f} ms  {g.nnz/ms/1e6:.2f} Gedges/s  alg {balg/ms/1e6:.0f} GB/s ({balg/ms/1e6/8000*100:.1f}% of 8TB/s)", flush=True)
    _lib.lib().mp_spmm_plan_config(320, 4, 1024, 256)

if __name__ == "__main__":
    check_small()
    bench(1_000_000, 256)
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        bench(10_000_000, 256, iters=5)

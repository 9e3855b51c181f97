"""Library GEMM layouts against the engine's transform kernel at M = 10^7 (which one should each backward product use?)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import ops
dev = torch.device("cuda:0")
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
for (M, F, d) in [(10_000_000, 256, 256), (10_000_000, 256, 128), (10_000_000, 128, 256), (5_000_000, 512, 512), (150_000, 512, 512)]:
    x = torch.randn(M, F, device=dev)
    g = torch.randn(M, d, device=dev)
    W = torch.randn(F, d, device=dev) / F ** 0.5           # [in, out] as the layers store it
    Wt = W.t().contiguous()
    b = torch.randn(d, device=dev)
    out = torch.empty(M, d, device=dev)
    res = {"M": M, "F": F, "d": d}
    res["fwd_engine_x@W+b_relu"] = timeit(lambda: ops._raw_dense_fused(x, W, None, None, b, True))
    res["fwd_lib_addmm_then_relu"] = timeit(lambda: torch.relu_(torch.addmm(b, x, W)))
    res["fwd_lib_x@W_only"] = timeit(lambda: torch.mm(x, W))
    res["dX_engine_g@Wt(contig)"] = timeit(lambda: ops._raw_dense_fused(g, Wt, None, None, None, False))
    res["dX_lib_g@W.t()_NT"] = timeit(lambda: torch.mm(g, W.t()))
    res["dX_lib_g@Wt(contig)_NN"] = timeit(lambda: torch.mm(g, Wt))
    res["dW_engine_wgrad"] = timeit(lambda: ops._raw_dense_wgrad(x, g))
    res["dW_lib_x.t()@g_TN"] = timeit(lambda: torch.mm(x.t(), g))
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)
    del x, g, out

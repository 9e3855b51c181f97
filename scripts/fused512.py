"""The one-kernel aggregate -> transform at config C5's width (F = d_out = 512, two K halves over the row tile) against
the two-kernel order, N = 10^7; also F = 512 -> 256.  JSON lines."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops, placement
dev = torch.device("cuda:0")
n = int(os.environ.get("NODES", "10000000"))
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
del ei
g.plan()

def timeit(fn, iters=4, warm=2):
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

F = 512
x = placement.empty_or_torch((n, F), dev)
x.uniform_(-1, 1)
for d in (512, 256):
    W = torch.randn(F, d, device=dev) * 0.04
    b = torch.randn(d, device=dev)
    P = placement.empty_or_torch((n, F), dev, reads=(x,))
    out = placement.empty_or_torch((n, d), dev, reads=(x, P), tries=9, accept=-1.0)
    t_agg = timeit(lambda: ops._raw_spmm(g, x, 0, out=P))
    t_dense = timeit(lambda: ops._dense_into(out, P, W, b, True))
    ref = out[:2048].clone()
    t_f32 = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True, out=out, bf16x3=False))
    t_one = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True, out=out))
    err = float((out[:2048] - ref).abs().max() / ref.abs().max())
    alg = (g.nnz * (F * 4 + 8) + n * (d * 4 + 4)) / 1e9
    print(json.dumps({"lib": os.path.basename(os.environ.get("MP_ENGINE_LIB", "")), "what": f"relu((A x) W + b), F = {F}, d_out = {d}, N = {n}", "aggregation_ms": t_agg,
                      "transform_ms": t_dense, "two_kernel_ms": t_agg + t_dense, "one_kernel_ms": t_one, "one_kernel_exact_f32_mfma_ms": t_f32,
                      "algorithmic_GB": alg, "one_kernel_frac_hbm": alg / t_one * 1e3 / 8000.0,
                      "mfma_tflops_inside_one_kernel": 2.0 * n * F * d / (t_one * 1e-3) / 1e12,
                      "max_rel_diff_first_2048_rows": err}), flush=True)
    del P, out

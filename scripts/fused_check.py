"""mp_agg_dense_f32 against aggregation + transform as two launches: correctness sweep, then timing at C4."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import graphgen, ops
from graphgym_amd.graph import CSRGraph
dev = torch.device("cuda:0")
torch.manual_seed(0)
def two_step(g, x, W, b, relu, S, s):
    P, _ = ops._raw_spmm(g, x, 0, S=S, self_scale=s)
    return ops._raw_dense_fused(P, W, None, None, b, relu), P
worst = 0.0
for (n, E, F, d, weighted, selfterm, hub) in [
        (1000, 12000, 256, 256, True, False, False), (1000, 12000, 256, 256, False, True, False),
        (37, 200, 64, 10, True, False, False), (5000, 40000, 128, 64, True, True, True),
        (3333, 50000, 256, 130, True, False, True), (64, 0 + 64, 256, 512, False, False, False),
        (20000, 400000, 256, 256, True, False, True)]:
    ei = torch.randint(0, n, (2, E), device=dev)
    if hub:   # a few very long rows, some inside one tile, plus empty rows
        extra = torch.stack([torch.randint(0, n, (9000,), device=dev), torch.full((9000,), 5, device=dev)])
        extra2 = torch.stack([torch.randint(0, n, (3000,), device=dev), torch.full((3000,), 7, device=dev)])
        ei = torch.cat([ei, extra, extra2], dim=1)
        ei = ei[:, (ei[1] % 11 != 3)]           # rows = 3 mod 11 empty
    w = torch.rand(ei.size(1), device=dev) if weighted else None
    g = CSRGraph.from_edge_index(ei, n, w)
    x = torch.randn(n, F, device=dev)
    W = torch.randn(F, d, device=dev) / F ** 0.5
    b = torch.randn(d, device=dev)
    S, s = (x, 1.3) if selfterm else (None, 0.0)
    for relu in (False, True):
        ref, Pref = two_step(g, x, W, b, relu, S, s)
        out, P = ops._raw_agg_dense(g, x, W, b, relu, S, s, want_P=True)
        out2, _ = ops._raw_agg_dense(g, x, W, b, relu, S, s)
        errP = float((P - Pref).abs().max() / Pref.abs().max().clamp_min(1))
        err = float((out - ref).abs().max() / ref.abs().max().clamp_min(1))
        assert torch.equal(out, out2)
        worst = max(worst, err, errP)
        print(f"n={n} E={ei.size(1)} F={F} d={d} w={weighted} self={selfterm} hub={hub} relu={relu}: err={err:.2e} errP={errP:.2e}", flush=True)
        assert err < 1e-5 and errP < 1e-5
print("worst", worst)
if os.environ.get("NO_BIG"):
    sys.exit(0)
n, d = 10000000, 256
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev, permute_seed=(7 if os.environ.get('PERMUTE') else None))
g = CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm()
del ei
x = torch.rand(n, d, device=dev) * 2 - 1
W = torch.randn(d, d, device=dev) * 0.05
b = torch.randn(d, device=dev) * 0.1
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
from graphgym_amd._lib import lib
res = {}
res["agg_ms"] = timeit(lambda: ops._raw_spmm(g, x, 0))
res["two_step_ms"] = timeit(lambda: two_step(g, x, W, b, True, None, 0.0))
for u, var in ((8, 32),):
    lib().mp_fused_config(u, var)
    res[f"fused_u{u}_var{var}_ms"] = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True))
lib().mp_fused_config(8, 0)
res["fused_ms"] = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True))
res["fused_saveP_ms"] = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True, want_P=True))
ref, _ = two_step(g, x, W, b, True, None, 0.0)
out, _ = ops._raw_agg_dense(g, x, W, b, True)
res["big_err"] = float((out - ref).abs().max() / ref.abs().max())
print(json.dumps(res))

"""How the one-kernel aggregate -> transform scales with the MFMA work per tile (d_out) at the C4 shape."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import graphgen, ops
from graphgym_amd.graph import CSRGraph
dev = torch.device("cuda:0")
n, F = 10000000, int(os.environ.get("F", "256"))
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
g = CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm()
del ei
x = torch.rand(n, F, device=dev) * 2 - 1
g.plan()
def timeit(fn, iters=8, warm=2):
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
y = torch.empty(n, F, device=dev)
res = {"F": F, "agg_ms": timeit(lambda: ops._raw_spmm(g, x, 0, out=y))}
del y
from graphgym_amd._lib import lib
for var, name in ((160, "split_roles"), (416, "split_roles_B_regs_only"), (296, "phaseB_regs_only"), (36, "phaseA"), (40, "phaseB"), (32, "full")):
    lib().mp_fused_config(8, var)
    W = torch.randn(F, 256, device=dev) * 0.05
    out = torch.empty(n, 256, device=dev)
    res[f"d256_{name}_ms"] = timeit(lambda: ops._raw_agg_dense(g, x, W, None, True, out=out))
    del out
lib().mp_fused_config(8, 32)
for d in ():
    W = torch.randn(F, d, device=dev) * 0.05
    out = torch.empty(n, d, device=dev)
    res[f"fused_dout{d}_ms"] = timeit(lambda: ops._raw_agg_dense(g, x, W, None, True, out=out))
    del out
print(json.dumps(res))

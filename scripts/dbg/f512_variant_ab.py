"""F = 512 -> 512 one-kernel layer: 4 multiplying waves with two column blocks each (default) against 8 with one
(the default since round 4; MP_FUSED_VARIANT=7: the four-wave form), with and without the self term / kept rows, same process and buffers."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops, placement
dev = torch.device("cuda:0")
n = 10_000_000
g = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n, 5, 12345, device=dev), n, add_self_loops=True).gcn_norm("row")
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters, 3)
F = 512
x = placement.empty_or_torch((n, F), dev); x.uniform_(-1, 1)
W = torch.randn(F, F, device=dev) * 0.04
b = torch.randn(F, device=dev)
out = placement.empty_or_torch((n, F), dev, reads=(x,))
r = {}
outs = {}
for rep in range(2):
    for v in ("7", "0"):
        os.environ["MP_FUSED_VARIANT"] = v
        for name, kw in (("plain", {}), ("self", {"S": x, "self_scale": 1.0})):
            t = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True, out=out, **kw))
            k = f"{name}_v{v}"
            r[k] = min(r.get(k, 1e9), t)
            outs[k] = out[:4096].clone()
r["plain_same_bits"] = bool(torch.equal(outs["plain_v7"], outs["plain_v0"]))
r["self_same_bits"] = bool(torch.equal(outs["self_v7"], outs["self_v0"]))
print(json.dumps(r))

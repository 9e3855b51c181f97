"""In-process A/B of builds of the one-kernel layer: every library listed in LIBS (comma separated paths) runs on the
SAME buffers, interleaved, so placement does not enter the comparison."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops, placement, _lib
dev = torch.device("cuda:0")
n, d = 10_000_000, int(os.environ.get("DIM", "256"))
dout = int(os.environ.get("DOUT", str(d)))
g = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n, 5, 12345, device=dev, permute_seed=1 if os.environ.get("PERMUTE") else None), n, add_self_loops=True).gcn_norm("row")
g.plan()
torch.cuda.empty_cache()
x = torch.empty((n, d), device=dev).uniform_(-1, 1)
y = placement.empty_or_torch((n, dout), dev, reads=(x,), tries=9, accept=-1.0)
P = placement.empty_or_torch((n, d), dev, reads=(x,), tries=4)
W = torch.randn(d, dout, device=dev) * 0.05
b = torch.randn(dout, device=dev)
libs = {}
base = _lib.lib()
for path in os.environ["LIBS"].split(","):
    h = C.CDLL(path)
    for name, (res, args) in _lib.PROTOTYPES.items():
        fn = getattr(h, name); fn.restype = res; fn.argtypes = args
    libs[os.path.basename(path)] = h

def run(h, keepP):
    _lib._lib = h
    ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=True) if not keepP else ops._raw_agg_dense.__wrapped__ if False else None
def once(h, keepP):
    _lib._lib = h
    L = h
    Wc = W.contiguous(); Wsp = ops._split_bf16_t(Wc)
    from graphgym_amd._lib import ptr, check
    from graphgym_amd.graph import _stream
    check(L.mp_agg_dense_f32(ptr(g.rowptr), ptr(g.col), ptr(g.val), n, 0, ptr(x), x.stride(0), d, None, 0, 0.0, ptr(Wc), Wc.stride(0), dout,
                             ptr(b), 1, None, ptr(P) if keepP else None, P.stride(0) if keepP else 0, ptr(y), y.stride(0), ptr(Wsp), _stream()))
def t(h, keepP, reps=5):
    once(h, keepP); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): once(h, keepP)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = {k: {"out_only": [], "keepP": []} for k in libs}
for rnd in range(3):
    for k, h in libs.items():
        res[k]["out_only"].append(round(t(h, False), 3))
        res[k]["keepP"].append(round(t(h, True), 3))
_lib._lib = base
ya, _ = ops._raw_spmm(g, x, 0, out=y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops._raw_spmm(g, x, 0, out=y)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"d": d, "dout": dout, "plain_agg_ms": round(e0.elapsed_time(e1) / 5, 3), "variants": res}))

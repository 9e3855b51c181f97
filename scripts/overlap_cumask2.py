"""Do an aggregation and an MFMA transform with no dependency run concurrently on CU-masked streams?"""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import graphgen, ops
from graphgym_amd._lib import lib, check
from graphgym_amd.graph import CSRGraph
dev = torch.device("cuda:0")
n, d = 10000000, 256
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
g = CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm()
del ei
x = torch.rand(n, d, device=dev) * 2 - 1
W = torch.randn(d, d, device=dev) * 0.05
b = torch.randn(d, device=dev) * 0.1
P = torch.empty(n, d, device=dev)
P2 = torch.rand(n, d, device=dev)
out = torch.empty(n, d, device=dev)
g.plan()
def masked_stream(words):
    w = (C.c_uint32 * len(words))(*words)
    h = C.c_void_p()
    check(lib().mp_stream_create_cu_mask(w, len(words), C.byref(h)), "mp_stream_create_cu_mask")
    return torch.cuda.ExternalStream(h.value)
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
agg = lambda: ops._raw_spmm(g, x, 0, out=P)
gemm = lambda: ops._dense_into(out, P2, W, b, True)
def both(sa, sb):
    def run():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            agg()
        with torch.cuda.stream(sb):
            gemm()
        cur.wait_stream(sa); cur.wait_stream(sb)
    return run
res = {}
res["plain_streams_both"] = timeit(both(torch.cuda.Stream(), torch.cuda.Stream()))
for name, wa in (("half", [0x0000FFFF] * 8), ("3/8", [0x00000FFF] * 8), ("1/4", [0x000000FF] * 8), ("5/8", [0x000FFFFF] * 8)):
    wb = [(~w) & 0xFFFFFFFF for w in wa]
    sa, sb = masked_stream(wa), masked_stream(wb)
    def solo(s, fn):
        def run():
            cur = torch.cuda.current_stream(); s.wait_stream(cur)
            with torch.cuda.stream(s):
                fn()
            cur.wait_stream(s)
        return run
    res[name + "_agg_alone"] = timeit(solo(sa, agg))
    res[name + "_gemm_alone"] = timeit(solo(sb, gemm))
    res[name + "_both"] = timeit(both(sa, sb))
    print(name, {k: round(v, 2) for k, v in res.items() if k.startswith(name)}, flush=True)
print(json.dumps(res))

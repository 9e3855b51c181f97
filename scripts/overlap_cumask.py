"""Aggregation and MFMA transform side by side on disjoint compute units (CU-masked streams)."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import graphgen, ops
from graphgym_amd._lib import lib, check
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

def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    h = C.c_void_p()
    check(lib().mp_stream_create_cu_mask(words, 8, C.byref(h)), "mp_stream_create_cu_mask")
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

def on(stream, fn):
    def run():
        cur = torch.cuda.current_stream()
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            fn()
        cur.wait_stream(stream)
    return run

agg = lambda: ops._raw_spmm(g, x, 0, out=P)
gemm = lambda: ops._dense_into(out, P, W, b, True)
res = {"agg_ms": timeit(agg), "gemm_ms": timeit(gemm)}
ALL = (1 << 256) - 1
def pattern(name, frac_num, frac_den):
    if name == "lowbits":      # the first k bits
        k = 256 * frac_num // frac_den
        return (1 << k) - 1
    if name == "stride":       # frac_num of every frac_den consecutive bits
        m = 0
        for i in range(256):
            if i % frac_den < frac_num:
                m |= 1 << i
        return m
    if name == "per32":        # the first k of every 32 bits
        k = 32 * frac_num // frac_den
        m = 0
        for i in range(256):
            if i % 32 < k:
                m |= 1 << i
        return m
chunks = 16
bounds = [n * i // chunks for i in range(chunks + 1)]
parts = [g.row_slice(bounds[i], bounds[i + 1]) for i in range(chunks)]
for p in parts:
    p.plan()
evs = [torch.cuda.Event() for _ in range(chunks)]
for name in ("lowbits", "stride", "per32"):
    for num, den in ((1, 2), (5, 8), (3, 4)):
        ma = pattern(name, num, den)
        sa, sb = masked_stream(ma), masked_stream(ALL & ~ma)
        key = f"{name}_{num}/{den}"
        res[key + "_agg_alone"] = timeit(on(sa, agg))
        res[key + "_gemm_alone"] = timeit(on(sb, gemm))
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
        res[key + "_piped"] = timeit(piped)
        print(key, {k: round(v, 2) for k, v in res.items() if k.startswith(key)}, flush=True)
print(json.dumps(res))

"""Sweep: CUs per XCD given to the aggregation, chunk count, order; transform on the complement or unmasked."""
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
out = torch.empty(n, d, device=dev)
g.plan()
def masked_stream(bits):
    w = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    h = C.c_void_p()
    check(lib().mp_stream_create_cu_mask(w, 8, C.byref(h)), "mp_stream_create_cu_mask")
    return torch.cuda.ExternalStream(h.value)
ALL = (1 << 256) - 1
def timeit(fn, iters=6, warm=2):
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
def seq():
    ops._raw_spmm(g, x, 0, out=P); ops._dense_into(out, P, W, b, True)
print("sequential", round(timeit(seq), 2), flush=True)
ref = out.clone()
partsets = {}
for chunks in (16, 32, 64):
    bounds = [n * i // chunks // 128 * 128 for i in range(chunks)] + [n]
    parts = [g.row_slice(bounds[i], bounds[i + 1]) for i in range(chunks)]
    for p in parts:
        p.plan()
    partsets[chunks] = (bounds, parts)
for k in (12, 14, 16, 18, 20):
    ma = (1 << (8 * k)) - 1
    sa = masked_stream(ma)
    for gname, sb in (("compl", masked_stream(ALL & ~ma)), ("full", torch.cuda.Stream())):
        for chunks in (16, 32, 64):
            bounds, parts = partsets[chunks]
            evs = [torch.cuda.Event() for _ in range(chunks)]
            order = list(range(chunks))[::-1]
            def piped():
                cur = torch.cuda.current_stream()
                sa.wait_stream(cur); sb.wait_stream(cur)
                for i in order:
                    with torch.cuda.stream(sa):
                        ops._raw_spmm(parts[i], x, 0, out=P[bounds[i]:bounds[i + 1]])
                        evs[i].record(sa)
                    with torch.cuda.stream(sb):
                        sb.wait_event(evs[i])
                        ops._dense_into(out[bounds[i]:bounds[i + 1]], P[bounds[i]:bounds[i + 1]], W, b, True)
                cur.wait_stream(sa); cur.wait_stream(sb)
            out.zero_()
            t = timeit(piped)
            print(f"agg_cus={k} gemm={gname} chunks={chunks} reversed: {t:.2f} ms equal={bool(torch.equal(out, ref))}", flush=True)

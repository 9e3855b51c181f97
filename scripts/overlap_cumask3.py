"""Timeline of the chunked aggregate -> transform pipeline on CU-masked streams."""
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
def masked_stream(words):
    w = (C.c_uint32 * len(words))(*words)
    h = C.c_void_p()
    check(lib().mp_stream_create_cu_mask(w, len(words), C.byref(h)), "mp_stream_create_cu_mask")
    return torch.cuda.ExternalStream(h.value)
sa, sb = masked_stream([0x0000FFFF] * 8), masked_stream([0xFFFF0000] * 8)
rp = g.rowptr
for chunks, balanced in ((4, False), (4, True), (8, True), (16, True), (16, False)):
    if balanced:   # equal stored entries + rows cost per chunk
        cost = rp.to(torch.int64) + 4 * torch.arange(n + 1, device=dev)
        tgt = torch.arange(1, chunks, device=dev) * (int(cost[-1]) // chunks)
        cuts = torch.searchsorted(cost, tgt).tolist()
        bounds = [0] + [int(c) // 128 * 128 for c in cuts] + [n]
    else:
        bounds = [n * i // chunks for i in range(chunks + 1)]
    parts = [g.row_slice(bounds[i], bounds[i + 1]) for i in range(chunks)]
    for p in parts:
        p.plan()
    mk = lambda: [torch.cuda.Event(enable_timing=True) for _ in range(chunks)]
    a0, a1, b0, b1 = mk(), mk(), mk(), mk()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def piped(record):
        cur = torch.cuda.current_stream()
        if record: t0.record(cur)
        sa.wait_stream(cur); sb.wait_stream(cur)
        for i, p in enumerate(parts):
            with torch.cuda.stream(sa):
                a0[i].record(sa)
                ops._raw_spmm(p, x, 0, out=P[bounds[i]:bounds[i + 1]])
                a1[i].record(sa)
            with torch.cuda.stream(sb):
                sb.wait_event(a1[i])
                b0[i].record(sb)
                ops._dense_into(out[bounds[i]:bounds[i + 1]], P[bounds[i]:bounds[i + 1]], W, b, True)
                b1[i].record(sb)
        cur.wait_stream(sa); cur.wait_stream(sb)
        if record: t1.record(cur)
    for _ in range(3):
        piped(False)
    torch.cuda.synchronize()
    piped(True)
    torch.cuda.synchronize()
    print(f"chunks={chunks} balanced={balanced} total={t0.elapsed_time(t1):.2f} ms")
    print("  agg  :", " ".join(f"{t0.elapsed_time(a0[i]):.1f}-{t0.elapsed_time(a1[i]):.1f}" for i in range(chunks)))
    print("  gemm :", " ".join(f"{t0.elapsed_time(b0[i]):.1f}-{t0.elapsed_time(b1[i]):.1f}" for i in range(chunks)), flush=True)

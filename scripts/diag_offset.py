"""Kernel time vs the relative placement of X and Y in HBM (developer diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import ops, graphgen
dev = torch.device("cuda:0")
n, d = 10_000_000, 256
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
gu = ga.CSRGraph.from_edge_index(ei, n)
del ei
g.plan(); gu.plan()
torch.cuda.empty_cache()
row_bytes = d * 4
size = n * row_bytes
slack = 1 << 30
pool = torch.empty((3 * size + 4 * slack) // 4, dtype=torch.float32, device=dev)
base = pool.data_ptr()
def view(off_bytes):
    assert off_bytes % 16 == 0 and off_bytes + size <= pool.numel() * 4
    return pool[off_bytes // 4: off_bytes // 4 + n * d].view(n, d)
xoff = size + 2 * slack           # X in the middle of the pool
x = view(xoff)
x.uniform_(-1, 1)
def t(gr, xx, yy, k=4):
    ops._raw_spmm(gr, xx, 0, out=yy)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): ops._raw_spmm(gr, xx, 0, out=yy)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
print("pool base", hex(base), "x", hex(x.data_ptr()), flush=True)
# Y below X at distance size + delta, and above X at distance size + delta
for name, gr in (("loops+norm", g), ("no loops unweighted", gu)):
    for delta in [0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 64 << 20, 256 << 20, 1 << 30]:
        lo = view(xoff - size - delta)
        hi = view(xoff + size + delta)
        print(f"{name:20s} delta={delta:>10d}: Y below X {t(gr, x, lo):7.3f} ms   Y above X {t(gr, x, hi):7.3f} ms", flush=True)
# fine scan of Y above X in 1 KiB .. 64 KiB steps
for step, cnt in ((1024, 16), (16384, 16), (1 << 20, 16)):
    row = []
    for k in range(cnt):
        row.append(t(g, x, view(xoff + size + k * step), 3))
    print(f"scan step {step}: " + " ".join(f"{v:.2f}" for v in row), flush=True)

"""Kernel time vs (addr(X) - addr(Y)) modulo large powers of two (developer diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import ops, graphgen
dev = torch.device("cuda:0")
n, d = 10_000_000, 256
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
del ei
g.plan()
torch.cuda.empty_cache()
size = n * d * 4
MiB = 1 << 20
slack = 2 << 30
pool = torch.empty((3 * size + 4 * slack) // 4, dtype=torch.float32, device=dev)
base = pool.data_ptr()
def view_at(addr):
    off = addr - base
    assert off % 16 == 0 and 0 <= off and off + size <= pool.numel() * 4, (off,)
    return pool[off // 4: off // 4 + n * d].view(n, d)
def t(xx, yy, k=3):
    ops._raw_spmm(g, xx, 0, out=yy)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): ops._raw_spmm(g, xx, 0, out=yy)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
# X at an address that is 0 mod 64 MiB
xa = ((base + size + slack + 64 * MiB - 1) // (64 * MiB)) * (64 * MiB)
x = view_at(xa); x.uniform_(-1, 1)
S = ((size + 8 * MiB - 1) // (8 * MiB)) * (8 * MiB)      # size rounded up to 8 MiB
print("base", hex(base), "x", hex(xa), "S/8MiB", S // (8 * MiB), flush=True)
for unit, label in ((8 * MiB, "8MiB"), (2 * MiB, "2MiB"), (512 * 1024, "512KiB")):
    row = []
    for k in range(16):
        ya = xa - S - 64 * MiB + k * unit          # y below x; (x - y) = S + 64MiB - k*unit
        row.append((((xa - ya) // MiB) % 64, t(x, view_at(ya))))
    print(f"Y below X, step {label}: " + " ".join(f"[{m}MiB]{v:.2f}" for m, v in row), flush=True)
row = []
for k in range(16):
    ya = xa + S + k * 8 * MiB
    row.append((((ya - xa) // MiB) % 64, t(x, view_at(ya))))
print("Y above X, step 8MiB: " + " ".join(f"[{m}MiB]{v:.2f}" for m, v in row), flush=True)
# does X's own alignment matter?  shift X by 8 MiB steps with Y fixed relative (x - y = S + 8 MiB)
for k in range(4):
    xa2 = xa + k * 8 * MiB
    x2 = view_at(xa2); x2.copy_(x) if k else None
    print(f"x at +{8*k}MiB, y = x - S - 8MiB: {t(x2, view_at(xa2 - S - 8 * MiB)):.2f}   y = x - S - 24MiB: {t(x2, view_at(xa2 - S - 24 * MiB)):.2f}", flush=True)

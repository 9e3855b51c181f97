"""Placement bands: one 100 GB arena, X at its start, Y at increasing distances (developer diagnostic)."""
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
GB = 10 ** 9
pool = torch.empty(int(110 * GB) // 4, dtype=torch.float32, device=dev)
def view(off):
    off = (off // 256) * 256
    return pool[off // 4: off // 4 + n * d].view(n, d)
x = view(0); x.uniform_(-1, 1)
def t(xx, yy, k=3):
    ops._raw_spmm(g, xx, 0, out=yy)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): ops._raw_spmm(g, xx, 0, out=yy)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
print("pool", hex(pool.data_ptr()), flush=True)
for off_gb in (11, 14, 18, 22, 26, 30, 34, 38, 42, 46, 50, 54, 58, 62, 66, 70, 74, 78, 82, 86, 90, 94, 98):
    print(f"Y at +{off_gb:3d} GB: {t(x, view(off_gb * GB)):.2f} ms", flush=True)
# and X elsewhere
x2 = view(50 * GB); x2.copy_(x)
for off_gb in (0, 20, 36, 62, 80, 98):
    print(f"X at +50 GB, Y at +{off_gb:3d} GB: {t(x2, view(off_gb * GB)):.2f} ms", flush=True)

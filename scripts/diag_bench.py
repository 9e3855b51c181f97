"""Why does bench.py read 8 % slower than scripts/sweep.py on the same box?  (developer diagnostic)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import ops, graphgen, _lib
dev = torch.device("cuda:0")
n, d = 10_000_000, 256
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
del ei
g.plan()
def per_step(x, y, k=30):
    st = [torch.cuda.Event(enable_timing=True) for _ in range(k)]; en = [torch.cuda.Event(enable_timing=True) for _ in range(k)]
    for i in range(k):
        st[i].record(); ops._raw_spmm(g, x, 0, out=y); en[i].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in zip(st, en))
    return t[0], t[len(t)//2], t[-1]
def b2b(x, y, k=30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): ops._raw_spmm(g, x, 0, out=y)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
gen = torch.Generator(device=dev).manual_seed(7)
x = torch.rand((n, d), device=dev, generator=gen) * 2 - 1
y = torch.empty((n, d), dtype=torch.float32, device=dev)
torch.cuda.empty_cache()
print("mem", torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9, flush=True)
for _ in range(5): ops._raw_spmm(g, x, 0, out=y)
print("A bench-like x (generator, in-place mul/sub temp):", per_step(x, y), "b2b", b2b(x, y), flush=True)
x2 = torch.rand(n, d, device=dev) * 2 - 1
y2 = torch.empty(n, d, device=dev)
for _ in range(3): ops._raw_spmm(g, x2, 0, out=y2)
print("B fresh x2,y2:", per_step(x2, y2), "b2b", b2b(x2, y2), flush=True)
print("C x with y2:", per_step(x, y2), " x2 with y:", per_step(x2, y), flush=True)
print("ptrs", hex(x.data_ptr()), hex(y.data_ptr()), hex(x2.data_ptr()), hex(y2.data_ptr()), hex(g.col.data_ptr()), hex(g.val.data_ptr()), flush=True)
x3 = x.clone()
print("D clone of x:", per_step(x3, y), flush=True)

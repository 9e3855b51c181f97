"""Is the fast/slow split a property of memory placement?  Time the aggregation, a plain copy, a read and a
write for several separately allocated X / Y tensors (developer diagnostic)."""
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
def tm(fn, k=3):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
T = [torch.empty(n, d, device=dev) for _ in range(6)]
for t in T: t.uniform_(-1, 1)
print("addresses:", [hex(t.data_ptr()) for t in T], flush=True)
print("read  (sum)  ms:", " ".join(f"{tm(lambda t=t: t.sum()):.2f}" for t in T), flush=True)
print("write (fill) ms:", " ".join(f"{tm(lambda t=t: t.fill_(0.5)):.2f}" for t in T), flush=True)
for t in T: t.uniform_(-1, 1)
print("aggregation X=row, Y=col (ms):")
for i, xt in enumerate(T):
    print(f"  X{i}: " + " ".join("  -  " if i == j else f"{tm(lambda: ops._raw_spmm(g, xt, 0, out=T[j])):5.2f}" for j in range(6)), flush=True)
    xt.uniform_(-1, 1)
print("copy X=row -> Y=col (ms):")
for i, xt in enumerate(T):
    print(f"  X{i}: " + " ".join("  -  " if i == j else f"{tm(lambda: T[j].copy_(xt)):5.2f}" for j in range(6)), flush=True)

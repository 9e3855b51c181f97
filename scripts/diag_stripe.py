"""Layout experiment: rows of X striped over k 36-GB memory regions vs one region (timing only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import ops, graphgen, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
n, d = 10_000_000, 256
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
del ei
g.plan()
torch.cuda.empty_cache()
GB = 10 ** 9
REGION = 36 * GB
total_gb = int(sys.argv[1]) if len(sys.argv) > 1 else 250
pool = torch.empty(int(total_gb * GB) // 4, dtype=torch.float32, device=dev)
pool[: int(min(total_gb, 40) * GB) // 4].uniform_(-1, 1)
def view(off):
    off = (off // 256) * 256
    return pool[off // 4: off // 4 + n * d].view(n, d)
def t(xx, yy, k=3):
    ops._raw_spmm(g, xx, 0, out=yy)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): ops._raw_spmm(g, xx, 0, out=yy)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
x = view(0)
nreg = total_gb // 36
print("regions in arena:", nreg, flush=True)
y_last = view((nreg - 1) * REGION + 1 * GB)            # Y in the last region
print(f"plain X in region 0, Y in region {nreg-1}: {t(x, y_last):.2f} ms", flush=True)
for k in range(2, nreg):                              # X striped over regions 0..k-1, Y in the last region
    L.mp_spmm_debug_xregions(k, REGION // 4)
    print(f"X striped over {k} regions, Y in region {nreg-1}: {t(x, y_last):.2f} ms", flush=True)
L.mp_spmm_debug_xregions(nreg, REGION // 4)          # X over all regions, Y sharing the last one
print(f"X striped over all {nreg} regions (Y shares one): {t(x, y_last):.2f} ms", flush=True)
L.mp_spmm_debug_xregions(0, 0)

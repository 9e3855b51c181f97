import ctypes as C, gc, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import graphgym_amd as ga
from graphgym_amd import ops, placement, _lib
from graphgym_amd._lib import lib, check
dev = torch.device("cuda:0")
from test_fused_gpu import make_graph

def rel(a, r):
    return float((a.cpu().double() - r).abs().max() / max(1.0, float(r.abs().max())))

# ---- part 1: F=512 id backward pieces
n, E, F, d, n_id = 800, 9000, 512, 512, 64
ei, w = make_graph(n, E, seed=n + F + 1, hubs=False, weighted=True)
gen = torch.Generator().manual_seed(2)
x = torch.randn(n, F, generator=gen); W = torch.randn(F, d, generator=gen) / F ** 0.5
Wid = torch.randn(F, d, generator=gen) / F ** 0.5
ids = torch.randperm(n, generator=gen)[:n_id]
G = ga.CSRGraph.from_edge_index(ei.to(dev), n, w.to(dev), dst_row=0)
up = torch.randn(n, d, generator=gen)
A = torch.zeros(n, n, dtype=torch.float64).index_put_((ei[0], ei[1]), w.double(), accumulate=True)
gm = up.to(dev)
# main dx = (A^T g) W^T
ref_main = (A.t() @ up.double()) @ W.double().t()
gt = G.transpose()
Wt = W.t().contiguous().to(dev)
print("kernel ok for transposed:", ops._agg_dense_kernel_ok(gt, gm, Wt, None))
o1, _ = ops._raw_agg_dense(gt, gm, Wt)
print("fused dx_main err", rel(o1, ref_main))
T0, _ = ops._raw_spmm(gt, gm, 0)
print("two-kernel dx_main err", rel(T0 @ Wt, ref_main), "agg alone", rel(T0, A.t() @ up.double()))
o2 = torch.ops.mp.agg_dense_raw(gm, Wt, None, G.handle, 1, 0, None, 0.0, False, False)[0]
print("op dx_main err", rel(o2, ref_main))
# forward-direction fused at F=512 for comparison
o3, _ = ops._raw_agg_dense(G, x.to(dev), W.to(dev))
print("fused fwd err", rel(o3, (A @ x.double()) @ W.double()))
# T
S = torch.zeros(n, n, dtype=torch.float64); S[ids, ids] = 1
Tref = (A @ S).t()[ids] @ up.double()
br = G.id_branch(ids.to(dev))
T, _ = ops._raw_spmm(br.t, gm, 0)
print("T err", rel(T, Tref))
# various n to see if the fused kernel on this transposed graph has issues
for nn in (700, 800, 801, 1024):
    ei2, w2 = make_graph(nn, 9000, seed=nn, hubs=False, weighted=True)
    G2 = ga.CSRGraph.from_edge_index(ei2.to(dev), nn, w2.to(dev), dst_row=0)
    A2 = torch.zeros(nn, nn, dtype=torch.float64).index_put_((ei2[0], ei2[1]), w2.double(), accumulate=True)
    x2 = torch.randn(nn, 512, generator=gen)
    for gg, AA, nm in ((G2, A2, "fwd"), (G2.transpose(), A2.t(), "T")):
        o, _ = ops._raw_agg_dense(gg, x2.to(dev), W.to(dev))
        print(nn, nm, "err", rel(o, (AA @ x2.double()) @ W.double()))

# ---- part 2: placement prediction vs measurement
ar = placement.arena(dev)
GiB = 1 << 30
nrow = 8 * GiB // 1024
xx = ar.empty((nrow, 256)); xx.uniform_(-1, 1)
fp = ar._footprint(xx.data_ptr(), 8 * GiB)
pen = fp @ ar.conflict
print("x granules", np.nonzero(fp)[0], "pen min/median/max", pen.min(), np.median(pen), pen.max())
yy = ar.empty((nrow, 256), reads=(xx,))
print("y at granule", (yy.data_ptr() - ar.base) / placement.GRANULE, "predicted", getattr(yy, "_mp_predicted_conflict", None))
t_placed = [placement._probe(xx.data_ptr(), yy.data_ptr(), 8 * GiB, 3) for _ in range(3)]
print("placed times", t_placed)
del yy; gc.collect()
L = lib()
rows = []
for gidx in range(0, ar.n_gran - 2, 1):
    p = np.ones(ar.n_gran, dtype=np.float32); p[gidx:gidx + 2] = 0.0
    out = C.c_void_p()
    st = L.mp_arena_alloc_placed(8 * GiB, p.ctypes.data_as(C.c_void_p), ar.n_gran, placement.GRANULE, C.byref(out))
    if st != 0: continue
    off = (out.value - ar.base) / placement.GRANULE
    t = placement._probe(xx.data_ptr(), out.value, 8 * GiB, 3)
    pred = float(ar._footprint(out.value, 8 * GiB) @ pen)
    rows.append((off, t, pred))
    check(L.mp_arena_release(out))
best = min(r[1] for r in rows)
for off, t, pred in rows:
    print("gran %5.2f  t %.3f (+%4.1f%%)  predicted +%4.1f%%" % (off, t, 100 * (t / best - 1), 100 * pred))
np.set_printoptions(linewidth=250, precision=0, suppress=True)
print((100 * ar.conflict).astype(int))

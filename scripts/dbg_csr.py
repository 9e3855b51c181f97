import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphgym_amd as ga
from graphgym_amd import graphgen
from graphgym_amd.ego import ego_batch
dev = torch.device("cuda:0")
n = 20_000
ei0 = graphgen.ba_edge_index(n, 4, seed=1, device=dev)
base = ga.CSRGraph.from_edge_index(ei0, n)
cen = torch.randint(0, n, (300,), generator=torch.Generator().manual_seed(5)).to(dev)
cen[:3] = torch.tensor([0, 1, 2], device=dev)
ei, orig, ids, ego_of, g = ego_batch(base, cen, 1, csr="add")
n2 = orig.numel()
want = ga.CSRGraph.from_edge_index(ei, n2, add_self_loops=True)
cg = (g.rowptr[1:] - g.rowptr[:-1]); cw = (want.rowptr[1:] - want.rowptr[:-1])
bad = torch.nonzero(cg != cw).view(-1)
print("rows differing:", bad.numel(), bad[:10].tolist())
for r in bad[:10].tolist():
    print(r, "got", int(cg[r]), "want", int(cw[r]), "orig", int(orig[r]), "ego", int(ego_of[r]), "centre", int(cen[ego_of[r]]))
print("dup centres:", cen.numel() - torch.unique(cen).numel())
ei_n, _, _, _, gn = ego_batch(base, cen, 1, csr="none")
print("edge lists equal:", torch.equal(ei, ei_n))

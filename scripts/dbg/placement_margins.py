import gc, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import graphgym_amd as ga
from graphgym_amd import graphgen, ops, placement
dev = torch.device("cuda:0")
ar = placement.arena(dev)
n, d = 1 << 22, 256
ei = graphgen.ba_edge_index(n, 5, seed=3, device=dev)
g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row"); del ei; g.plan()
x = ar.empty((n, d)); x.uniform_(-1, 1)
def agg_ms(y):
    ops._raw_spmm(g, x, 0, out=y); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops._raw_spmm(g, x, 0, out=y); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
out = {}
for trial in range(3):
    y = ar.empty((n, d), reads=(x,)); tp = agg_ms(y); del y; gc.collect()
    yv = ar.empty((n, d), reads=(x,), verify="all"); tv = agg_ms(yv); del yv; gc.collect()
    y4 = ar.empty((n, d), reads=(x,), verify=4); t4 = agg_ms(y4); del y4; gc.collect()
    times = []
    for gi in range(ar.n_gran):
        yc = ar.empty_at((n, d), gi)
        if yc is None: continue
        times.append(agg_ms(yc)); del yc
    b, w, m = min(times), max(times), sorted(times)[len(times)//2]
    print(json.dumps({"trial": trial, "pred": round(tp/b, 4), "verified_all": round(tv/b, 4), "verified_4": round(t4/b, 4),
                      "median": round(m/b, 4), "worst": round(w/b, 4), "best_ms": round(b, 3), "n": len(times)}), flush=True)

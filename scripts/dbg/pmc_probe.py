"""Workload for PMC passes: the plain aggregation and the one-kernel layer at the C4 shape, each with the output in a
fast-band and in a slow-band position relative to X (scripts/dbg/buffer_quality.py: buffers 0 / 8 vs 0 / 1).
Order of the big launches: agg(good) x2, agg(bad) x2, fused(good) x2, fused(bad) x2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops
dev = torch.device("cuda:0")
n, d = 10_000_000, 256
g = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n, 5, 12345, device=dev), n, add_self_loops=True).gcn_norm("row")
g.plan()
torch.cuda.empty_cache()
bufs = [torch.empty((n, d), device=dev) for _ in range(9)]
x, y_good, y_bad = bufs[0].uniform_(-1, 1), bufs[8], bufs[1]
W = torch.randn(d, d, device=dev) * 0.05
b = torch.randn(d, device=dev)
Wsp = ops._split_w(W)
torch.cuda.synchronize()
for y in (y_good, y_bad):
    for _ in range(2):
        ops._raw_spmm(g, x, 0, out=y)
for y in (y_good, y_bad):
    for _ in range(2):
        ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=True)
torch.cuda.synchronize()

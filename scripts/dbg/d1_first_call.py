"""Why does the first d=1 unweighted aggregation of a process take 130 ms (profiles/r02_gin_step_kernels.csv)?
Times successive launches on a fresh graph, with and without other work in front."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops
dev = torch.device("cuda:0")
n = int(os.environ.get("NODES", "10000000"))
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
def timed(f, tag):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); y = f(); e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1):.3f} ms", flush=True)
    return y
for weighted in (False, True):
    g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True)
    if weighted:
        g = g.gcn_norm("row")
    for kind in ("ones", "rand"):
        x = torch.ones(n, 1, device=dev) if kind == "ones" else torch.rand(n, 1, device=dev)
        for i in range(4):
            timed(lambda: ops.spmm(g, x, "sum"), f"weighted={weighted} x={kind} call {i}")
    xr = torch.ones(n, 1, device=dev, requires_grad=True)
    for i in range(3):
        y = timed(lambda: ops.spmm(g, xr, "sum"), f"weighted={weighted} grad fwd {i}")
        timed(lambda: y.sum().backward(), f"weighted={weighted} grad bwd {i}")

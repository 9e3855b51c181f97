"""One GCN layer (C2 size; NODES=10000000 for C4) on the engine, aggregate then transform: by default the
one-kernel path (mp_agg_dense_f32); MP_FUSED=0 gives the aggregation kernel followed by the MFMA transform kernel.
Run under rocprofv3 (scripts/prof_layer.sh) to get per-kernel time and MFMA counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, layers

dev = torch.device("cuda:0")
n, d = int(os.environ.get("NODES", "1000000")), 256
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
layer = layers.GCNConvLayer(d, d, bias=True, order="aggregate_first").to(dev)
x = torch.rand(n, d, device=dev) * 2 - 1
holder = type("H", (), {})()
with torch.no_grad():
    for _ in range(int(os.environ.get("ITERS", "10"))):
        y = layer(x, ei, holder=holder)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))

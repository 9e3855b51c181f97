"""which aten ops fill / zero large tensors inside a training step (harness model KIND, small graph)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from graphgym_amd import graphgen, harness as H
dev = torch.device("cuda:0")
n, d = 700_000, 256
kind = os.environ.get("KIND", "gcn")
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
x = torch.ones(n, 1, device=dev); labels = torch.randint(0, 10, (n,), device=dev); idx = torch.arange(n, device=dev)
model = H.TfgNodeModel(kind, 1, d, 10).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01)
holder = H.Batch()
def fl():
    return H.tfg_loss(model([x, ei], holder=holder), idx, labels, model.kernel_parameters())
H.train_step(model, opt, fl)
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in ("zero", "fill", "full", "ones", "threshold", "add", "mul", "copy", "clone", "contiguous", "cat")):
            def sz(t):
                return tuple(t.shape) if isinstance(t, torch.Tensor) else None
            big = [sz(a) for a in args if isinstance(a, torch.Tensor) and a.numel() >= n]
            o = out if isinstance(out, torch.Tensor) else None
            if big or (o is not None and o.numel() >= n):
                print(name, big, None if o is None else tuple(o.shape), flush=True)
        return out
with Spy():
    H.train_step(model, opt, fl)

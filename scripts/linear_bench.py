"""graphgym_amd.nn.Linear against torch.nn.Linear, forward + backward, at the node counts of the C4 / C5 configs."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import nn as mpnn
dev = torch.device("cuda:0")
def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, fi, fo) in [(10_000_000, 256, 256), (10_000_000, 256, 10), (10_000_000, 1, 256), (10_000_000, 256, 128),
                    (150_000, 512, 512), (150_000, 512, 10), (65_000, 128, 128)]:
    x = torch.randn(M, fi, device=dev, requires_grad=True)
    up = torch.randn(M, fo, device=dev)
    res = {"M": M, "in": fi, "out": fo}
    for name, lin in (("torch", torch.nn.Linear(fi, fo).to(dev)), ("engine", mpnn.Linear(fi, fo).to(dev))):
        res[name + "_fwd_ms"] = timeit(lambda: lin(x))
        def fb():
            y = lin(x)
            y.backward(up)
            x.grad = None
        res[name + "_fwd_bwd_ms"] = timeit(fb)
    print(json.dumps(res), flush=True)
    del x, up

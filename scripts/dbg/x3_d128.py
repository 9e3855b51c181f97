"""dense_x3 at F = 256 -> d = 128 (the MeanGraphSage halves): contiguous and strided outputs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graphgym_amd import ops
dev = torch.device("cuda:0")
def tm(fn, k=5):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
M = 10_000_000
for F, d in ((256, 128), (128, 128), (256, 64), (128, 256)):
    P = torch.rand(M, F, device=dev) - 0.5
    W = (torch.rand(F, d, device=dev) - 0.5) / 8
    b = torch.rand(d, device=dev)
    out = torch.empty(M, d, device=dev)
    big = torch.empty(M, 2 * d, device=dev)
    print(f"F={F} d={d}: contiguous out {tm(lambda: ops._raw_dense_x3(P, W, b, True, out=out)):.2f} ms | "
          f"left half of a [M, {2 * d}] buffer {tm(lambda: ops._raw_dense_x3(P, W, b, True, out=big[:, :d])):.2f} ms | "
          f"right half {tm(lambda: ops._raw_dense_x3(P, W, b, True, out=big[:, d:])):.2f} ms", flush=True)
    del P, out, big

"""first-layer weight gradient (F = 1): dW = x^T (g * [y > 0]) at 1e7 x 1 x 256, with and without the masked-gradient output"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graphgym_amd import ops
dev = torch.device("cuda:0")
M, d = 10_000_000, 256
P = torch.ones(M, 1, device=dev); G = torch.randn(M, d, device=dev); Y = torch.relu(torch.randn(M, d, device=dev))
def tm(fn, k=5):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
for F in (1, 8):
    Pf = torch.randn(M, F, device=dev)
    for gm in (True, False):
        t = tm(lambda: ops._raw_dense_wgrad_relu(Pf, G, Y, want_bias=True, want_gm=gm))
        gb = M * d * 4 * (3 if gm else 2) / 1e9
        print(f"F={F} masked-gradient output={gm}: {t:.2f} ms ({gb / t:.2f} TB/s)", flush=True)
    t = tm(lambda: ops._raw_dense_wgrad(Pf, G, want_bias=True))
    print(f"F={F} no mask: {t:.2f} ms ({M * d * 4 / 1e9 / t:.2f} TB/s)", flush=True)

"""weight gradient dW = P^T g (with / without the ReLU mask) at the step's shapes"""
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
for M, F, d in ((10_000_000, 256, 256), (10_000_000, 128, 128), (10_000_000, 256, 128), (10_000_000, 512, 256), (1_000_000, 256, 256)):
    if M * (F + 3 * d) * 4 > 100e9:
        continue
    P = torch.randn(M, F, device=dev); G = torch.randn(M, d, device=dev); Y = torch.relu(torch.randn(M, d, device=dev))
    t0 = tm(lambda: ops._raw_dense_wgrad(P, G, want_bias=True))
    t1 = tm(lambda: ops._raw_dense_wgrad_relu(P, G, Y, want_bias=True))
    t2 = tm(lambda: ops._raw_dense_wgrad_relu(P, G, Y, want_bias=True, want_gm=False))
    print(f"M={M} F={F} d={d}: plain {t0:.2f} ms | ReLU mask + masked-gradient output {t1:.2f} ms | mask only {t2:.2f} ms "
          f"({2.0 * M * F * d / t0 / 1e9:.0f} TF plain)", flush=True)
    del P, G, Y
    torch.cuda.empty_cache()

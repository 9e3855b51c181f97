"""Fused post-aggregation transform (mp_dense_fused_f32, f32 MFMA) vs the library composition
(two torch.matmul + add + bias + relu), developer benchmark."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import ops, _lib
dev = torch.device("cuda:0")
def tm(fn, k=5):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
for M in (1_000_000, 10_000_000):
    for F, d in ((256, 256), (128, 128), (512, 512), (64, 256)):
        P = torch.rand(M, F, device=dev) - 0.5; Q = torch.rand(M, F, device=dev) - 0.5
        W = torch.rand(F, d, device=dev) - 0.5; Wi = torch.rand(F, d, device=dev) - 0.5; b = torch.rand(d, device=dev)
        t_single_lib = tm(lambda: torch.relu(torch.addmm(b, P, W)))
        t_single = tm(lambda: ops._raw_dense_fused(P, W, None, None, b, True))
        t_dual_lib = tm(lambda: torch.relu(P @ W + Q @ Wi + b))
        t_dual = tm(lambda: ops._raw_dense_fused(P, W, Q, Wi, b, True))
        g = torch.rand(M, d, device=dev) - 0.5
        t_wg_lib = tm(lambda: P.t() @ g)
        t_wg = tm(lambda: ops._raw_dense_wgrad(P, g))
        del g
        fl = 2.0 * M * F * d
        print(f"M={M} F={F} d={d}: single lib {t_single_lib:7.2f} ms ({fl/t_single_lib/1e9:6.1f} TF) fused {t_single:7.2f} ms ({fl/t_single/1e9:6.1f} TF) | "
              f"dual lib {t_dual_lib:7.2f} ms fused {t_dual:7.2f} ms ({2*fl/t_dual/1e9:6.1f} TF) | wgrad P^T g: lib {t_wg_lib:7.2f} ms engine {t_wg:7.2f} ms ({fl/t_wg/1e9:6.1f} TF)", flush=True)
        del P, Q
        torch.cuda.empty_cache()

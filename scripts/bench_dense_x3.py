"""The streaming transform (mp_dense_x3_f32) against float64, the general MFMA kernel (mp_dense_fused_f32) and the
library GEMM: correctness at ragged sizes, then time at the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)


def tm(fn, k=5):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k


ops.X3_MIN_ROWS = 1
for M, F, d, relu, trans in ((1, 64, 64, False, False), (255, 64, 128, True, False), (257, 256, 256, True, False),
                             (70001, 256, 256, False, True), (300000, 96, 256, True, False),
                             (131072, 512, 64, False, False)):
    P = torch.randn(M, F, device=dev)
    W = torch.randn(d, F, device=dev) / 8 if trans else torch.randn(F, d, device=dev) / 8
    b = torch.randn(d, device=dev)
    assert ops.dense_x3_supported(P, F, d)
    out = ops._raw_dense_x3(P, W, b, relu, trans=trans)
    ref = P.double() @ (W.double().t() if trans else W.double()) + b.double()
    if relu:
        ref = torch.relu(ref)
    err = float((out.double() - ref).abs().max()) / float(ref.abs().max())
    lib_err = float(((P @ (W.t() if trans else W) + b).double() - (ref if not relu else ref)).abs().max()) / float(ref.abs().max()) if not relu else float("nan")
    print(f"M={M} F={F} d={d} relu={relu} trans={trans}: max err / max |ref| = {err:.2e} (library fp32: {lib_err:.2e})", flush=True)
    assert err < 2e-6, err
print("correct", flush=True)

for M in (131072, 500_000, 1_000_000, 10_000_000):
    for F, d in ((256, 256), (128, 128), (256, 64), (512, 256)):
        if M * (F + d) * 4 > 60e9:
            continue
        P = torch.rand(M, F, device=dev) - 0.5
        W = (torch.rand(F, d, device=dev) - 0.5) / 8
        b = torch.rand(d, device=dev)
        out = torch.empty(M, d, device=dev)
        t_x3 = tm(lambda: ops._raw_dense_x3(P, W, b, True, out=out))
        os.environ["MP_X3"] = "0"
        t_gen = tm(lambda: ops._raw_dense_fused(P, W, None, None, b, True))
        os.environ["MP_X3"] = "1"
        t_lib = tm(lambda: torch.mm(P, W, out=out))
        gb = M * (F + d) * 4 / 1e9
        print(f"M={M} F={F} d={d}: streaming {t_x3:7.3f} ms ({gb / t_x3:5.2f} TB/s, {2.0 * M * F * d / t_x3 / 1e9:6.1f} TF) | "
              f"general {t_gen:7.3f} ms | library mm {t_lib:7.3f} ms", flush=True)
        del P, out
        torch.cuda.empty_cache()

"""A/B of dense_x3_pc_kernel (d = 256) with and without the software-pipelined split of the P fragment
(MP_X3_PIPE=1): same process, same buffers; both checked against float64 first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)


def tm(fn, k=10):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k


ops.X3_MIN_ROWS = 1
for t64 in ("0", "1"):
    os.environ["MP_X3_PIPE"] = t64
    for M, F, relu in ((257, 256, True), (70001, 256, False), (300001, 96, True), (1_000_003, 512, True)):
        P = torch.randn(M, F, device=dev)
        W = torch.randn(F, 256, device=dev) / 8
        b = torch.randn(256, device=dev)
        out = ops._raw_dense_x3(P, W, b, relu)
        ref = P.double() @ W.double() + b.double()
        if relu:
            ref = torch.relu(ref)
        err = float((out.double() - ref).abs().max()) / float(ref.abs().max())
        print(f"PIPE={t64} M={M} F={F}: err {err:.2e}", flush=True)
        assert err < 2e-6
        del P, out, ref
for M in (1_000_000, 10_000_000):
    for F in (256, 512):
        P = torch.rand(M, F, device=dev) - 0.5
        W = (torch.rand(F, 256, device=dev) - 0.5) / 8
        b = torch.rand(256, device=dev)
        out = torch.empty(M, 256, device=dev)
        r = {}
        for rep in range(2):
            for t64 in ("0", "1"):
                os.environ["MP_X3_PIPE"] = t64
                r[t64] = min(r.get(t64, 1e9), tm(lambda: ops._raw_dense_x3(P, W, b, True, out=out)))
        print(f"M={M} F={F} d=256: 32x256 {r['0']:.3f} ms | pipelined {r['1']:.3f} ms", flush=True)
        del P, out
        torch.cuda.empty_cache()

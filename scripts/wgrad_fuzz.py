"""Randomised shapes for the weight-gradient kernels (mp_dense_wgrad_f32 / mp_dense_wgrad_relu_f32; F > 128 with F, d
multiples of 8 runs the loader / MFMA-wave kernel) against a float64 evaluation; bitwise reproducibility checked."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(os.environ.get("SEED", "0")))
def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))
worst = 0.0
for case in range(int(os.environ.get("CASES", "200"))):
    M = ri(1, 300) if case % 4 == 0 else (ri(300, 20000) if case % 4 < 3 else ri(20000, 300000))
    F = 8 * ri(1, 75) if case % 5 else ri(1, 600)
    d = 8 * ri(1, 66) if case % 5 else ri(1, 520)
    if M * (F + 3 * d) * 4 > 2e9:
        continue
    P = torch.randn(M, F, generator=g).to(dev)
    G = torch.randn(M, d, generator=g).to(dev)
    relu = bool(ri(0, 1))
    if relu:
        Y = torch.relu(torch.randn(M, d, generator=g)).to(dev)
        want_gm = bool(ri(0, 1))
        dW, db, gm = ops._raw_dense_wgrad_relu(P, G, Y, want_bias=True, want_gm=want_gm)
        dW2, db2, _ = ops._raw_dense_wgrad_relu(P, G, Y, want_bias=True, want_gm=want_gm)
        Gm = torch.where(Y > 0, G, torch.zeros_like(G))
        if want_gm:
            assert torch.equal(gm, Gm), ("masked gradient", case, M, F, d)
    else:
        dW, db = ops._raw_dense_wgrad(P, G, want_bias=True)
        dW2, db2 = ops._raw_dense_wgrad(P, G, want_bias=True)
        Gm = G
    assert torch.equal(dW, dW2) and torch.equal(db, db2), ("not reproducible", case, M, F, d)
    ref = P.double().t() @ Gm.double()
    mag = P.double().abs().t() @ Gm.double().abs()
    eW = float(((dW.double() - ref).abs() / mag.clamp_min(1e-30)).max())
    rb = Gm.double().sum(0)
    eb = float(((db.double() - rb).abs() / Gm.double().abs().sum(0).clamp_min(1e-30)).max())
    worst = max(worst, eW, eb)
    if eW > 1e-5 or eb > 1e-5:
        print("MISMATCH", case, M, F, d, relu, eW, eb, flush=True)
        sys.exit(1)
print("cases ok, worst relative error", worst)

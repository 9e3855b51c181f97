"""graphgym_amd.nn.BatchNorm1d (training mode on the engine's kernels) vs torch.nn.BatchNorm1d in float64 on
the host: outputs, input / affine gradients, running statistics; with and without the fused ReLU; widths that
take the vector and the scalar paths; a column with a large mean (shifted sums must not cancel).
Tolerances: tests/_tol.py (per row / per statistic 1e-5 against float64, or twice torch's own float32 CPU BatchNorm's
distance from it)."""
import pytest
import torch

from _tol import both, close, close_all

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,d", [(2, 4), (1000, 256), (4097, 100), (333, 7), (20000, 512), (50000, 64)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_training_parity(dev, N, d, relu):
    from graphgym_amd.nn import BatchNorm1d
    g = torch.Generator().manual_seed(N + d)
    x = torch.randn(N, d, generator=g) * (torch.rand(d, generator=g) * 3 + 0.1) + torch.randn(d, generator=g) * 5
    x[:, 0] += 1000.0                                              # large mean, unit variance
    dy = torch.randn(N, d, generator=g)
    ours = BatchNorm1d(d, eps=1e-5, momentum=0.1, relu=relu).to(dev)
    w, b = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g)
    with torch.no_grad():
        ours.weight.copy_(w); ours.bias.copy_(b)
    xg = x.to(dev).requires_grad_(True)
    y = ours(xg)
    y.backward(dy.to(dev))
    mask = (y.detach() > 0).cpu() if relu else None

    def ref_fn(c):      # torch.nn.BatchNorm1d on the CPU (graphgym/models/layer.py:26-35), train step then eval
        ref = torch.nn.BatchNorm1d(d, eps=1e-5, momentum=0.1).to(c(x).dtype)
        with torch.no_grad():
            ref.weight.copy_(c(w)); ref.bias.copy_(c(b))
        xr = c(x).clone().requires_grad_(True)
        yr = ref(xr)
        if relu:        # through the engine's activation pattern (an input within rounding of 0 has no defined subgradient)
            yr = yr * mask.to(yr.dtype)
        yr.backward(c(dy))
        ref.eval()
        ye = ref(c(x))
        return (yr.detach(), xr.grad, ref.weight.grad, ref.bias.grad, ref.running_mean.clone(), ref.running_var.clone(),
                (torch.relu(ye) if relu else ye).detach())
    r64, r32 = both(ref_fn)
    if relu:
        pre = torch.nn.functional.batch_norm(x.double(), None, None, w.double(), b.double(), True, 0.1, 1e-5)
        off = (pre > 0) != mask
        assert not bool(off.any()) or float(pre.abs()[off].max()) <= 1e-5 * float(pre.abs().max())
    # column 0 is a stress column (mean 1000, unit variance): ANY float32 BatchNorm — torch's included — rounds that mean
    # to float32 (half an ulp of 1000 = 3e-5 of the column's standard deviation), and the column sits in every row, so
    # every row may need the float32 reference's own distance here; the median-ratio rule still holds the engine to it
    close(y, (r64[0], r32[0]), what="bn forward", max_ref32_frac=1.0)
    if N >= 8:
        close(xg.grad, (r64[1], r32[1]), what="bn dx", max_ref32_frac=1.0)
    else:   # a batch of two: x_hat = +-1 and dx is the rounding residue of terms that cancel exactly; one scale
        close_all(xg.grad, (r64[1], r32[1]), what="bn dx (degenerate batch)")
    col = lambda t: t.detach().reshape(-1, 1)                      # one statistic per column: each its own scale
    close_all(ours.weight.grad, (r64[2], r32[2]), what="bn dgamma")  # parameter gradients: reductions over all rows,
    close_all(ours.bias.grad, (r64[3], r32[3]), what="bn dbeta")     # signed sums that cancel: one scale (tests/_tol.py)
    close(col(ours.running_mean), (col(r64[4]), col(r32[4])), what="running_mean")
    close(col(ours.running_var), (col(r64[5]), col(r32[5])), what="running_var")
    assert int(ours.num_batches_tracked) == 1
    ours.eval()                                                    # eval: running statistics, library path
    close(ours(x.to(dev)), (r64[6], r32[6]), what="bn eval", max_ref32_frac=1.0)


def test_state_dict_interchanges_with_torch(dev):
    from graphgym_amd.nn import BatchNorm1d
    a, b = BatchNorm1d(16, relu=True), torch.nn.BatchNorm1d(16)
    assert set(a.state_dict()) == set(b.state_dict())
    b.load_state_dict(a.state_dict())


def test_engine_linear_matches_torch_linear(dev):
    """graphgym_amd.nn.Linear: same parameters / state dict as torch.nn.Linear; forward and backward against a
    float64 evaluation on both sides of its size switch (engine kernel on wide outputs over many rows, library
    otherwise), errors measured against the sum of absolute terms of each result"""
    import torch
    from graphgym_amd import nn as mpnn
    for (M, fi, fo, relu) in [(70000, 64, 128, False), (70000, 64, 128, True), (3000, 64, 128, True), (70000, 32, 10, False)]:
        torch.manual_seed(1)
        ref = torch.nn.Linear(fi, fo).to(dev)
        lin = mpnn.Linear(fi, fo, relu=relu).to(dev)
        assert set(lin.state_dict()) == set(ref.state_dict())
        lin.load_state_dict(ref.state_dict())
        x = torch.randn(M, fi, device=dev, requires_grad=True)
        up = torch.randn(M, fo, device=dev)
        y = lin(x)
        y.backward(up)
        xd, Wd, bd = x.detach().double(), ref.weight.detach().double(), ref.bias.detach().double()
        pre = xd @ Wd.t() + bd
        g = up.double() * ((y.detach() > 0) if relu else 1.0)    # the mask the forward pass produced (pre ~ 0 can flip in f64)
        checks = [(y, torch.relu(pre) if relu else pre, xd.abs() @ Wd.abs().t() + bd.abs()),
                  (x.grad, g @ Wd, g.abs() @ Wd.abs()),
                  (lin.weight.grad, g.t() @ xd, g.abs().t() @ xd.abs()),
                  (lin.bias.grad, g.sum(0), g.abs().sum(0))]
        for got, want, mag in checks:
            assert float(((got.double() - want).abs() / mag.clamp_min(1e-30)).max()) < 1e-5


@pytest.mark.parametrize("N,d", [(1000, 256), (4097, 100), (333, 7), (50000, 64)])
@pytest.mark.parametrize("affine", [(True, True), (True, False), (False, False)])
def test_relu_mask_recomputed_from_x_is_the_forwards(dev, N, d, affine):
    """mp_bn_train_bwd_relu_f32 (mask recomputed from x, y not read) gives the bits of mp_bn_train_bwd_f32 with the
    forward's output as the mask: dx, dgamma and dbeta are torch.equal — including inputs that sit on the ReLU edge"""
    g = torch.Generator().manual_seed(N * 7 + d)
    x = (torch.randn(N, d, generator=g) * 2 + torch.randn(d, generator=g)).to(dev)
    w = (torch.rand(d, generator=g) + 0.5).to(dev) if affine[0] else None
    b = (torch.randn(d, generator=g) * 0.1).to(dev) if affine[1] else None
    dy = torch.randn(N, d, generator=g).to(dev)
    y, mean, invstd, _ = torch.ops.mp.bn_fwd_raw(x, w, b, 1e-5, True)
    assert 0.2 < float((y > 0).float().mean()) < 0.8
    a = torch.ops.mp.bn_bwd_raw(dy, y, x, w, mean, invstd)
    c = torch.ops.mp.bn_bwd_raw(dy, None, x, w, mean, invstd, b, True)
    for u, v, name in zip(a, c, ("dx", "dgamma", "dbeta")):
        assert torch.equal(u, v), name

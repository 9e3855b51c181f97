"""The streaming dense transform (mp_dense_x3_f32 + mp_split_w_bf16x3, graphgym_amd/csrc/dense_x3.hip) against the
float64 product — the Linear / kernel product of the layers (TfgIDLayer.py:510-523, layer.py:136-147,
idconv.py:371-399) and its input gradient.  Tolerance: 1e-5 of the largest |reference| per output row (tests/_tol.py);
the fp32 library product is measured beside it for scale."""
import pytest
import torch

from graphgym_amd import ops
from graphgym_amd import _lib
from _tol import assert_close_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _all_sizes(monkeypatch):
    monkeypatch.setattr(ops, "X3_MIN_ROWS", 1)


def _ref(P, W, b, relu, trans):
    out = P.double() @ (W.double().t() if trans else W.double())
    if b is not None:
        out = out + b.double()
    return torch.relu(out) if relu else out


@pytest.mark.parametrize("M,F,d,relu,trans,bias", [
    (1, 64, 64, False, False, True),          # one row: every other lane of the block is padding
    (255, 64, 128, True, False, True),        # just under one row block
    (257, 256, 256, True, False, True),       # one row into the second block
    (1000, 96, 256, False, False, False),     # F a multiple of 32 that is not a power of two, no bias
    (70001, 256, 256, False, True, False),    # the input-gradient form: P @ W^T
    (300000, 160, 64, True, False, True),     # more row blocks than workgroups (persistent loop, seams)
    (131072, 512, 128, False, False, True),
])
def test_dense_x3_matches_float64(M, F, d, relu, trans, bias):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + F)
    P = torch.randn(M, F, generator=g).to(dev)
    W = (torch.randn(d, F, generator=g) / 8 if trans else torch.randn(F, d, generator=g) / 8).to(dev)
    b = torch.randn(d, generator=g).to(dev) if bias else None
    assert ops.dense_x3_supported(P, F, d)
    out = ops._raw_dense_x3(P, W, b, relu, trans=trans)
    ref = _ref(P, W, b, relu, trans)
    assert_close_rows(out, ref, 1e-5, what="dense_x3")


def test_dense_x3_strided_views():
    """leading dimensions: P a column slice of a wider matrix, out a column slice of a wider buffer (concat layers)"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    big = torch.randn(5000, 192, generator=g).to(dev)
    P = big[:, 64:192]                                   # [5000, 128], stride 192, 16-byte aligned
    W = (torch.randn(128, 64, generator=g) / 8).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    buf = torch.zeros(5000, 128, device=dev)
    view = buf[:, 64:]
    assert ops.dense_x3_supported(P, 128, 64, view)
    ops._raw_dense_x3(P, W, b, True, out=view)
    ref = _ref(P, W, b, True, False)
    assert_close_rows(view, ref, 1e-5, what="dense_x3 into a view")
    assert float(buf[:, :64].abs().max()) == 0.0         # nothing written outside the view


def test_split_w_layout_and_values():
    """W_split[s][k / 8][c][k % 8] = plane s of W[k][c]; the three planes sum back to W to 2^-24"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    W = torch.randn(64, 96, generator=g).to(dev)
    for trans in (False, True):
        sp = ops._split_w(W.t().contiguous() if trans else W, trans)       # both describe B = W
        assert sp.shape == (3, 8, 96, 8) and sp.dtype == torch.bfloat16
        planes = sp.float().permute(0, 1, 3, 2).reshape(3, 64, 96)
        assert torch.equal(planes[0], W.to(torch.bfloat16).float())
        back = planes.double().sum(0)
        assert float((back - W.double()).abs().max()) <= 2.0 ** -22 * float(W.abs().max())


def test_dense_x3_rejects_what_it_does_not_cover():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    P = torch.zeros(64, 96, device=dev)
    sp = torch.zeros(3, 12, 64, 8, dtype=torch.bfloat16, device=dev)
    out = torch.zeros(64, 96, device=dev)
    ptr = _lib.ptr
    assert L.mp_dense_x3_f32(ptr(P), 96, ptr(sp), None, 0, ptr(out), 96, 64, 96, 96, None) == 2      # d = 96
    assert L.mp_dense_x3_f32(ptr(P), 96, ptr(sp), None, 0, ptr(out), 64, 64, 32, 64, None) == 2      # F < 64
    assert L.mp_dense_x3_f32(ptr(P), 97, ptr(sp), None, 0, ptr(out), 64, 64, 96, 64, None) == 5      # ldp % 4
    assert L.mp_dense_x3_f32(ptr(P), 96, ptr(sp), None, 7, ptr(out), 64, 64, 96, 64, None) == 1      # bad activation
    assert L.mp_dense_x3_f32(ptr(P), 96, ptr(sp), None, 0, ptr(out), 64, 0, 96, 64, None) == 0       # no rows: no-op


def test_layers_route_through_the_streaming_kernel():
    """torch.ops.mp.dense_fused and its backward use mp_dense_x3_f32 at its shapes and agree with autograd on float64"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    P = torch.randn(3000, 128, generator=g).to(dev).requires_grad_(True)
    W = (torch.randn(128, 256, generator=g) / 8).to(dev).requires_grad_(True)
    b = torch.randn(256, generator=g).to(dev).requires_grad_(True)
    out = ops.dense_fused(P, W, bias=b, relu=True)
    go = torch.randn(3000, 256, generator=g).to(dev)
    out.backward(go)
    Pd, Wd, bd = (t.detach().double().requires_grad_(True) for t in (P, W, b))
    pre = Pd @ Wd + bd
    # the engine's own activation pattern (tests/_tol.py policy: borderline pre-activations may flip)
    mask = (out.detach() > 0).double()
    ref = pre * mask
    ref.backward(go.double())
    assert_close_rows(out, torch.relu(pre.detach()), 1e-5, what="forward")
    assert_close_rows(P.grad, Pd.grad, 1e-5, what="dP")
    assert_close_rows(W.grad, Wd.grad, 1e-5, ref32=(P.detach().t() @ (go * mask.float())), what="dW")

"""Operators around the aggregation inside a training step: the weight-gradient pass with the ReLU backward folded in
(mp_dense_wgrad_relu_f32) and the softmax cross-entropy over labelled rows (mp_softmax_ce_*; graphgym/loss.py:53-68),
against float64 torch."""
import pytest
import torch
import torch.nn.functional as F

from _tol import assert_close_all

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,Fi,d", [(1000, 256, 256), (777, 264, 132), (500, 1433, 128), (333, 1, 7), (70000, 128, 64),
                                    (129, 8, 260),
                                    # the loader / MFMA-wave kernel: one tile, d <= 128, several f and d tiles with ragged
                                    # edges, more rows than one workgroup's range, a range that ends inside a 16-row tile
                                    (9001, 256, 128), (3000, 512, 512), (2500, 520, 264), (1100000, 136, 72), (4097, 200, 256)])
def test_wgrad_with_relu_mask(dev, M, Fi, d):
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(M + d)
    P = torch.randn(M, Fi, generator=g)
    G = torch.randn(M, d, generator=g)
    Y = torch.relu(torch.randn(M, d, generator=g))                  # a ReLU output: exact zeros where it was clipped
    r = ops._raw_dense_wgrad_relu(P.to(dev), G.to(dev), Y.to(dev), want_bias=True)
    assert r is not None
    dW, db, gm = r
    ref_gm = torch.where(Y > 0, G, torch.zeros_like(G))
    assert torch.equal(gm.cpu(), ref_gm)                            # the masked gradient is exact
    assert_close_all(dW, P.double().t() @ ref_gm.double(), 1e-5, ref32=P.t() @ ref_gm, what="dW")
    assert_close_all(db, ref_gm.double().sum(0), 1e-5, ref32=ref_gm.sum(0), what="db")
    # aliasing the output with the input gradient is allowed (masking is idempotent)
    G2 = G.to(dev).clone()
    import ctypes as C
    from graphgym_amd._lib import lib, ptr, check
    from graphgym_amd.graph import _stream
    L = lib()
    nb = C.c_size_t(0)
    check(L.mp_dense_wgrad_ws_bytes(M, Fi, d, C.byref(nb)))
    ws = torch.empty(max(nb.value, 1), dtype=torch.uint8, device=dev)
    dW2 = torch.empty(Fi, d, device=dev)
    Pd, Yd = P.to(dev), Y.to(dev)
    check(L.mp_dense_wgrad_relu_f32(ptr(Pd), Fi, ptr(G2), d, ptr(Yd), d, ptr(G2), d, M, Fi, d, ptr(dW2), None, ptr(ws),
                                    nb.value, _stream()))
    assert torch.equal(G2.cpu(), ref_gm) and torch.equal(dW2, dW)
    # a first layer takes no input gradient: the masked gradient is not written at all, the sums are the same bits
    dW3, db3, gm3 = ops._raw_dense_wgrad_relu(P.to(dev), G.to(dev), Y.to(dev), want_bias=True, want_gm=False)
    assert gm3 is None and torch.equal(dW3, dW) and torch.equal(db3, db)
    dW4, db4, gm4 = torch.ops.mp.dense_wgrad_relu_raw(P.to(dev), G.to(dev), Y.to(dev), True, False)
    assert gm4.numel() == 0 and torch.equal(dW4, dW)


def test_relu_layers_backward_has_no_separate_mask_pass(dev):
    """the backward of a transform with a ReLU epilogue dispatches mp::dense_wgrad_relu_raw and no aten threshold op"""
    from graphgym_amd import ops
    from torch.utils._python_dispatch import TorchDispatchMode
    seen = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen.append(str(func))
            return func(*args, **(kwargs or {}))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5000, 64, generator=g).to(dev).requires_grad_(True)
    W = torch.randn(64, 64, generator=g).to(dev).requires_grad_(True)
    b = torch.randn(64, generator=g).to(dev).requires_grad_(True)
    out = ops.dense_fused(x, W, bias=b, relu=True)
    with Spy():
        out.sum().backward()
    assert any("dense_wgrad_relu_raw" in s for s in seen), seen
    assert not any("threshold_backward" in s for s in seen), seen
    xr, Wr, br = (t.detach().cpu().double().requires_grad_(True) for t in (x, W, b))
    torch.relu(xr @ Wr + br).sum().backward()
    assert_close_all(W.grad, Wr.grad, 1e-5, what="dW")
    assert_close_all(x.grad, xr.grad, 1e-5, what="dx")
    assert_close_all(b.grad, br.grad, 1e-5, what="db")


@pytest.mark.parametrize("subset", [False, True])
def test_softmax_cross_entropy_matches_torch(dev, subset):
    from graphgym_amd import nn as mpnn
    g = torch.Generator().manual_seed(3)
    N, Cn = 400_000, 7
    z = (torch.randn(N, Cn, generator=g) * 3).requires_grad_(True)
    idx = torch.randperm(N, generator=g)[: N // 3] if subset else None
    y = torch.randint(0, Cn, (N // 3 if subset else N,), generator=g)
    zd = z.detach().to(dev).requires_grad_(True)
    loss = mpnn.softmax_cross_entropy(zd, y.to(dev), None if idx is None else idx.to(dev))
    (loss * 1.7).backward()
    z64 = z.detach().double().requires_grad_(True)
    ref = F.cross_entropy(z64 if idx is None else z64[idx], y, reduction="mean")
    (ref * 1.7).backward()
    assert abs(float(loss) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    assert float((zd.grad.cpu().double() - z64.grad).abs().max()) <= 1e-6 * float(z64.grad.abs().max())
    # the harness' loss (graphgym/loss.py:53-68) goes through it
    from graphgym_amd import harness as H
    lab_idx = torch.arange(N, device=dev) if idx is None else idx.to(dev)
    l2 = H.tfg_loss(zd.detach(), lab_idx, y.to(dev), [])
    assert abs(float(l2) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))


@pytest.mark.parametrize("M,Fi,d", [(5000, 256, 10), (4099, 64, 7), (70000, 128, 16), (3000, 512, 3), (100, 4, 1)])
def test_narrow_output_weight_gradient(dev, M, Fi, d):
    """the classifier head's weight gradient x^T g (d <= 16 columns) on the engine's narrow-output kernel, with the bias
    gradient from the same pass, against float64"""
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(M + d)
    P = torch.randn(M, Fi, generator=g)
    G = torch.randn(M, d, generator=g)
    dW, db = ops._raw_dense_wgrad(P.to(dev), G.to(dev), want_bias=True)
    assert_close_all(dW, P.double().t() @ G.double(), 1e-5, ref32=P.t() @ G, what="dW")
    assert_close_all(db, G.double().sum(0), 1e-5, ref32=G.sum(0), what="db")
    dW2, db2 = ops._raw_dense_wgrad(P.to(dev), G.to(dev), want_bias=True)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)                # fixed summation order


def test_narrow_head_linear_backward(dev):
    """graphgym_amd.nn.Linear with <= 16 outputs over many rows: same gradients as torch.nn.Linear in float64"""
    from graphgym_amd import nn as mpnn
    g = torch.Generator().manual_seed(9)
    N = 1 << 17
    x = torch.randn(N, 64, generator=g)
    lin = mpnn.Linear(64, 10).to(dev)
    ref = torch.nn.Linear(64, 10).double()
    with torch.no_grad():
        ref.weight.copy_(lin.weight.detach().cpu().double()); ref.bias.copy_(lin.bias.detach().cpu().double())
    xd = x.to(dev).requires_grad_(True)
    up = torch.randn(N, 10, generator=g)
    lin(xd).backward(up.to(dev))
    xr = x.double().requires_grad_(True)
    ref(xr).backward(up.double())
    assert_close_all(lin.weight.grad, ref.weight.grad, 1e-5, ref32=(up.t() @ x), what="dW")
    assert_close_all(lin.bias.grad, ref.bias.grad, 1e-5, ref32=up.sum(0), what="db")
    assert_close_all(xd.grad, xr.grad, 1e-5, what="dx")


def test_softmax_cross_entropy_input_contract(dev):
    """labels / index are validated — a length mismatch raises on the host, an out-of-range label (torch's
    ignore_index -100 is not implemented) or index is never dereferenced and poisons the loss with NaN, and a row
    listed twice in the index accumulates both gradient terms like F.cross_entropy(logits[index])"""
    g = torch.Generator().manual_seed(5)
    N, Cn = 200_000, 8
    z = torch.randn(N, Cn, generator=g)
    y = torch.randint(0, Cn, (N,), generator=g)
    zd = z.to(dev)
    with pytest.raises((ValueError, RuntimeError)):
        torch.ops.mp.softmax_ce(zd, y.to(dev)[: N // 2], torch.arange(N // 2 + 1, device=dev))
    with pytest.raises((ValueError, RuntimeError)):
        torch.ops.mp.softmax_ce(zd[: N // 2], y.to(dev), None)
    bad = y.clone(); bad[12345] = -100
    assert torch.isnan(torch.ops.mp.softmax_ce(zd, bad.to(dev), None))
    bad = y.clone(); bad[7] = Cn
    rows = torch.ops.mp.softmax_ce_rows_raw(zd, bad.to(dev), None)
    assert torch.isnan(rows[7]) and not torch.isnan(rows[8:]).any() and not torch.isnan(rows[:7]).any()
    idx = torch.arange(N); idx[99] = N + 5
    assert torch.isnan(torch.ops.mp.softmax_ce(zd, y.to(dev), idx.to(dev)))
    dz = torch.ops.mp.softmax_ce_bwd_raw(zd, y.to(dev), idx.to(dev), torch.ones((), device=dev), 1.0 / N)
    assert torch.isfinite(dz).all() and float(dz[99].abs().max()) == 0.0        # no gradient through the bad entry
    # duplicates in the index: both terms arrive
    idx = torch.randint(0, N // 4, (N,), generator=g)                             # every row ~4 times
    zr = z.double().requires_grad_(True)
    F.cross_entropy(zr[idx], y, reduction="mean").backward()
    zq = zd.clone().requires_grad_(True)
    torch.ops.mp.softmax_ce(zq, y.to(dev), idx.to(dev)).backward()
    assert_close_all(zq.grad, zr.grad, 1e-5, what="dlogits with a non-unique index")
    # fewer labels than rows without an index: the unlabelled rows get exact zeros
    zq = zd.clone().requires_grad_(True)
    torch.ops.mp.softmax_ce(zq, y.to(dev)[: N // 2], None).backward()
    assert float(zq.grad[N // 2:].abs().max()) == 0.0


def test_bf16x3_split_follows_in_place_weight_updates(dev):
    """the bf16x3 split of W must not be cached across calls on `_version` — `w.data.uniform_()` (glorot /
    reset_parameters, idconv.py:125-128) rewrites W without bumping it"""
    import graphgym_amd as ga
    from graphgym_amd import graphgen, layers, ops
    n, d = 4000, 256
    ei = graphgen.ba_edge_index(n, 4, seed=2).to(dev)
    g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
    gen = torch.Generator().manual_seed(1)
    x = (torch.rand(n, d, generator=gen) * 2 - 1).to(dev)
    W = torch.nn.Parameter(torch.empty(d, d, device=dev))
    layers.glorot(W)
    out1, _ = ops._raw_agg_dense(g, x, W.detach(), bf16x3=True)
    v = W._version
    layers.glorot(W)                                   # .data.uniform_: same storage, same _version
    assert W._version == v
    out2, _ = ops._raw_agg_dense(g, x, W.detach(), bf16x3=True)
    ref2, _ = ops._raw_agg_dense(g, x, W.detach(), bf16x3=False)
    assert not torch.equal(out1, out2)
    from _tol import assert_close_rows
    # (the point is WHICH weights were used — a stale split is off by O(1) — so the bar is north_star's 1e-5 against the
    # exact-fp32 product of the same launch, not the few-ulp agreement test_bf16x3_product_is_fp32_accurate measures)
    assert_close_rows(out2, ref2.double(), 1e-5, what="bf16x3 after an in-place re-initialisation")


def test_ginidconv_forward_reads_no_device_scalar_and_captures(dev):
    """`float(self.eps)` on the device buffer was a host sync per forward and broke HIP-graph capture;
    eps is now carried as a host float that follows the constructor and load_state_dict"""
    from graphgym_amd import graphgen, layers
    from graphgym_amd.harness import Batch
    n, d = 3000, 64
    ei = graphgen.ba_edge_index(n, 3, seed=4).to(dev)
    ids = torch.arange(0, n, 10, device=dev)
    layer = layers.GINIDConv(d, d).to(dev)
    sd = layer.state_dict()
    assert "model.eps" in sd                                             # the reference's buffer (idconv.py:361)
    sd["model.eps"] = torch.tensor([0.25])
    layer.load_state_dict(sd)
    assert layer.model._eps_value() == 0.25 and float(layer.model.eps) == 0.25
    x = torch.randn(n, d, device=dev)
    b = Batch(node_feature=x.clone(), edge_index=ei, node_id_index=ids)
    with torch.no_grad():
        eager = layer(b).node_feature.clone()                            # warm the CSR cache
        b.node_feature = x.clone()
        static_in = b.node_feature
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            b.node_feature = static_in
            layer(b)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        b.node_feature = static_in
        with torch.cuda.graph(graph):
            out = layer(b).node_feature
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    # oracle: GINIDConvLayer.forward restated (idconv.py:367-376), float64, eps = 0.25
    from oracle import ref_layers as RL
    from _tol import assert_close_rows
    m = layer.model
    p64 = lambda t: t.detach().cpu().double()

    def mlp(seq):
        w0, b0, w1, b1 = p64(seq[0].weight), p64(seq[0].bias), p64(seq[2].weight), p64(seq[2].bias)
        return lambda t: torch.relu(t @ w0.t() + b0) @ w1.t() + b1
    ref = RL.ginid_conv(x.cpu().double(), ei.cpu(), ids.cpu(), mlp(m.nn), mlp(m.nn_id), eps=0.25)
    assert_close_rows(out, ref, 1e-5, what="ginidconv with eps = 0.25")

"""The one-kernel aggregate -> transform (mp_agg_dense_f32) against the CPU oracle: SparseAdj.matmul
(sparse_adj.py:91-97) followed by the layer's kernel product, bias and activation, forward and backward.
Tolerances: tests/_tol.py — every output row within 1e-5 of its own magnitude against the oracle in float64, or twice
the float32 oracle's own distance from it (policed); weight / bias gradients (reductions over all rows) on one scale.
Bitwise reproducible run to run."""
import pytest
import torch

from _tol import both, close, close_all
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


def relu_like_engine(pre, out_engine):
    """A ReLU input within fp32 rounding of zero has an arbitrary subgradient (and with ~10^5 activations per case a few
    ARE that close): differentiate the float64 reference through the engine's own pattern (output > 0), after checking
    that the pattern differs from the reference's sign test only where the input is ~0."""
    mask = (out_engine.detach().cpu() > 0)
    own = pre.detach() > 0
    off = own != mask
    if bool(off.any()):
        worst = float(pre.detach().abs()[off].max())
        assert worst <= 1e-5 * float(pre.detach().abs().max()), f"activation pattern differs at |input| = {worst:.3e}"
    return pre * mask.to(pre.dtype)


def make_graph(n, E, seed, hubs=False, weighted=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, E), generator=g)
    if hubs:   # two very long rows that share a 32-row tile, and empty rows
        a = torch.stack([torch.full((7000,), 5), torch.randint(0, n, (7000,), generator=g)])
        b = torch.stack([torch.full((2500,), 9), torch.randint(0, n, (2500,), generator=g)])
        ei = torch.cat([ei, a, b], dim=1)
        ei = ei[:, ei[0] % 13 != 4]
    w = torch.rand(ei.size(1), generator=g) + 0.05 if weighted else None
    return ei, w


@pytest.mark.parametrize("n,E,F,d,weighted,hubs,self_scale", [
    (1000, 12000, 256, 256, True, False, 0.0),
    (37, 150, 64, 10, True, False, 0.0),
    (2049, 30000, 128, 64, False, True, 1.25),
    (3333, 40000, 256, 130, True, True, 0.0),
    (64, 64, 256, 512, False, False, 2.0),
    (33, 0 + 1, 64, 2, True, False, 0.0),
    (700, 9000, 512, 512, True, False, 1.0),      # config C5's width: two K halves over the row tile
    (1500, 20000, 512, 130, False, True, 0.0),
    (90, 500, 512, 256, True, False, 0.0),
    (1300, 16000, 256, 512, True, True, 0.5),      # wide output: eight consumer waves
    (777, 8000, 256, 320, False, False, 0.0),      # ... and a ragged last column block
    (900, 9000, 512, 384, True, True, 0.0),        # F = 512 with a ragged second block: the one-role kernel
])
def test_agg_dense_matches_oracle(dev, n, E, F, d, weighted, hubs, self_scale):
    import graphgym_amd as ga
    from graphgym_amd import ops
    ei, w = make_graph(n, E, seed=n + F, hubs=hubs, weighted=weighted)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(n, F, generator=gen)
    W = torch.randn(F, d, generator=gen) / F ** 0.5
    b = torch.randn(d, generator=gen)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, None if w is None else w.to(dev), dst_row=0)
    assert ops.agg_dense_supported(G, x.to(dev), W.to(dev))
    for relu in (False, True):
        up = torch.randn(n, d, generator=gen)
        xd = x.to(dev).requires_grad_(True)
        Wd = W.to(dev).requires_grad_(True)
        bd = b.to(dev).requires_grad_(True)
        out = ops.agg_dense(G, xd, Wd, bias=bd, relu=relu, self_scale=self_scale)
        out.backward(up.to(dev))

        def ref(c):     # SparseAdj.matmul (sparse_adj.py:91-97), then the kernel product, bias, activation
            xr, Wr, br = (c(t).clone().requires_grad_(True) for t in (x, W, b))
            pre = (R.SparseAdj(ei, None if w is None else c(w), [n, n]) @ xr + self_scale * xr) @ Wr + br
            o = relu_like_engine(pre, out) if relu else pre
            o.backward(c(up))
            return o.detach(), xr.grad, Wr.grad, br.grad
        r64, r32 = both(ref)
        close(out, (r64[0], r32[0]), what="agg_dense forward")
        close(xd.grad, (r64[1], r32[1]), what="agg_dense dx")
        close_all(Wd.grad, (r64[2], r32[2]), what="agg_dense dW")
        close_all(bd.grad, (r64[3], r32[3]), what="agg_dense db")
        out2 = ops.agg_dense(G, xd, Wd, bias=bd, relu=relu, self_scale=self_scale)
        assert torch.equal(out, out2)                             # no atomics: bitwise reproducible


def test_agg_dense_falls_back_outside_its_shapes(dev):
    import graphgym_amd as ga
    from graphgym_amd import ops
    n = 300
    ei, w = make_graph(n, 3000, seed=3)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, w.to(dev), dst_row=0)
    gen = torch.Generator().manual_seed(2)
    for F, d in ((48, 32), (256, 7)):
        x, W = torch.randn(n, F, generator=gen), torch.randn(F, d, generator=gen) / F ** 0.5
        assert not ops.agg_dense_supported(G, x.to(dev), W.to(dev))
        ref = both(lambda c: (R.SparseAdj(ei, c(w), [n, n]) @ c(x)) @ c(W))
        close(ops.agg_dense(G, x.to(dev), W.to(dev)), ref, what=f"two-kernel fallback {F} -> {d}")
    # the C entry point says so itself
    import ctypes as C
    from graphgym_amd._lib import lib, ptr
    x, W, out = torch.randn(n, 48, device=dev), torch.randn(48, 32, device=dev), torch.empty(n, 32, device=dev)
    st = lib().mp_agg_dense_f32(ptr(G.rowptr), ptr(G.col), ptr(G.val), n, 0, ptr(x), 48, 48, None, 0, 0.0, ptr(W), 32,
                                32, None, 0, None, None, 0, ptr(out), 32, None, None)
    assert st == 2                                                # MP_ERR_UNSUPPORTED


def test_layers_use_the_fused_path_and_agree(dev):
    """GCN / GIN layers at equal widths take the one-kernel path; same numbers as the explicit two-kernel order"""
    import graphgym_amd as ga
    from graphgym_amd import layers as L, ops
    n, F = 500, 64
    ei, _ = make_graph(n, 5000, seed=5, weighted=False)
    ei = ei.to(dev)
    x = torch.randn(n, F, generator=torch.Generator().manual_seed(4)).to(dev)
    calls = []
    orig = ops._raw_agg_dense
    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    ops._raw_agg_dense = spy
    try:
        torch.manual_seed(0)
        fused = L.GCN(F, activation=torch.relu, in_features=F)
        two = L.GCN(F, activation=torch.relu, in_features=F, order="transform_first")
        two.load_state_dict(fused.state_dict())
        fused, two = fused.to(dev), two.to(dev)
        a = fused([x, ei]); nfused = len(calls)
        b = two([x, ei])
        assert nfused == 1 and len(calls) == 1
        close(a, b.detach().double(), what="one-kernel GCN layer vs transform-first")     # engine vs engine, per row
        conv = L.GINConvLayer(L._mlp2(F, F)).to(dev)
        h = conv(x, ei)
        assert len(calls) == 2
        g = ga.CSRGraph.from_edge_index(ei, n)
        ref = conv.nn(ops.spmm(g, x, "sum", self_scale=1.0))
        close(h, ref.detach().double(), what="one-kernel GIN head vs two kernels")
        pyg = L.GCNConvLayer(F, F).to(dev)
        pyg(x, ei)
        assert len(calls) == 3
    finally:
        ops._raw_agg_dense = orig


@pytest.mark.parametrize("n,E,F,units,weighted,hubs", [(900, 9000, 64, 64, False, False), (2100, 30000, 256, 256, True, True),
                                                       (70, 300, 128, 12, False, False)])
def test_sage_concat_fused_matches_oracle(dev, n, E, F, units, weighted, hubs):
    """[x Ws ‖ mean_j(x_j) Wn] + b -> relu (MeanGraphSage / IDSAGE.call, TfgIDLayer.py:100-117), forward and backward"""
    import graphgym_amd as ga
    from graphgym_amd import ops
    ei, w = make_graph(n, E, seed=7 + n, hubs=hubs, weighted=weighted)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, F, generator=gen)
    Ws = torch.randn(F, units // 2, generator=gen) / F ** 0.5
    Wn = torch.randn(F, units // 2, generator=gen) / F ** 0.5
    b = torch.randn(units, generator=gen)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, None if w is None else w.to(dev), dst_row=0)
    assert ops.agg_dense_supported(G, x.to(dev), Wn.to(dev))
    for relu in (True, False):
        up = torch.randn(n, units, generator=gen)
        xd, Wsd, Wnd, bd = [t.to(dev).requires_grad_(True) for t in (x, Ws, Wn, b)]
        out = ops.sage_concat(G, xd, Wsd, Wnd, bd, relu=relu)
        out.backward(up.to(dev))

        def ref(c):
            xr, Wsr, Wnr, br = [c(t).clone().requires_grad_(True) for t in (x, Ws, Wn, b)]
            mean = R.coo_aggregate(ei[0], ei[1], None if w is None else c(w), xr, n, "mean")     # mean_reducer, :98
            pre = torch.cat([xr @ Wsr, mean @ Wnr], dim=1) + br
            o = relu_like_engine(pre, out) if relu else pre
            o.backward(c(up))
            return o.detach(), xr.grad, Wsr.grad, Wnr.grad, br.grad
        r64, r32 = both(ref)
        close(out, (r64[0], r32[0]), what="sage_concat forward")
        close(xd.grad, (r64[1], r32[1]), what="sage_concat dx")
        close_all(Wsd.grad, (r64[2], r32[2]), what="sage_concat dWs")
        close_all(Wnd.grad, (r64[3], r32[3]), what="sage_concat dWn")
        close_all(bd.grad, (r64[4], r32[4]), what="sage_concat db")
    # and the layer takes this path
    from graphgym_amd import layers as L
    calls = []
    orig = ops._raw_agg_dense
    ops._raw_agg_dense = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        layer = L.MeanGraphSage(units, in_features=F).to(dev)
        h = layer([x.to(dev), ei.to(dev)] + ([w.to(dev)] if w is not None else []))
        assert len(calls) == 1 and h.shape == (n, units)
    finally:
        ops._raw_agg_dense = orig


def test_inference_does_not_keep_the_aggregated_rows(dev):
    """under no_grad the kernel is asked for `out` only (no [N, F] write); with grad enabled P is kept for dW"""
    import graphgym_amd as ga
    from graphgym_amd import ops
    n, F = 400, 64
    ei, _ = make_graph(n, 4000, seed=9, weighted=False)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, dst_row=0)
    x = torch.randn(n, F).to(dev)
    W = torch.randn(F, F).to(dev).requires_grad_(True)
    seen = []
    orig = ops._raw_agg_dense
    def spy(*a, **k):
        seen.append(bool(k.get("want_P")))
        return orig(*a, **k)
    ops._raw_agg_dense = spy
    try:
        with torch.no_grad():
            a = ops.agg_dense(G, x, W)
        b = ops.agg_dense(G, x, W)
        b.sum().backward()
    finally:
        ops._raw_agg_dense = orig
    assert seen == [False, True] and torch.equal(a, b.detach()) and W.grad is not None


def test_full_size_properties_c2(dev):
    """C2 size (10^6 nodes, 1.1*10^7 stored entries, F = d_out = 256), where the oracle would take minutes:
    size-independent properties — agreement with the two-kernel order (aggregation kernel, then transform
    kernel, each checked against the oracle elsewhere), linearity in X, and the transposed-operator identity
    <y, (A x) W> = <(A' (y W')), x> that the backward pass relies on"""
    import graphgym_amd as ga
    from graphgym_amd import graphgen, ops
    n, F = 1_000_000, 256
    ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
    g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm()
    gen = torch.Generator(device=dev).manual_seed(3)
    x1 = torch.rand((n, F), device=dev, generator=gen) * 2 - 1
    x2 = torch.rand((n, F), device=dev, generator=gen) * 2 - 1
    W = (torch.rand((F, F), device=dev, generator=gen) - 0.5) * (2.0 / F ** 0.5)
    b = torch.rand((F,), device=dev, generator=gen) - 0.5
    one, P = ops._raw_agg_dense(g, x1, W, b, True, want_P=True)
    agg, _ = ops._raw_spmm(g, x1, 0)
    two = ops._raw_dense_fused(agg, W, None, None, b, True)
    # engine vs engine at 10^6 rows, per row (row maxima on the device: no [N, F] float64 copies on the host)
    def rows_close(a, ref, what):
        err = (a - ref).abs().amax(1).double()
        assert bool((err <= 1e-5 * ref.abs().amax(1).double()).all()), what
    rows_close(one, two, "one kernel vs two")
    rows_close(P, agg, "kept aggregated rows vs the aggregation kernel")
    # linearity (no bias, no activation)
    y1, _ = ops._raw_agg_dense(g, x1, W)
    y2, _ = ops._raw_agg_dense(g, x2, W)
    y12, _ = ops._raw_agg_dense(g, x1 + x2, W)
    rows_close(y12, y1 + y2, "linearity")
    # adjoint identity with the transposed operator (what dX = (A' g) W' computes)
    up = torch.rand((n, F), device=dev, generator=gen) - 0.5
    gt = g.transpose()
    back, _ = ops._raw_agg_dense(gt, up, W.t().contiguous())
    lhs = float((up.double() * y1.double()).sum())
    rhs = float((back.double() * x1.double()).sum())
    mag = float((up.double() * y1.double()).abs().sum())          # the sums cancel: compare against sum |terms|
    assert abs(lhs - rhs) <= 1e-6 * mag, (lhs, rhs, mag)


def test_star_hub_goes_to_the_plan_based_kernel(dev):
    """a row longer than FUSED_MAX_ROW (a star's centre) is not given to one workgroup: agg_dense falls back to the
    aggregation kernel (hub rows cut into pieces over many waves) + transform kernel, same numbers"""
    import graphgym_amd as ga
    from graphgym_amd import ops
    leaves, F = ops.FUSED_MAX_ROW + 5000, 64
    n = leaves + 1
    ei = torch.stack([torch.zeros(leaves, dtype=torch.int64), torch.arange(1, n)])      # row 0 <- every leaf
    ei = torch.cat([ei, ei.flip(0)], dim=1)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, dst_row=0)
    assert G.max_row_entries() == leaves
    gen = torch.Generator().manual_seed(0)
    x, W = torch.randn(n, F, generator=gen), torch.randn(F, F, generator=gen) / 8
    assert not ops.agg_dense_supported(G, x.to(dev), W.to(dev))
    out = ops.agg_dense(G, x.to(dev), W.to(dev))
    centre = both(lambda c: c(x)[1:].sum(0) @ c(W))      # 267 144 terms: the float32 oracle's own sum is ~1e-5 off
    leaf = both(lambda c: c(x)[0] @ c(W))
    close(out[0], centre, what="star centre")
    close(out[1], leaf, what="first leaf")
    close(out[n - 1], leaf, what="last leaf")


@pytest.mark.parametrize("n,E,F,d,weighted,hubs,n_id,self_scale", [
    (1200, 15000, 256, 256, True, False, 40, 0.0),
    (2049, 30000, 128, 64, True, True, 300, 0.0),
    (500, 4000, 64, 130, False, False, 1, 0.0),
    (800, 9000, 512, 512, True, False, 64, 0.0),
    (300, 2500, 256, 256, True, False, 300, 0.5),       # every node an identity node
])
def test_agg_dense_id_matches_oracle(dev, n, E, F, d, weighted, hubs, n_id, self_scale):
    """out = act(A (x W + S x W_id) + b) (gcn_id, TfgIDLayer.py:510-523; idconv.py:150-177) through the one-kernel layer
    plus the identity fix-up, forward and every gradient, against a float64 evaluation of the reference's own order
    (transform, scatter-add the identity rows, aggregate)"""
    import graphgym_amd as ga
    from graphgym_amd import ops
    ei, w = make_graph(n, E, seed=n + F + 1, hubs=hubs, weighted=weighted)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(n, F, generator=gen)
    W = torch.randn(F, d, generator=gen) / F ** 0.5
    Wid = torch.randn(F, d, generator=gen) / F ** 0.5
    b = torch.randn(d, generator=gen)
    ids = torch.randperm(n, generator=gen)[:n_id]
    if hubs and 5 not in ids.tolist():
        ids[0] = 5                                                  # a hub row's own node among the identity nodes
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, None if w is None else w.to(dev), dst_row=0)
    for relu in (False, True):
        up = torch.randn(n, d, generator=gen)
        xd, Wd, Wid_d, bd = (t.to(dev).requires_grad_(True) for t in (x, W, Wid, b))
        out = ops.agg_dense_id(G, xd, Wd, Wid_d, ids.to(dev), bias=bd, relu=relu, self_scale=self_scale)
        assert out is not None
        out.backward(up.to(dev))

        def ref(c):
            xr, Wr, Wir, br = (c(t).clone().requires_grad_(True) for t in (x, W, Wid, b))
            h = xr @ Wr
            h = h.index_add(0, ids, xr[ids] @ Wir)                     # TfgIDLayer.py:513-515
            pre = R.SparseAdj(ei, None if w is None else c(w), [n, n]) @ h + self_scale * (xr @ Wr) + br
            o = relu_like_engine(pre, out) if relu else pre
            o.backward(c(up))
            return o.detach(), xr.grad, Wr.grad, Wir.grad, br.grad
        r64, r32 = both(ref)
        close(out, (r64[0], r32[0]), what="agg_dense_id forward")
        close(xd.grad, (r64[1], r32[1]), what="agg_dense_id dx")
        close_all(Wd.grad, (r64[2], r32[2]), what="agg_dense_id dW")
        close_all(Wid_d.grad, (r64[3], r32[3]), what="agg_dense_id dW_id")
        close_all(bd.grad, (r64[4], r32[4]), what="agg_dense_id db")
        out2 = ops.agg_dense_id(G, xd, Wd, Wid_d, ids.to(dev), bias=bd, relu=relu, self_scale=self_scale)
        assert torch.equal(out, out2)
    # the two-kernel formulation (two-branch aggregation + dual GEMM) gives the same numbers
    P, Q = ops.idgnn_aggregate(G, ids.to(dev), x.to(dev))
    two = ops.dense_fused(P, W.to(dev), Q, Wid.to(dev), b.to(dev)) + self_scale * (x.to(dev) @ W.to(dev))
    one = ops.agg_dense_id(G, x.to(dev), W.to(dev), Wid.to(dev), ids.to(dev), bias=b.to(dev), self_scale=self_scale)
    close(one, two.double(), what="one-kernel ID layer vs two-branch aggregation + dual transform")
    with pytest.raises(ValueError, match="duplicate"):
        ops.agg_dense_id(G, x.to(dev), W.to(dev), Wid.to(dev), torch.tensor([1, 1], device=dev))


def test_id_layers_take_one_launch(dev):
    """Tfg-idgcn / gcnidconv at equal widths and the ID-GIN head run the one-kernel layer (one mp_agg_dense_f32 launch
    per forward) and agree with the reference's transform-first order"""
    from graphgym_amd import layers as L, ops
    n, F = 600, 128
    ei, _ = make_graph(n, 6000, seed=8, weighted=False)
    ei = torch.cat([ei, ei.flip(0)], dim=1).to(dev)
    x = torch.randn(n, F, generator=torch.Generator().manual_seed(4)).to(dev)
    ids = torch.arange(0, n, 7, device=dev)
    calls = []
    orig = ops._raw_agg_dense
    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    ops._raw_agg_dense = spy
    try:
        torch.manual_seed(0)
        for make in (lambda order: L.IDGCN(F, activation=torch.relu, in_features=F, order=order),
                     lambda order: L.GCNIDConvLayer(F, F, bias=True, order=order)):
            fused, two = make("auto"), make("transform_first")
            two.load_state_dict(fused.state_dict())
            fused, two = fused.to(dev), two.to(dev)
            calls.clear()
            with torch.no_grad():
                a = fused([x, ei, ids]) if isinstance(fused, L.IDGCN) else fused(x, ei, ids)
                assert len(calls) == 1
                b = two([x, ei, ids]) if isinstance(two, L.IDGCN) else two(x, ei, ids)
                assert len(calls) == 1
            close(a, b.double(), what="ID layer: one launch vs transform-first")
        gin = L.GINIDConvLayer(L._mlp2(F, F), L._mlp2(F, F)).to(dev)
        calls.clear()
        with torch.no_grad():
            a = gin(x, ei, ids)
            assert len(calls) == 1
            g = L.get_graph(None, ei, n, loops="remove")
            h = ops.spmm(g, x, "sum", self_scale=1.0)
            b = ops.index_add_rows(gin.nn(h), ids, gin.nn_id(h[ids]))
        close(a, b.double(), what="ID-GIN head: one launch vs two")
    finally:
        ops._raw_agg_dense = orig


@pytest.mark.parametrize("n,E,F,d,weighted,hubs", [(3000, 40000, 256, 256, True, True), (1000, 9000, 128, 64, False, False),
                                                    (800, 9000, 512, 512, True, False), (500, 3000, 64, 130, True, False)])
def test_bf16x3_product_is_fp32_accurate(dev, n, E, F, d, weighted, hubs):
    """the one-kernel layer on the bf16 matrix pipe with three-way split operands (six cross terms) against float64, held
    to the same 1e-5 as the exact-fp32 MFMA form, and within a few ulp of that form; weights with a wide dynamic range"""
    import graphgym_amd as ga
    from graphgym_amd import ops
    ei, w = make_graph(n, E, seed=n + F + 7, hubs=hubs, weighted=weighted)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, F, generator=gen) * torch.exp(torch.randn(n, 1, generator=gen))       # rows of different scales
    W = torch.randn(F, d, generator=gen) / F ** 0.5 * torch.exp(torch.randn(1, d, generator=gen))
    b = torch.randn(d, generator=gen)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, None if w is None else w.to(dev), dst_row=0)
    exact, _ = ops._raw_agg_dense(G, x.to(dev), W.to(dev), b.to(dev), False, bf16x3=False)
    split, _ = ops._raw_agg_dense(G, x.to(dev), W.to(dev), b.to(dev), False, bf16x3=True)
    agg = torch.zeros(n, F, dtype=torch.float64).index_add_(
        0, ei[0], x.double()[ei[1]] * (w.double().unsqueeze(1) if w is not None else 1.0))
    ref = agg @ W.double() + b.double()
    from _tol import assert_close_rows
    assert_close_rows(split, ref, 1e-5, ref32=exact, what="bf16x3 vs float64")
    e_split = float((split.cpu().double() - ref).abs().max())
    e_exact = float((exact.cpu().double() - ref).abs().max())
    assert e_split <= 4 * e_exact + 1e-7 * float(ref.abs().max()), (e_split, e_exact)
    split2, _ = ops._raw_agg_dense(G, x.to(dev), W.to(dev), b.to(dev), False, bf16x3=True)
    assert torch.equal(split, split2)


def test_agg_dense_with_a_residual(dev):
    """mp_agg_dense_add_f32: out = act((A X) W + b + R) equals the plain launch followed by an add, bit for bit without
    an activation, also when R is the output buffer itself; and the MeanGraphSage backward uses it (no add pass, no
    threshold_backward pass) while its gradients stay with float64 (test_sage_concat_fused_matches_oracle)"""
    import graphgym_amd as ga
    from graphgym_amd import ops
    from torch.utils._python_dispatch import TorchDispatchMode
    n, F, d = 4000, 128, 96
    ei, w = make_graph(n, 50000, seed=21, hubs=True, weighted=True)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, F, generator=gen).to(dev)
    W = (torch.randn(F, d, generator=gen) / F ** 0.5).to(dev)
    b = torch.randn(d, generator=gen).to(dev)
    R = torch.randn(n, d, generator=gen).to(dev)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, w.to(dev), dst_row=0)
    plain, _ = ops._raw_agg_dense(G, x, W, b, False)
    added, _ = ops._raw_agg_dense(G, x, W, b, False, residual=R)
    assert torch.equal(added, plain + R)
    buf = R.clone()
    ops._raw_agg_dense(G, x, W, b, False, out=buf, residual=buf)          # in place
    assert torch.equal(buf, added)
    act, _ = ops._raw_agg_dense(G, x, W, b, True, residual=R)            # the activation comes after the residual
    assert torch.equal(act, torch.relu(added))

    seen = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen.append(str(func))
            return func(*args, **(kwargs or {}))
    xs = torch.randn(n, 128, generator=gen).to(dev).requires_grad_(True)
    Ws = (torch.randn(128, 64, generator=gen) / 11).to(dev).requires_grad_(True)
    Wn = (torch.randn(128, 64, generator=gen) / 11).to(dev).requires_grad_(True)
    bs = torch.randn(128, generator=gen).to(dev).requires_grad_(True)
    out = ops.sage_concat(G, xs, Ws, Wn, bs, relu=True)
    with Spy():
        out.sum().backward()
    assert not any("threshold_backward" in s for s in seen), seen
    assert not any("aten.add_" in s for s in seen), seen

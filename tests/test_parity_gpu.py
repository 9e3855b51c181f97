"""Parity of the HIP path (through the C ABI: graphgym_amd.graph / ops -> libmpengine.so)
against the CPU oracle and the committed golden vectors.

Bars: integer / index work bit-exact; fp32 results within 1e-5 of EVERY OUTPUT ROW's own magnitude against a
float64 evaluation of the oracle on the same fp32 inputs — tests/_tol.py, the one tolerance regime of the suite
(`close(engine, both(lambda c: oracle(c(inputs))))`: the closure is evaluated in float64 and in the reference's
float32; rows outside 1e-5 may use twice the float32 oracle's own distance from float64, and _tol polices how many do).
The oracle is unpinned by the reference's numerics.
"""
import numpy as np
import pytest
import torch

from _tol import both, close, close_all, mag_of
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


def csr_reference(dst, src, N, w=None):
    """numpy restatement of the CSR order contract: stable sort by (dst, src)"""
    dst, src = np.asarray(dst), np.asarray(src)
    order = np.lexsort((np.arange(dst.size), src, dst))
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, src[order], order, (None if w is None else np.asarray(w)[order])


# --------------------------------------------------------------------------- graph build
def test_csr_build_is_bit_exact(dev):
    import graphgym_amd as ga
    g = torch.Generator().manual_seed(11)
    for N, E in [(1, 0), (5, 0), (1, 3), (17, 200), (1000, 20000), (300, 50000)]:
        ei = torch.randint(0, N, (2, E), generator=g)
        w = torch.rand(E, generator=g)
        G = ga.CSRGraph.from_edge_index(ei.to(dev), N, w.to(dev), validate=True)
        rowptr, col, order, wv = csr_reference(ei[1].numpy(), ei[0].numpy(), N, w.numpy())
        assert G.nnz == E
        assert (G.rowptr.cpu().numpy() == rowptr).all()
        assert (G.col.cpu().numpy() == col).all()
        assert (G.eid.cpu().numpy() == order).all()
        assert (G.val.cpu().numpy() == wv).all()
        # TF convention: edge_index[0] is the destination row
        G2 = ga.CSRGraph.from_edge_index(ei.to(dev), N, dst_row=0)
        rowptr2, col2, _, _ = csr_reference(ei[0].numpy(), ei[1].numpy(), N)
        assert (G2.rowptr.cpu().numpy() == rowptr2).all() and (G2.col.cpu().numpy() == col2).all()
        assert G2.val is None


@pytest.mark.parametrize("n", [1_000_000, 10_000_000])
def test_csr_and_transpose_index_exact_at_baseline_sizes(dev, n):
    """north_star: "bit-exact edge indexing" — at the C2 / C4 sizes, against an INDEPENDENT device computation
    (torch.sort / bincount / cumsum), not against the engine's own CSR: a mis-sorted or dropped entry anywhere in the
    1.2 * 10^8-key rocPRIM sort -> rowptr -> emit -> transpose chain fails a torch.equal here.

    Semantics pinned: row = destination (sparse_adj.py:91-97: unsorted_segment_sum over edge_index[0] in the TF
    convention; PyG: aggregation at edge_index[1]), entries of a row ordered by (source, input position), loops
    appended behind the input edges (sparse_adj.py:58-63 add_self_loop -> concat), and for the PyG policy
    add_remaining_self_loops (idconv.py:140-141): an existing loop is dropped and re-added once with its weight."""
    import graphgym_amd as ga
    from graphgym_amd import graphgen
    gen = torch.Generator(device=dev).manual_seed(n % 1000 + 3)
    ei = graphgen.ba_edge_index(n, 5, seed=12345, device=dev)           # comes out sorted by (dst, src): shuffle it,
    ei = ei[:, torch.randperm(ei.size(1), device=dev, generator=gen)]   # so that the sort has real work to do
    # duplicates (a multigraph keeps them, in input order) and explicit self loops on distinct nodes
    dup = ei[:, torch.randint(0, ei.size(1), (50_000,), device=dev, generator=gen)]
    loops = torch.randperm(n, device=dev, generator=gen)[:30_000]
    ei = torch.cat([ei, dup, torch.stack([loops, loops])], dim=1)
    ei = ei[:, torch.randperm(ei.size(1), device=dev, generator=gen)]
    E = ei.size(1)
    w = torch.rand(E, device=dev, generator=gen) + 0.5
    src, dst = ei[0], ei[1]
    ar = torch.arange(n, device=dev)

    def expect(s_in, d_in, w_in, eid_in):
        key = d_in * n + s_in
        skey, order = torch.sort(key, stable=True)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(d_in, minlength=n), 0)
        return skey, order, rowptr, w_in[order], eid_in[order]

    def check(G, s_in, d_in, w_in, eid_in, what):
        skey, order, rowptr, wv, eid = expect(s_in, d_in, w_in, eid_in)
        assert G.nnz == skey.numel(), what
        assert torch.equal(G.rowptr.long(), rowptr), what + ": rowptr"
        assert torch.equal(G.row_ids().long() * n + G.col.long(), skey), what + ": (row, col) keys"
        assert torch.equal(G.eid.long(), eid), what + ": input positions"
        assert torch.equal(G.val, wv), what + ": values"
        del skey, order, rowptr, wv, eid
        # transposed operator (the backward's A^T): rows = sources, entries ordered by (destination, CSR position)
        T = G.transpose()
        r, c = G.row_ids().long(), G.col.long()
        tkey, torder = torch.sort(c * n + r, stable=True)
        trp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        trp[1:] = torch.cumsum(torch.bincount(c, minlength=n), 0)
        assert torch.equal(T.rowptr.long(), trp), what + ": transposed rowptr"
        assert torch.equal(T.row_ids().long() * n + T.col.long(), tkey), what + ": transposed keys"
        assert torch.equal(T.pos.long(), torder), what + ": transposed positions"
        assert torch.equal(T.val, G.val[torder]), what + ": transposed values"

    # TF flavour: SparseAdj(edge_index, w).add_self_loop() — N loops of weight `fill` appended, nothing removed
    G = ga.CSRGraph.from_edge_index(ei, n, w, add_self_loops=True, fill=2.0)
    check(G, torch.cat([src, ar]), torch.cat([dst, ar]), torch.cat([w, torch.full((n,), 2.0, device=dev)]),
          torch.cat([torch.arange(E, device=dev), -1 - ar]), "add_self_loop")
    del G
    torch.cuda.empty_cache()
    # PyG flavour: add_remaining_self_loops — input loops dropped, one loop per node re-added carrying the old weight
    G = ga.CSRGraph.from_edge_index(ei, n, w, remove_self_loops=True, add_self_loops=True, keep_loop_weight=True)
    keep = src != dst
    lw = torch.ones(n, device=dev)
    lw[src[~keep]] = w[~keep]
    check(G, torch.cat([src[keep], ar]), torch.cat([dst[keep], ar]), torch.cat([w[keep], lw]),
          torch.cat([torch.arange(E, device=dev)[keep], -1 - ar]), "add_remaining_self_loops")
    # and the by-destination / by-source degrees the normalisation uses, against float64 index_add
    deg64 = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, G.row_ids().long(), G.val.double())
    assert float(((G.degree("row").double() - deg64).abs() / deg64).max()) <= 1e-6
    deg64 = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, G.col.long(), G.val.double())
    assert float(((G.degree("col").double() - deg64).abs() / deg64).max()) <= 1e-6


def test_edge_index_validation(dev):
    import graphgym_amd as ga
    ei = torch.tensor([[0, 5], [1, 0]], device=dev)
    with pytest.raises(ValueError):
        ga.CSRGraph.from_edge_index(ei, 3, validate=True)


def _dense(G):
    A = torch.zeros(G.num_nodes, G.num_nodes, dtype=torch.float64)
    rows = G.row_ids().cpu().long()
    v = torch.ones(G.nnz, dtype=torch.float64) if G.val is None else G.val.cpu().double()
    A.index_put_((rows, G.col.cpu().long()), v, accumulate=True)
    return A


def test_self_loop_policies_match_oracle(dev):
    import graphgym_amd as ga
    ei = torch.tensor([[0, 1, 1, 2, 3, 3, 0], [1, 1, 2, 2, 0, 3, 1]])
    w = torch.tensor([1., 5., 2., 7., 3., 4., 0.5])
    N = 5

    def dense_of(e, ww):
        A = torch.zeros(N, N, dtype=torch.float64)
        A.index_put_((e[1], e[0]), ww.double(), accumulate=True)
        return A
    # PyG add_remaining_self_loops (existing loop weight kept; idconv.py:140-141)
    e, ww = R.add_remaining_self_loops(ei, w, 2.0, N)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N, w.to(dev), remove_self_loops=True, add_self_loops=True,
                                    keep_loop_weight=True, fill=2.0)
    assert G.nnz == e.size(1)
    # two loops on node 1/3 would be ambiguous; here each node has at most one, so exact
    assert torch.equal(_dense(G), dense_of(e, ww))
    # remove_self_loops (idconv.py:370)
    e, ww = R.remove_self_loops(ei, w)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N, w.to(dev), remove_self_loops=True)
    assert G.nnz == e.size(1) and torch.equal(_dense(G), dense_of(e, ww))
    # remove + add_self_loops (idconv.py:302-304)
    e, _ = R.remove_self_loops(ei)
    e, _ = R.add_self_loops(e, None, 1.0, N)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N, None, remove_self_loops=True, add_self_loops=True)
    assert G.nnz == e.size(1) and torch.equal(_dense(G), dense_of(e, torch.ones(e.size(1))))
    # TF add_self_loop: unconditional append (sparse_adj.py:58-63)
    sa = R.SparseAdj(torch.stack([ei[1], ei[0]]), w, [N, N]).add_self_loop(fill_weight=2.0)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N, w.to(dev), add_self_loops=True, fill=2.0)
    A = torch.zeros(N, N, dtype=torch.float64)
    A.index_put_((sa.row, sa.col), sa.edge_weight.double(), accumulate=True)
    assert G.nnz == ei.size(1) + N and torch.equal(_dense(G), A)
    assert (G.eid.cpu() < 0).sum().item() == N


def test_gcn_norm_matches_golden_both_flavours(dev, golden):
    import graphgym_amd as ga
    z = golden("aggregation.npz")
    for name in sorted({k.split("/")[0] for k in z.files}):
        n = int(z[f"{name}/n"])
        ei = torch.from_numpy(z[f"{name}/edge_index"]).to(dev)
        w = torch.from_numpy(z[f"{name}/w"]).to(dev)
        # TF: add loops, degree by row
        G = ga.CSRGraph.from_edge_index(ei, n, w, add_self_loops=True).gcn_norm("row")
        ti, tw = z[f"{name}/tf_norm_index"], z[f"{name}/tf_norm_weight"]
        A = torch.zeros(n, n, dtype=torch.float64)
        A.index_put_((torch.from_numpy(ti[0]), torch.from_numpy(ti[1])), torch.from_numpy(tw).double(), accumulate=True)
        # the golden weights are the oracle's float32 evaluation; the float64 one is recomputed from the golden inputs
        eic, wc = ei.cpu(), w.cpu()
        s64 = both(lambda c: R.gcn_norm_adj(R.SparseAdj(torch.stack([eic[1], eic[0]]), c(wc), [n, n])))[0]
        A64 = torch.zeros(n, n, dtype=torch.float64).index_put_((s64.row, s64.col), s64.edge_weight, accumulate=True)
        close(_dense(G), (A64, A), what=f"{name}: TF gcn_norm_adj")
        # PyG: remaining loops, degree by source
        G = ga.CSRGraph.from_edge_index(ei, n, w, remove_self_loops=True, add_self_loops=True,
                                        keep_loop_weight=True).gcn_norm("col")
        pi, pw = z[f"{name}/pyg_norm_index"], z[f"{name}/pyg_norm_weight"]
        if name == "multigraph_selfloop":
            continue  # two loops on one node: which weight survives is unordered in the reference too
        A = torch.zeros(n, n, dtype=torch.float64)
        A.index_put_((torch.from_numpy(pi[1]), torch.from_numpy(pi[0])), torch.from_numpy(pw).double(), accumulate=True)
        e64, w64 = both(lambda c: R.pyg_gcn_norm(eic, n, c(wc)))[0]
        A64 = torch.zeros(n, n, dtype=torch.float64).index_put_((e64[1], e64[0]), w64, accumulate=True)
        close(_dense(G), (A64, A), what=f"{name}: PyG norm")


def test_transpose_and_degree(dev):
    import graphgym_amd as ga
    g = torch.Generator().manual_seed(3)
    N, E = 200, 3000
    ei = torch.randint(0, N, (2, E), generator=g)
    w = torch.rand(E, generator=g)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N, w.to(dev))
    T = G.transpose()
    assert torch.equal(_dense(T), _dense(G).t())
    assert (T.val.cpu() == G.val.cpu()[T.pos.cpu().long()]).all()
    A = _dense(G)
    # _dense(G) is float64: A.sum is the float64 reference; a degree is one number per row -> per-entry 1e-5
    close(G.degree("row")[:, None], A.sum(1)[:, None], what="degree by destination")
    close(G.degree("col")[:, None], A.sum(0)[:, None], what="degree by source")
    assert (G.row_ids().cpu().numpy() == np.repeat(np.arange(N), np.diff(G.rowptr.cpu().numpy()))).all()


def test_symmetric_operators_are_their_own_transpose(dev, monkeypatch):
    """CSRGraph.is_symmetric (mp_csr_is_symmetric; MP_SYM_CHECK=1 lets transpose() ask): an undirected graph — both
    directions stored, as the reference's loaders produce (transform.py:11-38) — is recognised and transpose() returns
    the graph itself (no sort); mirrored
    entries with different values, a missing mirror, a repeated entry or a rectangular operator are not; the sorted
    transpose of a symmetric operator equals the operator field for field, and the backward pass through it gives the
    gradient of the sorted one."""
    import graphgym_amd as ga
    from graphgym_amd import ops, graph as G_
    g = torch.Generator().manual_seed(8)
    N, E = 3000, 20_000
    half = torch.randint(0, N, (2, E), generator=g)
    half = half[:, half[0] != half[1]]
    half = torch.unique(torch.cat([half, half.flip(0)], 1), dim=1)          # undirected, no repeats
    w_half = torch.rand(N, N, generator=g)
    w_sym = ((w_half + w_half.t()) / 2)[half[0], half[1]]
    monkeypatch.setenv("MP_SYM_CHECK", "1")
    for w in (None, w_sym):
        S = ga.CSRGraph.from_edge_index(half.to(dev), N, None if w is None else w.to(dev), add_self_loops=True)
        before = dict(G_.BUILDS)
        assert S.is_symmetric() and S.transpose() is S
        assert G_.BUILDS.get("transpose", 0) == before.get("transpose", 0)        # no sort ran
        T = S._transpose_sorted()
        assert torch.equal(T.rowptr, S.rowptr) and torch.equal(T.col, S.col)
        assert (T.val is None and S.val is None) or torch.equal(T.val, S.val)
        Sn = S.gcn_norm("row")
        assert Sn.symmetric and Sn.transpose() is Sn
        x = torch.randn(N, 32, generator=g).to(dev).requires_grad_(True)
        up = torch.randn(N, 32, generator=g).to(dev)
        ops.spmm(Sn, x, "sum").backward(up)
        want = ops.spmm(Sn._transpose_sorted(), up, "sum")
        assert float((x.grad - want).abs().max()) <= 1e-6 * float(want.abs().max())
    # not symmetric: one value changed / one direction dropped / one entry repeated / rectangular / switched off
    w_bad = w_sym.clone()
    w_bad[5] += 0.25
    assert not ga.CSRGraph.from_edge_index(half.to(dev), N, w_bad.to(dev)).is_symmetric()
    assert not ga.CSRGraph.from_edge_index(half[:, 1:].to(dev), N).is_symmetric()
    rep = torch.cat([half, half[:, :1]], 1)
    assert not ga.CSRGraph.from_edge_index(rep.to(dev), N).is_symmetric()
    A = ga.CSRGraph.from_edge_index(half[:, 1:].to(dev), N)
    assert A.transpose() is not A and A.transpose().pos is not None
    monkeypatch.delenv("MP_SYM_CHECK")                                       # the default: nobody asks
    S2 = ga.CSRGraph.from_edge_index(half.to(dev), N)
    assert not S2.is_symmetric() and S2.transpose() is not S2
    assert S2.is_symmetric(run=True) and S2.transpose() is S2               # ... unless told to


# --------------------------------------------------------------------------- aggregation
def test_aggregation_matches_golden(dev, golden):
    import graphgym_amd as ga
    from graphgym_amd import ops
    z = golden("aggregation.npz")
    for name in sorted({k.split("/")[0] for k in z.files}):
        n = int(z[f"{name}/n"])
        ei = torch.from_numpy(z[f"{name}/edge_index"]).to(dev)
        w = torch.from_numpy(z[f"{name}/w"]).to(dev)
        Gu = ga.CSRGraph.from_edge_index(ei, n)
        Gw = ga.CSRGraph.from_edge_index(ei, n, w)
        eic, wc = ei.cpu(), w.cpu()
        for d in (1, 3, 64):
            x = torch.from_numpy(z[f"{name}/x{d}"]).to(dev)
            xc = x.cpu()
            for red in ("sum", "mean", "max"):
                # float64: the oracle on the golden inputs; float32: the committed golden output itself
                r64 = both(lambda c: R.coo_aggregate(eic[1], eic[0], None, c(xc), n, red))[0]
                mag = None if red == "max" else mag_of(lambda c: R.coo_aggregate(eic[1], eic[0], None, c(xc).abs(), n, red))
                close(ops.spmm(Gu, x, red), (r64, torch.from_numpy(z[f"{name}/agg_{red}_d{d}"])), what=f"{name} {red} d{d}",
                      mag=mag)
                r64 = both(lambda c: R.coo_aggregate(eic[1], eic[0], c(wc), c(xc), n, red))[0]
                mag = None if red == "max" else mag_of(lambda c: R.coo_aggregate(eic[1], eic[0], c(wc).abs(), c(xc).abs(), n, red))
                close(ops.spmm(Gw, x, red), (r64, torch.from_numpy(z[f"{name}/aggw_{red}_d{d}"])), what=f"{name} w {red} d{d}",
                      mag=mag)


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 63, 64, 96, 100, 128, 130, 256, 260, 512, 1024])
def test_feature_widths(dev, d):
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(d)
    N, E = 257, 3000
    ei = torch.randint(0, N, (2, E), generator=g)
    w = torch.rand(E, generator=g)
    x = torch.randn(N, d, generator=g)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N, w.to(dev))
    for red in ("sum", "mean", "max"):
        # narrow rows are single signed sums that may cancel: also held against the sum of absolute terms (_tol rule d)
        mag = None if red == "max" else mag_of(lambda c: R.coo_aggregate(ei[1], ei[0], c(w), c(x).abs(), N, red))
        close(ops.spmm(G, x.to(dev), red), both(lambda c: R.coo_aggregate(ei[1], ei[0], c(w), c(x), N, red)), what=red,
              mag=mag)
    # a strided view (leading dimension > d) takes the same path
    xp = torch.randn(N, d + 4, generator=g)
    close(ops.spmm(G, xp.to(dev)[:, :d], "sum"), both(lambda c: R.coo_aggregate(ei[1], ei[0], c(w), c(xp[:, :d]), N, "sum")),
          what="strided view")


def test_ragged_edge_cases(dev):
    """empty graph, single node, all-empty rows (> 64 per segment), hubs split into pieces,
    duplicates and self loops, a hub as the very last row"""
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(99)
    cases = []
    cases.append((4, torch.zeros(2, 0, dtype=torch.int64)))                      # no edges at all
    cases.append((1, torch.zeros(2, 5, dtype=torch.int64)))                      # one node, 5 self loops
    cases.append((5000, torch.tensor([[1, 2], [4999, 4999]])))                   # 4998 empty rows then one row
    hub = torch.stack([torch.randint(0, 300, (9000,), generator=g), torch.full((9000,), 7)])
    rest = torch.randint(0, 300, (2, 2000), generator=g)
    cases.append((300, torch.cat([hub, rest], dim=1)))                           # degree 9000 > hub_deg
    last = torch.stack([torch.randint(0, 50, (5000,), generator=g), torch.full((5000,), 49)])
    cases.append((50, last))                                                     # hub is the last row
    two = torch.cat([torch.stack([torch.randint(0, 64, (3000,), generator=g), torch.full((3000,), 10)]),
                     torch.stack([torch.randint(0, 64, (4000,), generator=g), torch.full((4000,), 11)])], dim=1)
    cases.append((64, two))                                                      # adjacent hubs
    for N, ei in cases:
        E = ei.size(1)
        w = torch.rand(E, generator=g) + 0.5
        x = torch.randn(N, 256, generator=g)
        for ww in (None, w):
            G = ga.CSRGraph.from_edge_index(ei.to(dev), N, None if ww is None else ww.to(dev))
            for red in ("sum", "mean", "max"):
                close(ops.spmm(G, x.to(dev), red),
                      both(lambda c: R.coo_aggregate(ei[1], ei[0], None if ww is None else c(ww), c(x), N, red)),
                      what=f"ragged N={N} {red}")


def test_plan_covers_every_row_once(dev):
    """host logic of the segmentation, read back from the device plan"""
    import graphgym_amd as ga
    from graphgym_amd import graphgen
    ei = graphgen.ba_edge_index(20000, 5, seed=4, device=dev)
    G = ga.CSRGraph.from_edge_index(ei, 20000, add_self_loops=True)
    plan, counts = G.plan()
    n_seg, n_hub, n_piece, cap_hub, cap_piece, seg_cost, hub_deg, piece_edges = list(counts)
    p = plan.cpu().numpy()
    seg_row = p[16:16 + n_seg + 1]
    assert seg_row[0] == 0 and seg_row[-1] == 20000 and (np.diff(seg_row) >= 0).all()
    rowptr = G.rowptr.cpu().numpy()
    deg = np.diff(rowptr)
    cost = np.diff(rowptr[seg_row]) + 4 * np.diff(seg_row)
    nohub = deg.copy(); nohub[deg > hub_deg] = 0
    assert cost.max() <= seg_cost + nohub.max() + hub_deg + 4     # bounded work per wave (last row may be a hub)
    assert n_hub == int((deg > hub_deg).sum())
    hub_row = p[16 + n_seg + 1:16 + n_seg + 1 + n_hub]
    assert sorted(hub_row.tolist()) == np.nonzero(deg > hub_deg)[0].tolist()
    assert n_piece == int(np.ceil(deg[deg > hub_deg] / piece_edges).sum())


def test_argmax_is_bit_exact_with_ties(dev):
    import graphgym_amd as ga
    from graphgym_amd import ops, _lib
    g = torch.Generator().manual_seed(8)
    N, E, d = 60, 900, 8
    ei = torch.randint(0, N, (2, E), generator=g)
    x = torch.randint(0, 3, (N, d), generator=g).float()             # many ties
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N)
    y, arg = ops._raw_spmm(G, x.to(dev), _lib.MAX, want_argmax=True)
    rows = G.row_ids().cpu().long()
    cols = G.col.cpu().long()
    ref_arg = R.coo_aggregate_argmax(rows, cols, None, x, N)          # same CSR-sorted entry order
    assert torch.equal(arg.cpu().long(), ref_arg)
    assert torch.equal(y.cpu(), R.coo_aggregate(rows, cols, None, x, N, "max"))       # a max of inputs: exact


def test_epilogue_fusion(dev):
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(12)
    N, E, d = 500, 6000, 128
    ei = torch.randint(0, N, (2, E), generator=g)
    x = torch.randn(N, d, generator=g)
    b = torch.randn(d, generator=g)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N)
    ref = both(lambda c: torch.relu(R.coo_aggregate(ei[1], ei[0], None, c(x), N, "sum") + 1.25 * c(x) + c(b)))
    close(ops.spmm(G, x.to(dev), "sum", self_scale=1.25, bias=b.to(dev), relu=True), ref, what="fused epilogue")


@pytest.mark.parametrize("reduce", ["sum", "mean", "max"])
def test_backward_matches_oracle_autograd(dev, reduce):
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(21)
    N, E, d = 400, 5000, 96
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :1500] = 3                                                  # a hub row (and, transposed, a hub column)
    w = torch.rand(E, generator=g) + 0.2
    x = torch.randn(N, d, generator=g)
    b = torch.randn(d, generator=g)
    dy = torch.randn(N, d, generator=g)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N, w.to(dev))
    xg = x.to(dev).requires_grad_(True)
    bg = b.to(dev).requires_grad_(True)
    y = ops.spmm(G, xg, reduce, self_scale=0.5, bias=bg, relu=True)
    y.backward(dy.to(dev))
    mask = (y.detach() > 0).cpu()       # differentiate the oracle through the engine's ReLU pattern (inputs at ~0 are arbitrary)

    def ref(c):
        xr, br = c(x).clone().requires_grad_(True), c(b).clone().requires_grad_(True)
        if reduce == "max":  # route the gradient to the same (CSR-order-first) winner the kernel picks
            rows, cols = G.row_ids().cpu().long(), G.col.cpu().long()
            agg = R.coo_aggregate(rows, cols, c(G.val.cpu()), xr, N, "max")
        else:
            agg = R.coo_aggregate(ei[1], ei[0], c(w), xr, N, reduce)
        pre = agg + 0.5 * xr + br
        assert bool(((pre.detach() > 0) == mask)[pre.detach().abs() > 1e-5 * float(pre.detach().abs().max())].all())
        yr = pre * mask.to(pre.dtype)
        yr.backward(c(dy))
        return yr.detach(), xr.grad, br.grad
    r64, r32 = both(ref)
    close(y, (r64[0], r32[0]), what=f"{reduce} forward")
    close(xg.grad, (r64[1], r32[1]), what=f"{reduce} dx")
    close_all(bg.grad, (r64[2], r32[2]), what=f"{reduce} dbias")


def test_two_branch_aggregation(dev):
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(31)
    N, E, d = 600, 8000, 64
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :2000] = 5
    w = torch.rand(E, generator=g)
    x = torch.randn(N, d, generator=g)
    ids = torch.randperm(N, generator=g)[:40]
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N, w.to(dev))
    xg = x.to(dev).requires_grad_(True)
    P, Q = ops.idgnn_aggregate(G, ids.to(dev), xg)
    dP, dQ = torch.randn(N, d, generator=g), torch.randn(N, d, generator=g)
    (P * dP.to(dev) + Q * dQ.to(dev)).sum().backward()
    sel = torch.zeros(N, 1)
    sel[ids] = 1

    def ref(c):
        xr = c(x).clone().requires_grad_(True)
        Pr = R.coo_aggregate(ei[1], ei[0], c(w), xr, N, "sum")
        Qr = R.coo_aggregate(ei[1], ei[0], c(w), xr * c(sel), N, "sum")
        (Pr * c(dP) + Qr * c(dQ)).sum().backward()
        return Pr.detach(), Qr.detach(), xr.grad
    r64, r32 = both(ref)
    close(P, (r64[0], r32[0]), what="P = A x")
    close(Q, (r64[1], r32[1]), what="Q = A S x")
    close(xg.grad, (r64[2], r32[2]), what="two-branch dx")
    # rows with no identity neighbour are exactly zero
    has = torch.zeros(N).index_add_(0, ei[1], sel[ei[0]].view(-1)) > 0
    assert float(Q.detach().cpu()[~has].abs().max()) == 0.0


def test_identity_row_update(dev):
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(41)
    h = torch.randn(50, 20, generator=g)
    u = torch.randn(7, 20, generator=g)
    ids = torch.tensor([3, 9, 0, 49, 17, 21, 8])
    hg, ug = h.to(dev).requires_grad_(True), u.to(dev).requires_grad_(True)
    out = ops.index_add_rows(hg, ids.to(dev), ug)
    assert torch.equal(out.detach().cpu(), h.index_add(0, ids, u))                 # one add per element: exact
    out.sum().backward()
    assert torch.equal(hg.grad.cpu(), torch.ones(50, 20)) and torch.equal(ug.grad.cpu(), torch.ones(7, 20))
    xs = ops.gather_rows(hg, ids.to(dev))
    assert torch.equal(xs.detach().cpu(), h[ids])


# --------------------------------------------------------------------------- attention
def test_attention_pieces(dev):
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(51)
    N, E, d, H = 120, 1500, 32, 4
    ei = torch.randint(0, N, (2, E), generator=g)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N, add_self_loops=True)
    rows, cols = G.row_ids().cpu().long(), G.col.cpu().long()
    Q, K, V = (torch.randn(N, d, generator=g) for _ in range(3))
    Qg, Kg, Vg = (t.to(dev).requires_grad_(True) for t in (Q, K, V))
    s = ops.sddmm_dot(G, Qg, Kg, H, 0.5)
    p = ops.edge_softmax(G, s)
    y = ops.spmm_edge_values(G, p, Vg, H)
    dy = torch.randn(N, d, generator=g)
    y.backward(dy.to(dev))
    dh = d // H

    def ref(c):
        Qr, Kr, Vr = (c(t).clone().requires_grad_(True) for t in (Q, K, V))
        sr = (Qr[rows].view(-1, H, dh) * Kr[cols].view(-1, H, dh)).sum(-1) * 0.5
        pr = R.softmax(sr, rows, N)
        yr = torch.zeros(N, H, dh, dtype=sr.dtype).index_add_(0, rows, pr.unsqueeze(-1) * Vr[cols].view(-1, H, dh)).view(N, d)
        yr.backward(c(dy))
        return sr.detach(), pr.detach(), yr.detach(), Qr.grad, Kr.grad, Vr.grad
    r64, r32 = both(ref)
    for got, i, what in ((s, 0, "scores"), (p, 1, "softmax"), (y, 2, "attention output"), (Qg.grad, 3, "dQ"),
                         (Kg.grad, 4, "dK"), (Vg.grad, 5, "dV")):
        close(got, (r64[i], r32[i]), what=what)
    # additive scores
    ai, aj = torch.randn(N, generator=g), torch.randn(N, generator=g)
    aig, ajg = ai.to(dev).requires_grad_(True), aj.to(dev).requires_grad_(True)
    sa = ops.sddmm_add(G, aig, ajg, 0.2)
    ds = torch.randn(G.nnz, 1, generator=g)
    sa.backward(ds.to(dev))
    def ref_add(c):
        air, ajr = c(ai).clone().requires_grad_(True), c(aj).clone().requires_grad_(True)
        sar = torch.nn.functional.leaky_relu(air[rows] + ajr[cols], 0.2).view(-1, 1)
        sar.backward(c(ds))
        return sar.detach(), air.grad, ajr.grad
    r64, r32 = both(ref_add)
    close(sa, (r64[0], r32[0]), what="additive scores")
    # one number per node, the sum of that node's signed per-entry terms ds_e * lrelu'(z_e): a width-1 result that
    # cancels by construction, held against its sum of ABSOLUTE terms (rule (d) of tests/_tol.py) — the same closure
    # with |ds| (lrelu' > 0)
    def mag_add(c):
        air, ajr = c(ai).clone().requires_grad_(True), c(aj).clone().requires_grad_(True)
        torch.nn.functional.leaky_relu(air[rows] + ajr[cols], 0.2).view(-1, 1).backward(c(ds).abs())
        return air.grad, ajr.grad
    m64 = both(mag_add)[0]
    close(aig.grad[:, None], (r64[1][:, None], r32[1][:, None]), what="d a_dst", mag=m64[0][:, None])
    close(ajg.grad[:, None], (r64[2][:, None], r32[2][:, None]), what="d a_src", mag=m64[1][:, None])


# --------------------------------------------------------------------------- full-size properties
def test_full_size_properties_c2_and_c4(dev):
    """BASELINE sizes: C2 (1M nodes / ~11M stored entries) and C4 (10M / ~110M), d = 256.
    Size-independent checks: ones -> weighted degree, a float64 checksum through the column
    sums, linearity, and an exact oracle comparison on a random sample of rows."""
    import graphgym_amd as ga
    from graphgym_amd import ops, graphgen
    for n in (1_000_000, 10_000_000):
        ei = graphgen.ba_edge_index(n, 5, seed=12345, device=dev)
        G = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
        del ei
        d = 256
        ones = torch.ones(n, d, device=dev)
        y = ops.spmm(G, ones, "sum")
        rows_all = G.row_ids().long()
        deg64 = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, rows_all, G.val.double())
        assert float(((y[:, 0].double() - deg64).abs() / deg64).max()) <= 1e-5      # per row, against its own float64 sum
        del rows_all
        deg = G.degree("row")                                            # the library's own K6 kernel
        assert float(((deg.double() - deg64).abs() / deg64).max()) <= 1e-5
        assert float((y - y[:, :1]).abs().max()) == 0.0                 # every column identical
        del ones, y
        gen = torch.Generator(device=dev).manual_seed(7)
        x = torch.rand(n, d, device=dev, generator=gen) * 2 - 1
        y = ops.spmm(G, x, "sum")
        # checksum of checksums: sum_i y[i,:] == sum_j colsum[j] * x[j,:]
        lhs = y.double().sum(0)
        rhs = (G.degree("col").double()[:, None] * x.double()).sum(0)
        assert float((lhs - rhs).abs().max()) <= 1e-6 * float(rhs.abs().max() + n ** 0.5)
        # linearity
        y2 = ops.spmm(G, x * 3.0, "sum")
        assert float((y2 - 3.0 * y).abs().max()) <= 1e-5 * float(y.abs().max())
        del y2
        # oracle on a sample of rows (gathered from the CSR itself)
        rows = torch.randint(0, n, (512,), device=dev, generator=gen)
        rows = torch.cat([rows, torch.arange(0, 8, device=dev)])        # the hubs too
        rp = G.rowptr.long()
        r64, r32 = [], []
        for r in rows.tolist():
            s, e = int(rp[r]), int(rp[r + 1])
            val, xg = G.val[s:e, None].cpu(), x[G.col[s:e].long()].cpu()
            seg = torch.zeros(e - s, dtype=torch.int64)
            r32.append(torch.zeros(1, d).index_add_(0, seg, val * xg))
            r64.append(torch.zeros(1, d, dtype=torch.float64).index_add_(0, seg, val.double() * xg.double()))
        close(y[rows], (torch.cat(r64), torch.cat(r32)), what=f"n={n}: sampled rows")
        del x, y, G
        torch.cuda.empty_cache()


@pytest.mark.parametrize("d", [64, 128, 256, 512])
def test_fused_eval_epilogue(dev, d):
    """conv bias + BatchNorm1d(eval) + ReLU + row L2-normalise folded into the flush
    (graphgym/models/layer.py:26-47, gnn.py:79-80) vs the unfused torch ops on the oracle's aggregate"""
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(d)
    N, E = 900, 12000
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :3000] = 11                                                # a hub row goes through the finalize kernel
    x = torch.randn(N, d, generator=g)
    bn = torch.nn.BatchNorm1d(d, eps=1e-5)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(d, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(d, generator=g))
        bn.running_mean.copy_(torch.randn(d, generator=g))
        bn.running_var.copy_(torch.rand(d, generator=g) + 0.5)
    bn.eval()
    conv_bias = torch.randn(d, generator=g)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N)
    def ref(c):
        agg = R.coo_aggregate(ei[1], ei[0], None, c(x), N, "sum") + 0.5 * c(x) + c(conv_bias)
        h = torch.nn.functional.batch_norm(agg, c(bn.running_mean), c(bn.running_var), c(bn.weight.detach()),
                                           c(bn.bias.detach()), False, 0.1, bn.eps)
        return torch.nn.functional.normalize(torch.relu(h), p=2, dim=-1)
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias + (conv_bias - bn.running_mean) * scale
    y = ops.spmm_fused_eval(G, x.to(dev), "sum", self_scale=0.5, col_scale=scale.detach().to(dev),
                            col_shift=shift.detach().to(dev), relu=True, l2norm=True)
    close(y, both(ref), what="folded eval epilogue")


def test_randomised_graphs_under_a_tiny_plan(dev):
    """Stress the segmentation and the hub split: with seg_cost / hub_deg / piece_edges forced down to 64
    every moderately long row becomes a hub cut into pieces and every segment holds a handful of rows.
    60 random graphs x 3 reduces x weighted/unweighted against the oracle evaluated in float64 (rows with
    thousands of same-sign terms: the fp32 oracle's own running sum is only good to ~1e-5 there)."""
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(2024)
    ga.CSRGraph.PLAN_CONFIG = (64, 1, 64, 64)       # handed to mp_spmm_plan_build per call: no library-wide state
    try:
        for trial in range(60):
            N = int(torch.randint(1, 400, (1,), generator=g))
            E = int(torch.randint(0, 6000, (1,), generator=g))
            d = [1, 3, 4, 8, 33, 64, 100, 128, 256, 320][trial % 10]
            ei = torch.randint(0, N, (2, E), generator=g)
            if E and trial % 3 == 0:                                  # concentrate destinations: long rows
                ei[1] = ei[1] % max(1, N // 20)
            if E and trial % 7 == 0:                                  # long runs of empty rows
                ei[1] = (ei[1] // 50) * 50
            w = torch.rand(E, generator=g) - 0.3
            x = torch.randn(N, d, generator=g)
            for ww in (None, w):
                G = ga.CSRGraph.from_edge_index(ei.to(dev), N, None if ww is None else ww.to(dev))
                for red in ("sum", "mean", "max"):
                    mag = None if red == "max" else mag_of(lambda c: R.coo_aggregate(
                        ei[1], ei[0], None if ww is None else c(ww).abs(), c(x).abs(), N, red))
                    close(ops.spmm(G, x.to(dev), red),
                          both(lambda c: R.coo_aggregate(ei[1], ei[0], None if ww is None else c(ww), c(x), N, red)),
                          what=f"tiny plan, trial {trial} {red}", mag=mag)
            ids = torch.randperm(N, generator=g)[:max(1, N // 10)]
            G = ga.CSRGraph.from_edge_index(ei.to(dev), N, w.to(dev))
            P, Q = ops.idgnn_aggregate(G, ids.to(dev), x.to(dev))
            sel = torch.zeros(N, 1); sel[ids] = 1
            close(P, both(lambda c: R.coo_aggregate(ei[1], ei[0], c(w), c(x), N, "sum")), what=f"tiny plan P, trial {trial}",
                  mag=mag_of(lambda c: R.coo_aggregate(ei[1], ei[0], c(w).abs(), c(x).abs(), N, "sum")))
            close(Q, both(lambda c: R.coo_aggregate(ei[1], ei[0], c(w), c(x * sel), N, "sum")), what=f"tiny plan Q, trial {trial}",
                  mag=mag_of(lambda c: R.coo_aggregate(ei[1], ei[0], c(w).abs(), c(x * sel).abs(), N, "sum")))
    finally:
        ga.CSRGraph.PLAN_CONFIG = None


@pytest.mark.parametrize("M,F,d", [(1, 8, 4), (127, 16, 64), (128, 64, 128), (129, 72, 100), (1000, 256, 256),
                                   (777, 264, 132), (300, 128, 512), (4100, 8, 260),
                                   (500, 1433, 128), (333, 1, 7), (64, 13, 33), (2000, 30, 1)])   # scalar-loader shapes
@pytest.mark.parametrize("dual", [False, True])
def test_dense_fused_mfma_kernel(dev, M, F, d, dual):
    """act(P @ W [+ Q @ W_id] + bias) on the f32 MFMA kernel vs float64 on the host (forward), and its
    autograd (library GEMMs) vs torch's own autograd of the unfused expression"""
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(M + F + d)
    P, Q = torch.randn(M, F, generator=g), torch.randn(M, F, generator=g)
    W, Wi = torch.randn(F, d, generator=g) / F ** 0.5, torch.randn(F, d, generator=g) / F ** 0.5
    b = torch.randn(d, generator=g)
    args = [t.to(dev).requires_grad_(True) for t in (P, W, Q, Wi, b)]
    out = ops.dense_fused(args[0], args[1], args[2] if dual else None, args[3] if dual else None, args[4], relu=True)
    dy = torch.randn(M, d, generator=g)
    out.backward(dy.to(dev))
    mask = (out.detach() > 0).cpu()                      # the oracle differentiates through the engine's ReLU pattern

    def ref(c):
        refs = [c(t).clone().requires_grad_(True) for t in (P, W, Q, Wi, b)]
        r = refs[0] @ refs[1] + refs[4]
        if dual:
            r = r + refs[2] @ refs[3]
        assert bool(((r.detach() > 0) == mask)[r.detach().abs() > 1e-5 * float(r.detach().abs().max())].all())
        o = r * mask.to(r.dtype)
        o.backward(c(dy))
        return [o.detach()] + [t.grad for t in refs]
    r64, r32 = both(ref)
    # sums of absolute terms (rule d of tests/_tol.py): with F = 1 or d = 1 a "row" is one signed dot product
    mag_out = P.double().abs() @ W.double().abs() + b.double().abs() + (Q.double().abs() @ Wi.double().abs() if dual else 0)
    gm = (dy.double() * mask).abs()
    mags = {"P": gm @ W.double().abs().t(), "Q": gm @ Wi.double().abs().t()}
    close(out, (r64[0], r32[0]), what="transform forward", mag=mag_out)
    for i, (a, name) in enumerate(zip(args, "P W Q Wid b".split())):
        if not dual and name in ("Q", "Wid"):
            continue
        # input gradients are per row; weight / bias gradients are reductions over all M rows: one scale
        if name in ("P", "Q"):
            close(a.grad, (r64[1 + i], r32[1 + i]), what=f"transform d{name}", mag=mags[name])
        else:
            close_all(a.grad, (r64[1 + i], r32[1 + i]), what=f"transform d{name}")
    # the raw kernel really ran for these shapes (no library fallback)
    assert ops._raw_dense_fused(P.to(dev), W.to(dev), None, None, None, False) is not None


def test_c_abi_error_codes_on_device(dev):
    """status codes, not exceptions or faults: too-small workspace, bad enums, bad leading dimensions"""
    import ctypes as C
    import graphgym_amd as ga
    from graphgym_amd import _lib
    from graphgym_amd._lib import ptr
    from graphgym_amd.graph import _stream
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    N, d = 64, 32
    ei = torch.stack([torch.randint(0, N, (5000,), generator=g), torch.zeros(5000, dtype=torch.int64)])  # one hub row
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N)
    plan, counts = G.plan()
    assert counts[1] == 1 and counts[2] > 0                      # the hub path is live, so a workspace is required
    x = torch.randn(N, d, device=dev)
    y = torch.empty(N, d, device=dev)
    args = lambda ws, nb, reduce=0, ldx=d, act=0: L.mp_spmm_csr_f32(
        ptr(G.rowptr), ptr(G.col), None, N, ptr(plan), counts, ptr(x), ldx, ptr(y), d, d, reduce, None, 0, 0.0,
        None, act, None, ws, nb, _stream())
    nb = C.c_size_t(0)
    assert L.mp_spmm_ws_bytes(counts, d, 0, 0, C.byref(nb)) == 0 and nb.value > 0
    ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
    assert args(ptr(ws), nb.value) == 0
    assert args(None, 0) == 3                                    # MP_ERR_WORKSPACE
    assert args(ptr(ws), nb.value - 1) == 3
    assert args(ptr(ws), nb.value, reduce=7) == 1                # MP_ERR_INVALID_ARG
    assert args(ptr(ws), nb.value, ldx=d - 1) == 1
    assert args(ptr(ws), nb.value, act=9) == 1
    with pytest.raises(_lib.EngineError, match="workspace"):
        _lib.check(args(None, 0), "mp_spmm_csr_f32")
    torch.cuda.synchronize()
    # fused-epilogue L2 normalisation refuses rows wider than one wave instead of computing a partial norm
    xw = torch.randn(N, 512, device=dev)
    yw = torch.empty(N, 512, device=dev)
    nbw = C.c_size_t(0)
    L.mp_spmm_ws_bytes(counts, 512, 0, 0, C.byref(nbw))
    wsw = torch.empty(nbw.value, dtype=torch.uint8, device=dev)
    st = L.mp_spmm_csr_epilogue_f32(ptr(G.rowptr), ptr(G.col), None, N, ptr(plan), counts, ptr(xw), 512, ptr(yw), 512,
                                    512, 0, None, 0, 0.0, None, None, 0, 1, 1e-12, ptr(wsw), nbw.value, _stream())
    assert st == 2                                               # MP_ERR_UNSUPPORTED
    # dense kernel: inconsistent operands are rejected
    P = torch.randn(10, 12, device=dev); W = torch.randn(12, 8, device=dev); out = torch.empty(10, 8, device=dev)
    assert L.mp_dense_fused_f32(ptr(P), 12, ptr(W), ptr(P), 12, None, None, 0, ptr(out), 8, 10, 12, 8, _stream()) == 1
    assert L.mp_dense_fused_f32(ptr(P), 11, ptr(W), None, 0, None, None, 0, ptr(out), 8, 10, 12, 8, _stream()) == 1


def test_by_source_normalisation_is_bitwise_reproducible(dev):
    """gcn_norm('col') (GCNIDConvLayer.norm, idconv.py:132-148: degrees scattered by SOURCE) sums in a fixed order over
    the transposed CSR: two builds are bit-identical, and match the oracle"""
    import graphgym_amd as ga
    g = torch.Generator().manual_seed(77)
    N, E = 3000, 60000
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[0, :5000] = 3                                               # a source with thousands of out-edges
    w = torch.rand(E, generator=g) + 0.01
    builds = []
    for _ in range(2):
        G = ga.CSRGraph.from_edge_index(ei.to(dev), N, w.to(dev), remove_self_loops=True, add_self_loops=True,
                                        keep_loop_weight=True).gcn_norm("col")
        builds.append(G.val.clone())
    assert torch.equal(builds[0], builds[1])
    def ref(c):
        ei_r, norm_r = R.pyg_gcn_norm(ei, N, c(w))
        return torch.zeros(N, N, dtype=norm_r.dtype).index_put_((ei_r[1], ei_r[0]), norm_r, accumulate=True)
    close(_dense(G), both(ref), what="PyG normalisation by source degree")


@pytest.mark.parametrize("heads,d", [(2, 64), (4, 256), (8, 128), (4, 24)])
def test_multi_head_aggregation_in_one_launch(dev, heads, d):
    """Y[r, slice h] = sum_e a[e, h] V[col_e, slice h] for all heads in ONE launch of the plan-based kernel
    (mp_spmm_csr_heads_f32), hub rows included, against float64; and the gradients of spmm_edge_values through it"""
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(heads * 100 + d)
    N, E = 900, 14000
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :3000] = 17                                              # a hub row: pieces + finalize
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N)
    a = torch.rand(G.nnz, heads, generator=g)
    V = torch.randn(N, d, generator=g)
    calls = []
    from graphgym_amd import _lib
    L = _lib.lib()
    y = ops._raw_spmm_heads(G, a.to(dev), V.to(dev), heads)
    rows, cols = G.row_ids().cpu().long(), G.col.cpu().long()
    dh = d // heads
    w = a.double().repeat_interleave(dh, dim=1)                    # [nnz, d]: head h's weight on its dh columns
    ref = torch.zeros(N, d, dtype=torch.float64).index_add_(0, rows, w * V.double()[cols])
    ad, Vd = a.to(dev).requires_grad_(True), V.to(dev).requires_grad_(True)
    up = torch.randn(N, d, generator=g)
    ops.spmm_edge_values(G, ad, Vd, heads).backward(up.to(dev))

    def ref_fn(c):
        ar, Vr = c(a).clone().requires_grad_(True), c(V).clone().requires_grad_(True)
        o = torch.zeros(N, d, dtype=ar.dtype).index_add_(0, rows, ar.repeat_interleave(dh, dim=1) * Vr[cols])
        o.backward(c(up))
        return o.detach(), ar.grad, Vr.grad
    r64, r32 = both(ref_fn)
    assert float((r64[0] - ref).abs().max()) == 0.0
    close(y, (r64[0], r32[0]), what="multi-head aggregation")
    close(ad.grad, (r64[1], r32[1]), what="d alpha")
    close(Vd.grad, (r64[2], r32[2]), what="dV")


@pytest.mark.parametrize("heads", [1, 3])
def test_row_softmax_with_hub_rows(dev, heads):
    """mp_csr_row_softmax_f32 / _bwd_f32 with rows far beyond the 16-lane groups' reach (40 000, 3 000 and 2 049 entries
    among short and empty rows): those rows are taken by the whole workgroup (attn.hip: softmax_row_loop).  Forward and
    backward against the oracle's softmax (TfgIDLayer.py:333-355 through tf_sparse segment softmax), every row summing
    to one, reproducible bit for bit."""
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(60 + heads)
    N = 40_000
    dst = torch.cat([torch.randint(0, N, (150_000,), generator=g), torch.full((40_000,), 7), torch.full((3_000,), N - 1),
                     torch.full((2_049,), 20_000)])
    dst = dst[(dst % 11) != 3]                                           # empty rows
    src = torch.randint(0, N, (dst.numel(),), generator=g)
    G = ga.CSRGraph.from_edge_index(torch.stack([dst, src]).to(dev), N, dst_row=0)
    rows = G.row_ids().cpu().long()
    assert int(torch.bincount(rows, minlength=N).max()) >= 40_000
    s = torch.randn(G.nnz, heads, generator=g) * 3
    up = torch.randn(G.nnz, heads, generator=g)
    sg = s.to(dev).requires_grad_(True)
    p = ops.edge_softmax(G, sg)
    p.backward(up.to(dev))

    def ref(c):
        sr = c(s).clone().requires_grad_(True)
        pr = R.softmax(sr, rows, N)
        pr.backward(c(up))
        return pr.detach(), sr.grad
    r64, r32 = both(ref)
    close(p, (r64[0], r32[0]), what="softmax with hub rows")
    al, dl = r64[0], up.double()
    rowdot = torch.zeros(N, heads, dtype=torch.float64).index_add_(0, rows, (al * dl).abs())
    close(sg.grad, (r64[1], r32[1]), what="softmax backward with hub rows", mag=al.abs() * (dl.abs() + rowdot[rows]))
    sums = torch.zeros(N, heads, dtype=torch.float64).index_add_(0, rows, p.detach().cpu().double())
    has = torch.bincount(rows, minlength=N) > 0
    assert float((sums[has] - 1.0).abs().max()) <= 1e-5
    p2 = ops.edge_softmax(G, sg.detach())
    assert torch.equal(p.detach(), p2)


@pytest.mark.parametrize("heads", [1, 4])
def test_additive_attention_coefficients_in_one_pass(dev, heads):
    """alpha = softmax_row(leaky_relu(a_dst[row] + a_src[col])) for all heads in one launch (mp_gat_alpha_f32), forward
    and the gradients to both per-node terms, against torch autograd through the oracle's softmax (idconv.py:319-327)"""
    import graphgym_amd as ga
    from graphgym_amd import ops
    g = torch.Generator().manual_seed(40 + heads)
    N, E = 700, 9000
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[1, :2500] = 3
    G = ga.CSRGraph.from_edge_index(ei.to(dev), N, remove_self_loops=True, add_self_loops=True)
    rows, cols = G.row_ids().cpu().long(), G.col.cpu().long()
    a_dst, a_src = torch.randn(N, heads, generator=g), torch.randn(N, heads, generator=g)
    dal = torch.randn(G.nnz, heads, generator=g)
    adg, asg = a_dst.to(dev).requires_grad_(True), a_src.to(dev).requires_grad_(True)
    alpha = ops.gat_alpha(G, adg, asg, 0.2)
    alpha.backward(dal.to(dev))
    def ref_fn(c):
        adr, asr = c(a_dst).clone().requires_grad_(True), c(a_src).clone().requires_grad_(True)
        s = torch.nn.functional.leaky_relu(adr[rows] + asr[cols], 0.2)
        r = R.softmax(s, rows, N)
        r.backward(c(dal))
        return r.detach(), adr.grad, asr.grad
    r64, r32 = both(ref_fn)
    close(alpha, (r64[0], r32[0]), what="additive attention coefficients")
    # the gradient of a softmax row sums to zero: d a_dst[i] = sum_e ds_e * lrelu' is a sum that cancels BY CONSTRUCTION
    # (exactly, when the slope is constant over the row).  Its natural scale is the sum of absolute terms
    # |alpha_e| (|dalpha_e| + sum_row |alpha dalpha|)  (rule d of tests/_tol.py)
    al, dl = r64[0], dal.double()
    rowdot = torch.zeros(N, heads, dtype=torch.float64).index_add_(0, rows, (al * dl).abs())
    terms = al.abs() * (dl.abs() + rowdot[rows])
    mag_dst = torch.zeros(N, heads, dtype=torch.float64).index_add_(0, rows, terms)
    mag_src = torch.zeros(N, heads, dtype=torch.float64).index_add_(0, cols, terms)
    close(adg.grad, (r64[1], r32[1]), what="d a_dst", mag=torch.maximum(mag_dst, r64[1].abs()))
    close(asg.grad, (r64[2], r32[2]), what="d a_src", mag=torch.maximum(mag_src, r64[2].abs()))
    sums = torch.zeros(N, heads, dtype=torch.float64).index_add_(0, rows, alpha.detach().cpu().double())
    assert float((sums - 1.0).abs().max()) <= 1e-5                  # every row has its self loop: coefficients sum to 1


# --------------------------------------------------------------------------- the aggregation on the tile structure
@pytest.mark.parametrize("n,E,d,reduce,weighted,self_scale,hubs", [
    (70_001, 600_000, 256, "sum", True, 0.0, True),
    (66_000, 400_000, 256, "mean", False, 0.0, False),
    (65_536, 500_000, 512, "sum", True, 1.5, True),
    (90_017, 300_000, 512, "mean", True, 0.0, False),
    (131_072, 0 + 64, 256, "sum", False, 0.0, False),          # almost every tile without a single entry
    (80_003, 500_000, 128, "sum", True, 0.5, True),
    (65_537, 200_000, 128, "mean", False, 0.0, False),
])
def test_tile_aggregation_matches_oracle_and_the_plan_kernel(dev, monkeypatch, n, E, d, reduce, weighted, self_scale, hubs):
    """mp_agg_rows_tiles_f32 (what ops._raw_spmm dispatches for plain sum / mean at d = 128 / 256 / 512 from 2^21 rows):
    against the float64 oracle per row, against the plan-based kernel (same values up to the order of a cut row's
    partial sums), bitwise reproducible, and actually taken (MP_AGG_TILES=0 gives the plan kernel's bits)."""
    import graphgym_amd as ga
    from graphgym_amd import ops, _lib
    g = torch.Generator().manual_seed(n + d)
    dst = torch.randint(0, n, (E,), generator=g)
    src = torch.randint(0, n, (E,), generator=g)
    if hubs:                                           # a few rows with tens of thousands of entries: cut between waves
        k = E // 3
        dst[:k] = torch.randint(0, 4, (k,), generator=g) * 1000 + 17
    ei = torch.stack([dst, src])
    w = (torch.rand(E, generator=g) + 0.1) if weighted else None
    x = torch.randn(n, d, generator=g)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, None if w is None else w.to(dev), dst_row=0)
    red = _lib.REDUCE[reduce]
    xd = x.to(dev)
    S = xd if self_scale else None
    monkeypatch.setenv("MP_AGG_TILES", "1")
    monkeypatch.setattr(ops, "AGG_TILES_MIN_ROWS", 1)          # (the product dispatches it from 2^21 rows: test_configs_gpu C4 / C5)
    before = ops.AGG_TILES_CALLS
    y1, _ = ops._raw_spmm(G, xd, red, S=S, self_scale=self_scale)
    y1b, _ = ops._raw_spmm(G, xd, red, S=S, self_scale=self_scale)
    assert ops.AGG_TILES_CALLS == before + 2                    # the tile kernel ran
    monkeypatch.setenv("MP_AGG_TILES", "0")
    y0, _ = ops._raw_spmm(G, xd, red, S=S, self_scale=self_scale)
    assert ops.AGG_TILES_CALLS == before + 2                    # ... and this was the plan-based one
    assert torch.equal(y1, y1b)

    def ref(c):
        adj = R.SparseAdj(ei, None if w is None else c(w), [n, n])
        agg = adj @ c(x)
        if reduce == "mean":
            cnt = torch.zeros(n, dtype=agg.dtype).index_add_(0, ei[0], torch.ones(E, dtype=agg.dtype))
            wsum = agg if w is None else torch.zeros_like(agg).index_add_(0, ei[0], c(x)[ei[1]] * c(w)[:, None])
            agg = wsum / cnt.clamp(min=1)[:, None]
        return agg + self_scale * c(x)
    refs = both(ref)
    close(y1, refs, what="tile aggregation")
    close(y0, refs, what="plan-based aggregation")
    diff = (y1 - y0).abs().amax(dim=1)
    scale = y0.abs().amax(dim=1).clamp(min=1e-30)
    assert float((diff / scale).max()) < 1e-5


@pytest.mark.parametrize("n,E,d,weighted,hubs", [
    (5000, 60_000, 256, True, False), (70_001, 300_000, 256, False, True), (4097, 3000, 128, True, False),
    (20_000, 200_000, 512, True, True), (64, 0, 256, False, False), (1000, 1, 128, False, False),
])
def test_tile_max_equals_the_plan_kernel_bit_for_bit(dev, monkeypatch, n, E, d, weighted, hubs):
    """reduce = max on the tile structure (mp_agg_rows_tiles_f32, values only — what ops.spmm dispatches when nothing is
    differentiated): a maximum does not depend on the order of its terms, so the rows equal the plan-based kernel's bit
    for bit (whose argmax the backward pass keeps using), rows without entries are 0 (generalconv.py:18 'max' through
    scatter's fill), and every value is the float32 maximum of w_ij * x[j] exactly."""
    import graphgym_amd as ga
    from graphgym_amd import ops, _lib
    g = torch.Generator().manual_seed(n + d + 1)
    dst = torch.randint(0, n, (E,), generator=g)
    src = torch.randint(0, n, (E,), generator=g)
    if hubs:
        k = E // 3
        dst[:k] = torch.randint(0, 4, (k,), generator=g) * 1000 + 17
    ei = torch.stack([dst, src])
    w = (torch.rand(E, generator=g) + 0.1) if weighted else None
    x = torch.randn(n, d, generator=g) - 1.5                   # mostly negative: an empty row (0) differs from any maximum
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, None if w is None else w.to(dev), dst_row=0)
    xd = x.to(dev)
    monkeypatch.setenv("MP_AGG_TILES", "1")
    monkeypatch.setattr(ops, "AGG_TILES_MIN_ROWS", 1)
    before = ops.AGG_TILES_CALLS
    with torch.no_grad():
        y1 = ops.spmm(G, xd, "max")
    want_calls = before + (1 if E > 0 else 0)                  # (an operator without entries stays with the plan kernel)
    assert ops.AGG_TILES_CALLS == want_calls
    y0, arg = ops._raw_spmm(G, xd, _lib.MAX, want_argmax=True)
    assert ops.AGG_TILES_CALLS == want_calls                    # an argmax is the plan-based kernel's
    assert torch.equal(y1, y0)
    msg = x[src] * (w[:, None] if weighted else 1.0)
    want = torch.zeros(n, d).scatter_reduce(0, dst[:, None].expand(E, d), msg, "amax", include_self=False) if E else \
        torch.zeros(n, d)
    assert torch.equal(y1.cpu(), want)
    # differentiated: the registered operator, argmax and all
    xg = xd.clone().requires_grad_(True)
    y2 = ops.spmm(G, xg, "max")
    assert ops.AGG_TILES_CALLS == want_calls and torch.equal(y2.detach(), y0)


@pytest.mark.parametrize("n,E,d,weighted,hubs,n_id", [
    (70_001, 600_000, 256, True, True, 700), (66_000, 400_000, 256, False, False, 3000), (65_536, 500_000, 512, True, True, 650),
    (80_003, 300_000, 128, True, False, 1), (4100, 9000, 256, False, False, 4100),
])
def test_two_branch_aggregation_on_the_tile_structure(dev, monkeypatch, n, E, d, weighted, hubs, n_id):
    """(P, Q) = (A x, A S x) (gcn_id, TfgIDLayer.py:510-517) as ops.idgnn_aggregate dispatches it from 2^21 rows at
    d = 128 / 256 / 512: P in the pass of the tile kernel, which also writes the zero rows of Q; the rows of Q next to an
    identity node by mp_id_rows_f32.  Against the float64 oracle, against the one-pass plan-based kernel, P bit-equal to
    the plain tile aggregation, Q exactly zero off the identity nodes' neighbours, gradients through both branches."""
    import graphgym_amd as ga
    from graphgym_amd import ops, _lib
    g = torch.Generator().manual_seed(n + d + 2)
    dst = torch.randint(0, n, (E,), generator=g)
    src = torch.randint(0, n, (E,), generator=g)
    if hubs:
        k = E // 3
        dst[:k] = torch.randint(0, 4, (k,), generator=g) * 1000 + 17
        src[k:2 * k] = torch.randint(0, 4, (k,), generator=g) * 1000 + 17     # hubs as sources: identity entries in many rows
    ei = torch.stack([dst, src])
    w = (torch.rand(E, generator=g) + 0.1) if weighted else None
    x = torch.randn(n, d, generator=g)
    ids = torch.randperm(n, generator=g)[:n_id]
    if hubs:
        ids[:2] = torch.tensor([17, 1017])
        ids = torch.unique(ids)
    G = ga.CSRGraph.from_edge_index(ei.to(dev), n, None if w is None else w.to(dev), dst_row=0)
    xd, idd = x.to(dev), ids.to(dev)
    monkeypatch.setenv("MP_AGG_TILES", "1")
    monkeypatch.setattr(ops, "AGG_TILES_MIN_ROWS", 1)
    before = ops.AGG_TILES_CALLS
    xg = xd.clone().requires_grad_(True)
    P, Q = ops.idgnn_aggregate(G, idd, xg)
    assert ops.AGG_TILES_CALLS == before + 1
    dP, dQ = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
    (P * dP.to(dev) + Q * dQ.to(dev)).sum().backward()
    P, Q = P.detach(), Q.detach()
    before = ops.AGG_TILES_CALLS                                 # (the backward pass aggregated on tiles too)
    P2, Q2 = ops.idgnn_aggregate(G, idd, xd)
    assert torch.equal(P, P2) and torch.equal(Q, Q2)             # reproducible
    y, _ = ops._raw_spmm(G, xd, _lib.SUM)
    assert ops.AGG_TILES_CALLS == before + 2 and torch.equal(P, y)
    monkeypatch.setenv("MP_AGG_TILES", "0")
    P0, Q0 = ops.idgnn_aggregate(G, idd, xd)
    assert ops.AGG_TILES_CALLS == before + 2                     # ... and this was the plan-based kernel
    sel = torch.zeros(n, 1)
    sel[ids] = 1

    def ref(c):
        xr = c(x).clone().requires_grad_(True)
        adj = R.SparseAdj(ei, None if w is None else c(w), [n, n])
        Pr, Qr = adj @ xr, adj @ (xr * c(sel))
        (Pr * c(dP) + Qr * c(dQ)).sum().backward()
        return Pr.detach(), Qr.detach(), xr.grad
    r64, r32 = both(ref)
    close(P, (r64[0], r32[0]), what="tile two-branch P")
    close(Q, (r64[1], r32[1]), what="tile two-branch Q")
    close(xg.grad, (r64[2], r32[2]), what="tile two-branch dx")
    close(P0, (r64[0], r32[0]), what="plan two-branch P")
    close(Q0, (r64[1], r32[1]), what="plan two-branch Q")
    has = torch.zeros(n).index_add_(0, dst, sel[src].view(-1)) > 0
    assert float(Q.cpu()[~has].abs().max()) == 0.0 if bool((~has).any()) else True
    assert torch.equal(Q == 0, Q0 == 0)

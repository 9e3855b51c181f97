"""The engine's SparseAdj mirror against the oracle's restatement of the reference class, method by method
(sparse_adj.py:16-151, sparse_ops.py, gcn_norm_adj TfgIDLayer.py:528-566)."""
import pytest
import torch

from _tol import both, close, close_all
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


def col(t):
    """per-entry comparison of a vector of independent quantities (edge weights, degrees): each its own scale"""
    return t.detach().reshape(-1, 1)


def test_sparse_adj_methods(dev):
    from graphgym_amd.sparse_adj import SparseAdj, diag_sparse_matmul, gcn_norm_adj, sparse_diag_matmul
    g = torch.Generator().manual_seed(0)
    n, E, d = 300, 4000, 24
    ei = torch.randint(0, n, (2, E), generator=g)
    w = torch.rand(E, generator=g) + 0.1
    h = torch.randn(n, d, generator=g)
    dg = torch.rand(n, generator=g) + 0.5
    ref = R.SparseAdj(ei, w, [n, n])
    adj = SparseAdj(ei.to(dev), w.to(dev), [n, n])
    assert adj.shape == [n, n] and torch.equal(adj.row.cpu(), ref.row) and torch.equal(adj.col.cpu(), ref.col)
    A = lambda c: R.SparseAdj(ei, c(w), [n, n])        # the oracle's class in the evaluation's dtype
    close(adj @ h.to(dev), both(lambda c: A(c) @ c(h)), what="matmul")
    close(col(adj.reduce_sum(axis=-1)), both(lambda c: col(A(c).reduce_sum(axis=-1))), what="reduce_sum rows")
    close(col(adj.reduce_sum(axis=0)), both(lambda c: col(A(c).reduce_sum(axis=0))), what="reduce_sum columns")
    close(col(adj.matmul_diag(dg.to(dev)).edge_weight), both(lambda c: col(A(c).matmul_diag(c(dg)).edge_weight)),
          what="matmul_diag")
    close(col(adj.rmatmul_diag(dg.to(dev)).edge_weight), both(lambda c: col(A(c).rmatmul_diag(c(dg)).edge_weight)),
          what="rmatmul_diag")
    close(sparse_diag_matmul(diag_sparse_matmul(dg.to(dev), adj), dg.to(dev)) @ h.to(dev),
          both(lambda c: R.sparse_diag_matmul(R.diag_sparse_matmul(c(dg), A(c)), c(dg)) @ c(h)), what="diag A diag @ h")
    close(adj.transpose() @ h.to(dev), both(lambda c: A(c).transpose() @ c(h)), what="transpose matmul")
    close(adj.rmatmul(h.t().contiguous().to(dev)), both(lambda c: (A(c).transpose() @ c(h)).t()), what="rmatmul")
    close(col(adj.softmax(axis=-1).edge_weight), both(lambda c: col(A(c).softmax(axis=-1).edge_weight)), what="softmax")
    a2, r2 = adj.add_self_loop(2.0), ref.add_self_loop(2.0)
    assert a2.edge_index.shape == r2.edge_index.shape
    close(a2 @ h.to(dev), both(lambda c: A(c).add_self_loop(2.0) @ c(h)), what="add_self_loop matmul")
    for renorm in (True, False):
        na = gcn_norm_adj(adj, renorm=renorm)
        close(col(na.edge_weight), both(lambda c: col(R.gcn_norm_adj(A(c), renorm=renorm).edge_weight)),
              what=f"gcn_norm_adj weights renorm={renorm}")
        close(na @ h.to(dev), both(lambda c: R.gcn_norm_adj(A(c), renorm=renorm) @ c(h)), what=f"gcn_norm_adj matmul {renorm}")
    with pytest.raises(Exception, match="Invalid axis"):
        adj.reduce_sum(axis=3)
    # default weights and shape inference (sparse_adj.py:31-46)
    a3 = SparseAdj(ei.to(dev))
    assert a3.shape == [int(ei.max()) + 1] * 2 and float(a3.edge_weight.min()) == 1.0
    # gradient of A @ h w.r.t. h is A' @ g
    hg = h.to(dev).requires_grad_(True)
    (adj @ hg).sum().backward()
    close(hg.grad, both(lambda c: A(c).transpose() @ c(torch.ones(n, d))), what="d(A h)/dh")


def test_sparse_adj_is_differentiable_in_edge_weight(dev):
    """the reference class is differentiable in edge_weight (gat_id: SparseAdj(att_score).softmax().dropout() @ V,
    TfgIDLayer.py:340-355): d/d edge_weight and d/d h of softmax(A) @ h, of the diagonal scalings and of the degree,
    against torch autograd through the oracle's restatement"""
    from graphgym_amd.sparse_adj import SparseAdj, diag_sparse_matmul, sparse_diag_matmul
    g = torch.Generator().manual_seed(3)
    n, E, d = 200, 3000, 16
    ei = torch.randint(0, n, (2, E), generator=g)
    w0 = torch.randn(E, generator=g)
    h0 = torch.randn(n, d, generator=g)
    dg0 = torch.rand(n, generator=g) + 0.5
    cot = torch.randn(n, d, generator=g)

    def run(adj_cls, w, h, dg, to):
        adj = adj_cls(to(ei), w, [n, n])
        sm = adj.softmax(axis=-1)
        sdm = (R.sparse_diag_matmul if adj_cls is R.SparseAdj else sparse_diag_matmul)
        dsm = (R.diag_sparse_matmul if adj_cls is R.SparseAdj else diag_sparse_matmul)
        scaled = sdm(dsm(dg, sm), dg)
        out = scaled @ h + (adj @ h) * 0.5
        deg = adj.reduce_sum(axis=-1)
        return out, deg

    def ref_fn(c):
        wr, hr, dr = (c(t).clone().requires_grad_(True) for t in (w0, h0, dg0))
        out_r, deg_r = run(R.SparseAdj, wr, hr, dr, lambda t: t)
        ((out_r * c(cot)).sum() + (deg_r * deg_r).sum()).backward()
        return out_r.detach(), deg_r.detach(), wr.grad, hr.grad, dr.grad
    r64, r32 = both(ref_fn)
    wg = w0.to(dev).requires_grad_(True)
    hg = h0.to(dev).requires_grad_(True)
    dgg = dg0.to(dev).requires_grad_(True)
    out_g, deg_g = run(SparseAdj, wg, hg, dgg, lambda t: t.to(dev))
    ((out_g * cot.to(dev)).sum() + (deg_g * deg_g).sum()).backward()
    close(out_g, (r64[0], r32[0]), what="softmax / scalings forward")
    close_all(deg_g, (r64[1], r32[1]), what="degree")             # signed weights: sums cancel, one scale
    close_all(wg.grad, (r64[2], r32[2]), what="d edge_weight")     # per-entry gradients through a softmax: one scale
    close(hg.grad, (r64[3], r32[3]), what="d h")
    close_all(dgg.grad, (r64[4], r32[4]), what="d diagonal")
    assert float(wg.grad.abs().max()) > 0

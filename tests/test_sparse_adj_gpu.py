"""The engine's SparseAdj mirror against the oracle's restatement of the reference class, method by method
(sparse_adj.py:16-151, sparse_ops.py, gcn_norm_adj TfgIDLayer.py:528-566)."""
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


def close(a, r, tol=1e-5):
    a, r = a.detach().cpu().double(), r.detach().double()
    assert a.shape == r.shape
    assert float((a - r).abs().max()) <= tol * max(1.0, float(r.abs().max()))


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
    close(adj @ h.to(dev), ref @ h)
    close(adj.reduce_sum(axis=-1), ref.reduce_sum(axis=-1))
    close(adj.reduce_sum(axis=0), ref.reduce_sum(axis=0))
    close(adj.matmul_diag(dg.to(dev)).edge_weight, ref.matmul_diag(dg).edge_weight, 1e-6)
    close(adj.rmatmul_diag(dg.to(dev)).edge_weight, ref.rmatmul_diag(dg).edge_weight, 1e-6)
    close(sparse_diag_matmul(diag_sparse_matmul(dg.to(dev), adj), dg.to(dev)) @ h.to(dev),
          R.sparse_diag_matmul(R.diag_sparse_matmul(dg, ref), dg) @ h)
    close(adj.transpose() @ h.to(dev), ref.transpose() @ h)
    close(adj.rmatmul(h.t().contiguous().to(dev)), (ref.transpose() @ h).t())
    close(adj.softmax(axis=-1).edge_weight, ref.softmax(axis=-1).edge_weight, 1e-5)
    a2, r2 = adj.add_self_loop(2.0), ref.add_self_loop(2.0)
    assert a2.edge_index.shape == r2.edge_index.shape
    close(a2 @ h.to(dev), r2 @ h)
    for renorm in (True, False):
        na, nr = gcn_norm_adj(adj, renorm=renorm), R.gcn_norm_adj(ref, renorm=renorm)
        close(na.edge_weight, nr.edge_weight, 1e-6)
        close(na @ h.to(dev), nr @ h)
    with pytest.raises(Exception, match="Invalid axis"):
        adj.reduce_sum(axis=3)
    # default weights and shape inference (sparse_adj.py:31-46)
    a3 = SparseAdj(ei.to(dev))
    assert a3.shape == [int(ei.max()) + 1] * 2 and float(a3.edge_weight.min()) == 1.0
    # gradient of A @ h w.r.t. h is A' @ g
    hg = h.to(dev).requires_grad_(True)
    (adj @ hg).sum().backward()
    close(hg.grad, ref.transpose() @ torch.ones(n, d))


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

    wr, hr, dr = w0.clone().requires_grad_(True), h0.clone().requires_grad_(True), dg0.clone().requires_grad_(True)
    out_r, deg_r = run(R.SparseAdj, wr, hr, dr, lambda t: t)
    ((out_r * cot).sum() + (deg_r * deg_r).sum()).backward()
    wg = w0.to(dev).requires_grad_(True)
    hg = h0.to(dev).requires_grad_(True)
    dgg = dg0.to(dev).requires_grad_(True)
    out_g, deg_g = run(SparseAdj, wg, hg, dgg, lambda t: t.to(dev))
    ((out_g * cot.to(dev)).sum() + (deg_g * deg_g).sum()).backward()
    close(out_g, out_r)
    close(deg_g, deg_r)
    close(wg.grad, wr.grad, 2e-5)
    close(hg.grad, hr.grad, 2e-5)
    close(dgg.grad, dr.grad, 2e-5)
    assert float(wg.grad.abs().max()) > 0

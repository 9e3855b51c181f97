"""Host-side mirror of the reference's ``SparseAdj`` (sparse_adj.py:16-151) and ``sparse_ops``
(sparse_ops.py:6-12) on the engine: same constructor, properties and methods, same COO-order edge
weights at the surface; underneath one cached destination-sorted CSR (``CSRGraph``).

    adj = SparseAdj(edge_index, edge_weight, [n, n])      # edge_index[0] = row (destination), [1] = col
    adj.add_self_loop(fill_weight) ; adj.reduce_sum(axis=-1) ; adj @ h ; adj.softmax(axis=-1)
    sparse_diag_matmul(adj, d) ; diag_sparse_matmul(d, adj) ; adj.transpose() ; adj.dropout(rate, training)
"""
import torch
import torch.nn.functional as F

from . import ops
from .graph import CSRGraph, _require_hip


class SparseAdj(object):
    def __init__(self, edge_index, edge_weight=None, shape=None, _csr=None):
        _require_hip(edge_index, "edge_index")
        self.edge_index = edge_index.to(torch.int64)
        if edge_weight is None:                                    # sparse_adj.py:31-38: ones
            edge_weight = torch.ones(self.edge_index.size(1), dtype=torch.float32, device=edge_index.device)
        self.edge_weight = edge_weight.to(torch.float32)
        if shape is None:                                          # sparse_adj.py:40-46
            n = int(self.edge_index.max().item()) + 1 if self.edge_index.numel() else 0
            shape = [n, n]
        self.shape = list(shape)
        self._csr = _csr

    @property
    def row(self):
        return self.edge_index[0]

    @property
    def col(self):
        return self.edge_index[1]

    def csr(self):
        """the engine-side graph (built once): rows = destinations, values = edge_weight (detached: kernels read the
        numbers; gradients w.r.t. edge_weight flow through the differentiable operators below)"""
        if self._csr is None:
            if self.shape[0] != self.shape[1]:
                raise ValueError("square adjacency expected")
            self._csr = CSRGraph.from_edge_index(self.edge_index, self.shape[0], self.edge_weight, dst_row=0,
                                                 validate=True)
        return self._csr

    def _needs_grad(self, *others):
        return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (self.edge_weight,) + others)

    def _coo_pos(self, g):
        """position in the CSR of every COO entry (inverse of g.eid), cached on the pattern"""
        owner = getattr(g, "_pattern_of", None) or g
        pos = owner.__dict__.get("_coo_pos")
        if pos is None:
            pos = torch.empty(g.nnz, dtype=torch.int64, device=g.device)
            pos[g.eid.long()] = torch.arange(g.nnz, device=g.device)
            owner.__dict__["_coo_pos"] = pos
        return pos

    def _with_weights(self, g, w_csr):
        """a SparseAdj over the same pattern whose weights are w_csr (CSR order, may carry a gradient graph)"""
        gv = g.with_values(w_csr.detach().contiguous())
        gv.eid = g.eid
        return SparseAdj(self.edge_index, w_csr[self._coo_pos(g)], self.shape, _csr=gv)

    def add_self_loop(self, fill_weight=1.0):                      # sparse_adj.py:58-63
        n = self.shape[0]
        diag = torch.arange(n, device=self.edge_index.device)
        ei = torch.cat([self.edge_index, torch.stack([diag, diag])], dim=1)
        ew = torch.cat([self.edge_weight, torch.full((n,), float(fill_weight), device=ei.device)])
        return SparseAdj(ei, ew, self.shape)

    def reduce_sum(self, axis=-1, keepdims=False):                 # sparse_adj.py:65-85
        if axis in (-1, 1):
            which = 0
        elif axis in (0, -2):
            which = 1
        else:
            raise Exception("Invalid axis value: {}, axis shoud be -1, -2, 0, or 1".format(axis))
        if self._needs_grad():       # differentiable in edge_weight: unsorted_segment_sum as torch autograd sees it
            out = torch.zeros(self.shape[which], dtype=torch.float32, device=self.edge_weight.device)
            out = out.index_add(0, self.edge_index[which], self.edge_weight)
        else:
            out = self.csr().degree("row" if which == 0 else "col")
        return out.unsqueeze(axis) if keepdims else out

    def matmul(self, h):                                           # sparse_adj.py:91-97
        g = self.csr()
        if self._needs_grad():       # gradient to the weights too: gat_id's softmax(scores) @ V (TfgIDLayer.py:340-355)
            return ops.spmm_edge_values(g, self.edge_weight[g.eid.long()].view(-1, 1), h, 1)
        return ops.spmm(g, h, "sum")

    def __matmul__(self, h):
        return self.matmul(h)

    def rmatmul(self, h):                                          # sparse_adj.py:100-107: (A' h')'
        return (self.transpose() @ h.t().contiguous()).t()

    def matmul_diag(self, diagonal):                               # sparse_adj.py:110-113
        if self._needs_grad(diagonal):
            return SparseAdj(self.edge_index, self.edge_weight * diagonal[self.col], self.shape)
        g = self.csr()
        return self._with_weights(g, g.scaled(col_scale=diagonal).val)

    def rmatmul_diag(self, diagonal):                              # sparse_adj.py:116-119
        if self._needs_grad(diagonal):
            return SparseAdj(self.edge_index, diagonal[self.row] * self.edge_weight, self.shape)
        g = self.csr()
        return self._with_weights(g, g.scaled(row_scale=diagonal).val)

    def transpose(self):                                           # sparse_adj.py:124-127
        return SparseAdj(torch.stack([self.col, self.row]), self.edge_weight, self.shape)

    def dropout(self, drop_rate, training=False):                  # sparse_adj.py:129-134
        w = F.dropout(self.edge_weight, drop_rate) if training and drop_rate > 0.0 else self.edge_weight
        return SparseAdj(self.edge_index, w, self.shape, _csr=self._csr if w is self.edge_weight else None)

    def softmax(self, axis=-1):                                    # sparse_adj.py:136-151
        if axis in (-1, 1):
            g = self.csr()
            w_csr = self.edge_weight[g.eid.long()] if self._needs_grad() else g.val
            return self._with_weights(g, ops.edge_softmax(g, w_csr.view(-1, 1)).view(-1))
        if axis in (0, -2):
            return self.transpose().softmax(-1).transpose()
        raise Exception("Invalid axis value: {}, axis shoud be -1, -2, 0, or 1".format(axis))

    def __str__(self):
        return "SparseAdj: \\nedge_index => \\n{}\\nedge_weight => {}\\nshape => {}".format(
            self.edge_index, self.edge_weight, self.shape)

    __repr__ = __str__


def sparse_diag_matmul(sparse_adj, diagonal):                      # sparse_ops.py:6-7
    return sparse_adj.matmul_diag(diagonal)


def diag_sparse_matmul(diagonal, sparse_adj):                      # sparse_ops.py:11-12
    return sparse_adj.rmatmul_diag(diagonal)


def gcn_norm_adj(sparse_adj, renorm=True, improved=False, cache=None):
    """TfgIDLayer.py:528-566 on the mirror class (the layers use the fused mp_gcn_norm_edges instead)"""
    if cache is not None:
        key = "gcn_normed_edge_{}_{}".format(renorm, improved)
        if cache.get(key) is not None:
            return cache[key]
    fill_weight = 2.0 if improved else 1.0
    if renorm:
        sparse_adj = sparse_adj.add_self_loop(fill_weight=fill_weight)
    deg = sparse_adj.reduce_sum(axis=-1)
    deg_inv_sqrt = torch.pow(deg, -0.5)
    deg_inv_sqrt = torch.where(torch.isinf(deg_inv_sqrt) | torch.isnan(deg_inv_sqrt),
                               torch.zeros_like(deg_inv_sqrt), deg_inv_sqrt)
    normed = sparse_diag_matmul(diag_sparse_matmul(deg_inv_sqrt, sparse_adj), deg_inv_sqrt)
    if not renorm:
        normed = normed.add_self_loop(fill_weight=fill_weight)
    if cache is not None:
        cache[key] = normed
    return normed

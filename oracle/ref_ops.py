"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by graphgym_amd/.

Restatement, op for op, of the reference's sparse operators in plain torch (CPU):
the same gather -> scale -> scatter sequence the reference's CPU path runs, with
the [E, d] message tensor materialised exactly as it does.

PARITY UNPINNED: the reference ships no tests, golden tensors or fixtures for this
path, and none of its hot-path modules can be imported here (tensorflow,
tf_geometric, torch_geometric, torch_scatter are absent).  Functions marked [3P]
restate third-party semantics from those libraries' published behaviour
(SURVEY.md App. A); everything else follows the cited reference lines.  The
restatement is cross-checked against two independent formulations
(oracle/spmm_ref.c and scipy.sparse) and hand-computed known answers in tests/.

TF flavour  : sparse_adj.py, sparse_ops.py, TfgIDLayer.py:528-566
PyG flavour : idconv.py:44-60,132-148 and the torch_geometric utilities it calls
"""
import torch


# --------------------------------------------------------------------------- #
# TF flavour: SparseAdj (sparse_adj.py:16-151).  edge_index[0] = row = dest.   #
# --------------------------------------------------------------------------- #
class SparseAdj:
    def __init__(self, edge_index, edge_weight=None, shape=None):
        # sparse_adj.py:18-48
        self.edge_index = edge_index.to(torch.int64)
        if edge_weight is None:
            edge_weight = torch.ones(self.edge_index.size(1), dtype=torch.get_default_dtype())
        # float32 as in the reference (sparse_adj.py:31-36 casts to tf.float32); the tests' float64 evaluation of
        # the same formulas runs under torch.set_default_dtype(torch.float64)
        self.edge_weight = edge_weight.to(torch.get_default_dtype())
        if shape is None:
            n = int(self.edge_index.max()) + 1 if self.edge_index.numel() else 0
            shape = [n, n]
        self.shape = list(shape)

    @property
    def row(self):
        return self.edge_index[0]

    @property
    def col(self):
        return self.edge_index[1]

    def add_self_loop(self, fill_weight=1.0):
        # sparse_adj.py:58-63 -> tfg add_self_loop_edge [3P]: append N diagonal entries, keep the rest
        n = self.shape[0]
        diag = torch.arange(n, dtype=torch.int64)
        ei = torch.cat([self.edge_index, torch.stack([diag, diag])], dim=1)
        ew = torch.cat([self.edge_weight, torch.full((n,), float(fill_weight))])
        return SparseAdj(ei, ew, self.shape)

    def _reduce_index(self, axis):
        # sparse_adj.py:65-76
        if axis in (-1, 1):
            return 0
        if axis in (0, -2):
            return 1
        raise Exception("Invalid axis value: {}, axis shoud be -1, -2, 0, or 1".format(axis))

    def reduce_sum(self, axis=-1):
        # sparse_adj.py:84-85: unsorted_segment_sum(edge_weight, index, num)
        ra = self._reduce_index(axis)
        out = torch.zeros(self.shape[ra], dtype=self.edge_weight.dtype)
        return out.index_add_(0, self.edge_index[ra], self.edge_weight)

    def matmul(self, h):
        # sparse_adj.py:91-97: gather, scale, unsorted_segment_sum
        repeated_h = h[self.col]
        repeated_h = repeated_h * self.edge_weight[:, None]
        out = torch.zeros((self.shape[0], h.size(1)), dtype=h.dtype)
        return out.index_add_(0, self.row, repeated_h)

    __matmul__ = matmul

    def matmul_diag(self, diagonal):
        # sparse_adj.py:110-113
        return SparseAdj(self.edge_index, self.edge_weight * diagonal[self.col], self.shape)

    def rmatmul_diag(self, diagonal):
        # sparse_adj.py:116-119
        return SparseAdj(self.edge_index, diagonal[self.row] * self.edge_weight, self.shape)

    def transpose(self):
        # sparse_adj.py:124-127
        return SparseAdj(torch.stack([self.col, self.row]), self.edge_weight, self.shape)

    def softmax(self, axis=-1):
        # sparse_adj.py:136-151 -> tfg segment_softmax [3P]
        ra = self._reduce_index(axis)
        w = segment_softmax(self.edge_weight, self.edge_index[ra], self.shape[ra])
        return SparseAdj(self.edge_index, w, self.shape)


def sparse_diag_matmul(sparse_adj, diagonal):   # sparse_ops.py:6-7
    return sparse_adj.matmul_diag(diagonal)


def diag_sparse_matmul(diagonal, sparse_adj):   # sparse_ops.py:11-12
    return sparse_adj.rmatmul_diag(diagonal)


def segment_max(data, segment_ids, num_segments):
    out = torch.full((num_segments,) + tuple(data.shape[1:]), float("-inf"), dtype=data.dtype)
    idx = segment_ids.view(-1, *([1] * (data.dim() - 1))).expand_as(data)
    return out.scatter_reduce(0, idx, data, "amax", include_self=True)


def segment_softmax(data, segment_ids, num_segments):
    """tf_geometric.nn.kernel.segment.segment_softmax [3P]: exp(x - max_seg) / (sum_seg + 1e-8)"""
    mx = segment_max(data, segment_ids, num_segments)[segment_ids]
    ex = torch.exp(data - mx)
    den = torch.zeros((num_segments,) + tuple(data.shape[1:]), dtype=data.dtype).index_add_(0, segment_ids, ex)
    return ex / (den + 1e-8)[segment_ids]


def gcn_norm_adj(sparse_adj, renorm=True, improved=False):
    """TfgIDLayer.py:528-566 (cache handling omitted: the reference always passes cache=None)"""
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
    return normed


def mean_reducer(neighbor_msg, node_index, num_nodes):
    """tfg mean_reducer [3P] = unsorted_segment_mean; empty segments -> 0"""
    s = torch.zeros((num_nodes, neighbor_msg.size(1)), dtype=neighbor_msg.dtype).index_add_(0, node_index, neighbor_msg)
    c = torch.zeros(num_nodes, dtype=neighbor_msg.dtype).index_add_(0, node_index, torch.ones(node_index.numel()))
    return s / c.clamp(min=1)[:, None]


# --------------------------------------------------------------------------- #
# PyG flavour [3P]: edge_index[0] = source j, edge_index[1] = destination i    #
# --------------------------------------------------------------------------- #
def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask], (None if edge_attr is None else edge_attr[mask])


def add_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    loop = torch.arange(num_nodes, dtype=torch.int64)
    ei = torch.cat([edge_index, torch.stack([loop, loop])], dim=1)
    if edge_weight is not None:
        edge_weight = torch.cat([edge_weight, torch.full((num_nodes,), float(fill_value), dtype=edge_weight.dtype)])
    return ei, edge_weight


def add_remaining_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    """every node ends with exactly one loop; an existing loop keeps its weight"""
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop = torch.arange(num_nodes, dtype=torch.int64)
    if edge_weight is not None:
        inv = ~mask
        loop_weight = torch.full((num_nodes,), float(fill_value), dtype=edge_weight.dtype)
        loop_weight[row[inv]] = edge_weight[inv]
        edge_weight = torch.cat([edge_weight[mask], loop_weight])
    ei = torch.cat([edge_index[:, mask], torch.stack([loop, loop])], dim=1)
    return ei, edge_weight


def scatter(src, index, dim_size, reduce="add"):
    """torch_scatter.scatter along dim 0 [3P]; empty segments -> 0 for every reduce"""
    shape = (dim_size,) + tuple(src.shape[1:])
    if reduce in ("add", "sum"):
        return torch.zeros(shape, dtype=src.dtype).index_add_(0, index, src)
    if reduce == "mean":
        s = torch.zeros(shape, dtype=src.dtype).index_add_(0, index, src)
        c = torch.zeros(dim_size, dtype=src.dtype).index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype))
        return s / c.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))
    if reduce == "max":
        out = segment_max(src, index, dim_size)
        return torch.where(torch.isinf(out) & (out < 0), torch.zeros_like(out), out)
    raise ValueError(reduce)


def scatter_add(src, index, dim_size):
    return scatter(src, index, dim_size, "add")


def softmax(src, index, num_nodes):
    """torch_geometric.utils.softmax [3P]"""
    mx = segment_max(src, index, num_nodes)[index]
    out = (src - mx).exp()
    den = torch.zeros((num_nodes,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, out)
    return out / (den[index] + 1e-16)


def propagate(edge_index, x, aggr, norm=None, edge_feature=None):
    """MessagePassing.propagate [3P] for message = norm * x_j (idconv.py:91-92,179-180,238-239):
    gather x_j = x[edge_index[0]], scale, scatter over edge_index[1]; with edge_feature the message is
    norm * (x_j + edge_feature) (generalconv.py:99-106)"""
    x_j = x[edge_index[0]]
    if edge_feature is not None:
        x_j = x_j + edge_feature
    msg = norm.view(-1, 1) * x_j if norm is not None else x_j
    return scatter(msg, edge_index[1], x.size(0), aggr)


def pyg_gcn_norm(edge_index, num_nodes, edge_weight=None, improved=False):
    """GCNIDConvLayer.norm / GeneralIDConvLayer.norm / GeneralConvLayer.norm
    (idconv.py:44-60,132-148; generalconv.py:44-60): degree scattered on edge_index[0]"""
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=torch.get_default_dtype())
    fill_value = 1.0 if not improved else 2.0
    edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill_value, num_nodes)
    row, col = edge_index[0], edge_index[1]
    deg = scatter_add(edge_weight, row, num_nodes)
    deg_inv_sqrt = deg.pow(-0.5)
    deg_inv_sqrt[deg_inv_sqrt == float("inf")] = 0
    return edge_index, deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]


# --------------------------------------------------------------------------- #
# The aggregation itself in the COO form both flavours reduce to              #
# --------------------------------------------------------------------------- #
def coo_aggregate(dst, src, w, x, num_nodes, reduce="sum"):
    """out[i] = reduce_{e: dst[e] = i} w[e] * x[src[e]]  (w None = ones).
    'sum' == SparseAdj.matmul; 'mean'/'max' == torch_scatter semantics."""
    msg = x[src]
    if w is not None:
        msg = msg * w[:, None]
    return scatter(msg, dst, num_nodes, {"sum": "add"}.get(reduce, reduce))


def coo_aggregate_argmax(dst, src, w, x, num_nodes):
    """per (row, column) position e of the winning entry under reduce='max'; ties go to the
    lowest e (torch_scatter's CPU loop updates only on strictly greater [3P])."""
    msg = x[src] if w is None else x[src] * w[:, None]
    E, d = msg.shape
    best = torch.full((num_nodes, d), float("-inf"))
    arg = torch.full((num_nodes, d), -1, dtype=torch.int64)
    for e in range(E):
        i = int(dst[e])
        better = msg[e] > best[i]
        best[i] = torch.where(better, msg[e], best[i])
        arg[i] = torch.where(better, torch.full_like(arg[i], e), arg[i])
    return arg


def coo_aggregate_sum_chunked(dst, src, w, x, out, chunk=4_000_000, max_seconds=None):
    """SparseAdj.matmul (sparse_adj.py:91-97) evaluated in edge chunks so the [E, d] message tensor
    the reference materialises stays bounded (it would be 102 GB at the headline size).  Accumulates
    into `out`; returns the number of edges processed (stops early after max_seconds)."""
    import time
    t0 = time.perf_counter()
    done = 0
    E = dst.numel()
    for s in range(0, E, chunk):
        e = min(s + chunk, E)
        msg = x[src[s:e]]
        if w is not None:
            msg = msg * w[s:e, None]
        out.index_add_(0, dst[s:e], msg)
        done = e
        if max_seconds is not None and time.perf_counter() - t0 >= max_seconds:
            break
    return done

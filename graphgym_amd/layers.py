"""Host-side mirror of the reference's GNN layer interfaces, running on the HIP engine.

PyG family (ctor / forward / parameter names of the classes they replace, so
``state_dict``s interchange):
    GCNIDConvLayer, GeneralIDConvLayer, SAGEIDConvLayer, GATIDConvLayer, GINIDConvLayer
                                         graphgym/contrib/layer/idconv.py:16-382
    GeneralConvLayer                     graphgym/contrib/layer/generalconv.py:12-114
    GCNConvLayer, SAGEConvLayer, GATConvLayer, GINConvLayer
                                         torch_geometric.nn.{GCN,SAGE,GAT,GIN}Conv [3P] as wrapped
                                         by graphgym/models/layer.py:135-174
TF family (keras-style ``layer([x, edge_index, id_index(, edge_weight)])``):
    IDGCN, IDSAGE, IDGIN, IDGAT          TfgIDLayer.py:15-525
    GCN, MeanGraphSage, GIN, GAT         tf_geometric.layers.* [3P] as used by main_zd.py:28-243

Every aggregation goes through graphgym_amd.ops -> libmpengine.so.  The CSR (self loops,
normalisation, transpose) is built once per batch and cached on the batch object; the
reference rebuilds all of it inside every layer call.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter

from . import ops
from .config import cfg
from .graph import CSRGraph


# ---- parameter init: torch_geometric.nn.inits.glorot / zeros [3P]; keras glorot_uniform ----
def glorot(tensor):
    if tensor is not None:
        stdv = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        tensor.data.uniform_(-stdv, stdv)


def zeros(tensor):
    if tensor is not None:
        tensor.data.fill_(0)


# ---- per-batch graph cache ---------------------------------------------------------------
_LOOP_FLAGS = {
    "none": dict(),
    "add": dict(add_self_loops=True),                                   # tfg add_self_loop_edge
    "remaining": dict(remove_self_loops=True, add_self_loops=True, keep_loop_weight=True),
    "remove": dict(remove_self_loops=True),
    "remove_add": dict(remove_self_loops=True, add_self_loops=True),
}


def get_graph(holder, edge_index, num_nodes, *, dst_row=1, loops="none", norm=None, fill=1.0,
              edge_weight=None):
    """CSR for (edge_index, self-loop policy, normalisation), cached on ``holder`` (the batch).

    loops: none | add | remaining | remove | remove_add ; norm: None | 'row' | 'col'."""
    key = (dst_row, loops, norm, float(fill))
    cache = None
    if holder is not None and edge_weight is None:
        stamp = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, int(num_nodes))
        cache = getattr(holder, "_mp_graph_cache", None)
        if cache is None or cache.get("stamp") != stamp:
            cache = {"stamp": stamp, "edge_index": edge_index}     # holding the tensor keeps its address unique
            try:
                setattr(holder, "_mp_graph_cache", cache)
            except Exception:
                cache = None
        if cache is not None and key in cache:
            return cache[key]
    base_key = (dst_row, loops, None, float(fill))
    if cache is not None and norm is not None and base_key in cache:
        g = cache[base_key]                                   # (e.g. seeded with the batch: seed_graph_cache)
    else:
        g = CSRGraph.from_edge_index(edge_index, num_nodes, edge_weight, dst_row=dst_row, fill=fill,
                                     **_LOOP_FLAGS[loops])
    if norm is not None:
        g = g.gcn_norm(norm if not g.symmetric else "row")    # (symmetric: degrees by row and by column coincide)
    if cache is not None:
        cache[key] = g
    return g


def seed_graph_cache(holder, edge_index, num_nodes, g, loops):
    """Put a CSRGraph that was built together with the batch (ego.ego_batch(csr=...): the expansion writes the batch's
    CSR directly) where get_graph looks for it.  `loops`: "none" (the entries of edge_index) or "add" (+ one self entry
    per node).  The graph must be symmetric and loop-free apart from the added entries, so every self-loop policy that
    coincides on such an input and both edge_index conventions map to it."""
    stamp = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, int(num_nodes))
    cache = {"stamp": stamp, "edge_index": edge_index}
    same = ("none", "remove") if loops == "none" else ("add", "remaining", "remove_add")
    for dst_row in (0, 1):
        for lp in same:
            cache[(dst_row, lp, None, 1.0)] = g
    setattr(holder, "_mp_graph_cache", cache)
    return cache


def _is_relu(fn):
    return fn in ("relu", torch.relu, F.relu) or isinstance(fn, nn.ReLU)


def _apply_act(h, fn):
    if fn is None:
        return h
    if _is_relu(fn):
        return torch.relu(h)
    if isinstance(fn, str):
        return getattr(F, fn)(h)
    return fn(h)


def _id_branch(h, x, id_index, weight_id):
    """h[id] += x[id] @ W_id  (K10)"""
    return ops.index_add_rows(h, id_index, torch.matmul(ops.gather_rows(x, id_index), weight_id))


def _pick_order(mode, dim_in, dim_out):
    """'auto': gather at the narrower width; at equal widths the one-kernel aggregate -> transform
    (ops.agg_dense) hides the MFMA work behind the gathers, so aggregate first when it applies"""
    if mode == "auto":
        if dim_in < dim_out or (dim_in == dim_out and dim_in in ops.FUSED_WIDTHS):
            return "aggregate_first"
        return "transform_first"
    return mode


def _fused_mlp_head(mlp, g, x, self_scale):
    """(1 + eps) x + sum_j x_j followed by an MLP that opens with Linear [-> ReLU]: the combine step and the
    first Linear (+ ReLU) run as one kernel; returns None when the MLP has another shape"""
    if not isinstance(mlp, nn.Sequential) or len(mlp) == 0 or not isinstance(mlp[0], nn.Linear):
        return None
    lin = mlp[0]
    if not ops.agg_dense_supported(g, x, lin.weight.t()) or x.dtype != torch.float32:
        return None
    own_relu = bool(getattr(lin, "relu", False))                  # graphgym_amd.nn.Linear(relu=True)
    next_relu = not own_relu and len(mlp) > 1 and type(mlp[1]) is nn.ReLU
    h = ops.agg_dense(g, x, lin.weight.t(), bias=lin.bias, relu=own_relu or next_relu, self_scale=self_scale)
    for m in list(mlp)[2 if next_relu else 1:]:
        h = m(h)
    return h


def _id_mlp_rows(out, g, x, id_index, mlp_id, self_scale):
    """out[id] += MLP_id(h[id]) (TfgIDLayer.py:160-165, idconv.py:373-375) when h = (1 + eps) x + sum_j x_j was never
    materialised (the one-kernel head): the identity nodes' rows of h are re-aggregated over their own in-edges only
    (an [n_id, N] operator)"""
    sub = g.select_rows(id_index)
    h_id = ops.spmm(sub, x, "sum") + self_scale * ops.gather_rows(x, id_index)
    return ops.index_add_rows(out, id_index, mlp_id(h_id))


# =========================================================================================
# PyG family
# =========================================================================================
class _CachedEdgesMixin:
    """the `cached=True` contract of the reference layers (idconv.py:69-87,157-175)"""

    def _check_cached(self, edge_index):
        if self.cached and self.cached_result is not None:
            if edge_index.size(1) != self.cached_num_edges:
                raise RuntimeError(
                    'Cached {} number of edges, but found {}. Please '
                    'disable the caching behavior of this layer by removing '
                    'the `cached=True` argument in its constructor.'.format(
                        self.cached_num_edges, edge_index.size(1)))

    def _graph(self, holder, edge_index, num_nodes, edge_weight, **kw):
        self._check_cached(edge_index)
        if self.cached and self.cached_result is not None:
            return self.cached_result
        g = get_graph(holder, edge_index, num_nodes, edge_weight=edge_weight, **kw)
        self.cached_num_edges = edge_index.size(1)
        if self.cached:
            self.cached_result = g
        return g


class GCNIDConvLayer(nn.Module, _CachedEdgesMixin):
    """idconv.py:104-189.  order: 'transform_first' follows the reference's operation order
    (x W, id rows += x W_id, then aggregate); 'aggregate_first' runs the two-branch aggregation
    kernel on x and the two GEMMs after it; 'auto' aggregates on the narrower side."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, bias=True,
                 normalize=True, order="auto", **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached, self.normalize, self.order = improved, cached, normalize, order
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        self.weight_id = Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        glorot(self.weight_id)
        zeros(self.bias)
        self.cached_result = None
        self.cached_num_edges = None

    _agg = "add"

    def graph_for(self, x, edge_index, edge_weight=None, holder=None):
        """the normalised operator this layer aggregates with (cached on the batch)"""
        if self.normalize:
            return self._graph(holder, edge_index, x.size(0), edge_weight, loops="remaining", norm="col",
                               fill=2.0 if self.improved else 1.0)
        return self._graph(holder, edge_index, x.size(0), edge_weight, loops="none")

    def forward(self, x, edge_index, id, edge_weight=None, holder=None):
        g = self.graph_for(x, edge_index, edge_weight, holder)
        order = _pick_order(self.order, self.in_channels, self.out_channels)
        if order == "aggregate_first" and self._agg in ("add", "sum"):
            out = ops.agg_dense_id(g, x, self.weight, self.weight_id, id, self.bias)   # one launch + the identity fix-up
            if out is not None:
                return out
            P, Q = ops.idgnn_aggregate(g, id, x)
            return ops.dense_fused(P, self.weight, Q, self.weight_id, self.bias)   # P W + Q W_id + b, one kernel
        h = _id_branch(ops.dense_fused(x, self.weight), x, id, self.weight_id)
        return ops.spmm(g, h, self._agg, bias=self.bias)

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)


class GeneralIDConvLayer(GCNIDConvLayer):
    """idconv.py:16-101: aggregation and normalisation come from cfg.gnn.agg / cfg.gnn.normalize_adj"""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, bias=True, **kwargs):
        super().__init__(in_channels, out_channels, improved=improved, cached=cached, bias=bias,
                         normalize=cfg.gnn.normalize_adj, **kwargs)
        self._agg = cfg.gnn.agg


class GeneralConvLayer(nn.Module, _CachedEdgesMixin):
    """generalconv.py:12-114"""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, bias=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached = improved, cached
        self.normalize = cfg.gnn.normalize_adj
        self.agg = cfg.gnn.agg
        self.self_msg = cfg.gnn.self_msg
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        if self.self_msg == 'concat':
            self.weight_self = Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        if self.self_msg == 'concat':
            glorot(self.weight_self)
        zeros(self.bias)
        self.cached_result = None
        self.cached_num_edges = None

    def forward(self, x, edge_index, edge_weight=None, edge_feature=None, holder=None):
        if self.self_msg not in ('none', 'add', 'concat'):
            raise ValueError('self_msg {} not defined'.format(self.self_msg))
        if self.self_msg == 'concat':
            x_self = ops.dense_fused(x, self.weight_self)
        h = ops.dense_fused(x, self.weight)
        if self.normalize:
            g = self._graph(holder, edge_index, x.size(0), edge_weight, loops="remaining", norm="col",
                            fill=2.0 if self.improved else 1.0)
        else:
            g = self._graph(holder, edge_index, x.size(0), edge_weight, loops="none")
        if edge_feature is None:
            x_msg = ops.spmm(g, h, self.agg, bias=self.bias)
        else:
            # message = norm * (x_j + edge_feature) (generalconv.py:99-106): per-entry messages [nnz, d_out], reduced by
            # destination on the aggregation kernel through the one-column-per-entry operator
            if edge_feature.size(0) != g.nnz or bool((g.eid < 0).any()):
                raise RuntimeError("edge_feature has {} rows, the operator has {} entries (self loops were added or "
                                   "removed: the reference fails here too)".format(edge_feature.size(0), g.nnz))
            msg = ops.gather_rows(h, g.col.long()) + edge_feature[g.eid.long()]
            x_msg = ops.spmm(g.edge_operator(), msg, self.agg, bias=self.bias)
        if self.self_msg == 'none':
            return x_msg
        if self.self_msg == 'add':
            return x_msg + h
        return x_msg + x_self

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)


class SAGEIDConvLayer(nn.Module):
    """idconv.py:192-263"""

    def __init__(self, in_channels, out_channels, normalize=False, concat=False, bias=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.normalize, self.concat = normalize, concat
        in_channels = 2 * in_channels if concat else in_channels
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        self.weight_id = Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        glorot(self.weight_id)
        zeros(self.bias)

    def forward(self, x, edge_index, id, edge_weight=None, size=None, res_n_id=None, holder=None):
        g = get_graph(holder, edge_index, x.size(0), loops="none" if self.concat else "remaining",
                      edge_weight=edge_weight)
        aggr_out = ops.spmm(g, x, "mean")
        if self.concat:
            aggr_out = torch.cat([x, aggr_out], dim=-1)
        out = ops.dense_fused(aggr_out, self.weight)
        if id is not None:
            out = _id_branch(out, aggr_out, id, self.weight_id)
        if self.bias is not None:
            out = out + self.bias
        if self.normalize:
            out = F.normalize(out, p=2, dim=-1)
        return out

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)


class GATIDConvLayer(nn.Module):
    """idconv.py:266-347 (additive attention, LeakyReLU, softmax over each destination's in-edges)"""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout=0,
                 bias=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.heads, self.concat = heads, concat
        self.negative_slope, self.dropout = negative_slope, dropout
        self.weight = Parameter(torch.Tensor(in_channels, heads * out_channels))
        self.weight_id = Parameter(torch.Tensor(in_channels, heads * out_channels))
        self.att = Parameter(torch.Tensor(1, heads, 2 * out_channels))
        if bias and concat:
            self.bias = Parameter(torch.Tensor(heads * out_channels))
        elif bias and not concat:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        glorot(self.weight_id)
        glorot(self.att)
        zeros(self.bias)

    def _attend(self, g, h):
        H, Cc = self.heads, self.out_channels
        hv = h.view(-1, H, Cc)
        a_i = (hv * self.att[:, :, :Cc]).sum(dim=-1)          # [N, H]  destination part
        a_j = (hv * self.att[:, :, Cc:]).sum(dim=-1)          # [N, H]  source part
        alpha = ops.gat_alpha(g, a_i, a_j, self.negative_slope)      # scores + row softmax, all heads, one launch
        alpha = F.dropout(alpha, p=self.dropout, training=self.training)
        out = ops.spmm_edge_values(g, alpha, h, H)
        if not self.concat:
            out = out.view(-1, H, Cc).mean(dim=1)
        return out + self.bias if self.bias is not None else out

    def forward(self, x, edge_index, id, size=None, holder=None):
        g = get_graph(holder, edge_index, x.size(0), loops="remove_add")
        h = ops.dense_fused(x, self.weight)
        if id is not None:
            h = _id_branch(h, x, id, self.weight_id)
        return self._attend(g, h)

    def __repr__(self):
        return '{}({}, {}, heads={})'.format(self.__class__.__name__, self.in_channels,
                                             self.out_channels, self.heads)


class GINIDConvLayer(nn.Module):
    """idconv.py:350-382"""

    def __init__(self, nn, nn_id, eps=0, train_eps=False, **kwargs):
        super().__init__()
        self.nn = nn
        self.nn_id = nn_id
        self.initial_eps = eps
        if train_eps:
            self.eps = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer('eps', torch.Tensor([eps]))
        self.train_eps = train_eps
        self._eps_host = float(eps)

    _loops = "remove"

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # a loaded state_dict may carry another eps than the constructor's: refresh the host copy HERE (the tensor in
        # the state dict is usually still on the host), so that forward never reads the device buffer back
        t = state_dict.get(prefix + 'eps')
        if t is not None and not self.train_eps:
            self._eps_host = float(t)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _eps_value(self):
        """eps of the fixed-eps form as a Python float, WITHOUT a device read (a `float(self.eps)` per forward is a host
        synchronisation per layer call and cannot be captured into a HIP graph).  The buffer `eps` stays the
        state-dict entry (idconv.py:361); the host copy follows the constructor and load_state_dict.  Code that
        rewrites the buffer in place calls `sync_eps()` afterwards."""
        return self._eps_host

    def sync_eps(self):
        self._eps_host = float(self.eps)
        return self._eps_host

    def _combine(self, g, x):
        # (1 + eps) * x + sum_j x_j ; with a constant eps the self term rides in the aggregation's epilogue
        if self.train_eps:
            return (1 + self.eps) * x + ops.spmm(g, x, "sum")
        return ops.spmm(g, x, "sum", self_scale=1.0 + self._eps_value())

    def forward(self, x, edge_index, id, holder=None):
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        g = get_graph(holder, edge_index, x.size(0), loops=self._loops)
        if not self.train_eps:
            out = _fused_mlp_head(self.nn, g, x, 1.0 + self._eps_value())
            if out is not None:
                return out if id is None else _id_mlp_rows(out, g, x, id, self.nn_id, 1.0 + self._eps_value())
        h = self._combine(g, x)
        out = self.nn(h)
        if id is not None:
            out = ops.index_add_rows(out, id, self.nn_id(ops.gather_rows(h, id)))
        return out

    def __repr__(self):
        return '{}(nn={})'.format(self.__class__.__name__, self.nn)


# ---- torch_geometric built-ins behind GraphGym's gcnconv / sageconv / gatconv / ginconv [3P] ----
class GCNConvLayer(nn.Module):
    """pyg.nn.GCNConv: x W, add remaining self loops, D^-1/2 A D^-1/2 (degree over destinations), sum, bias"""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, bias=True, order="auto",
                 **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.order = improved, order
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        zeros(self.bias)

    def graph_for(self, x, edge_index, edge_weight=None, holder=None):
        return get_graph(holder, edge_index, x.size(0), loops="remaining", norm="row",
                         fill=2.0 if self.improved else 1.0, edge_weight=edge_weight)

    def forward(self, x, edge_index, edge_weight=None, holder=None):
        g = self.graph_for(x, edge_index, edge_weight, holder)
        if _pick_order(self.order, self.in_channels, self.out_channels) == "aggregate_first":
            return ops.agg_dense(g, x, self.weight, bias=self.bias)
        return ops.spmm(g, ops.dense_fused(x, self.weight), "sum", bias=self.bias)


class SAGEConvLayer(nn.Module):
    """pyg.nn.SAGEConv: lin_l(mean_j x_j) + lin_r(x_i)"""

    def __init__(self, in_channels, out_channels, normalize=False, bias=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.normalize = in_channels, out_channels, normalize
        self.lin_l = nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x, edge_index, holder=None):
        g = get_graph(holder, edge_index, x.size(0), loops="none")
        out = self.lin_l(ops.spmm(g, x, "mean")) + self.lin_r(x)
        if self.normalize:
            out = F.normalize(out, p=2., dim=-1)
        return out


class GATConvLayer(nn.Module):
    """pyg.nn.GATConv (heads, shared lin_l, att_l on sources, att_r on destinations)"""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout=0.0,
                 bias=True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.heads, self.concat = heads, concat
        self.negative_slope, self.dropout = negative_slope, dropout
        self.lin_l = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_l = Parameter(torch.Tensor(1, heads, out_channels))
        self.att_r = Parameter(torch.Tensor(1, heads, out_channels))
        if bias and concat:
            self.bias = Parameter(torch.Tensor(heads * out_channels))
        elif bias and not concat:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.lin_l.weight)
        glorot(self.att_l)
        glorot(self.att_r)
        zeros(self.bias)

    def forward(self, x, edge_index, holder=None):
        g = get_graph(holder, edge_index, x.size(0), loops="remove_add")
        H, Cc = self.heads, self.out_channels
        h = self.lin_l(x)
        hv = h.view(-1, H, Cc)
        a_src = (hv * self.att_l).sum(dim=-1)
        a_dst = (hv * self.att_r).sum(dim=-1)
        alpha = F.dropout(ops.gat_alpha(g, a_dst, a_src, self.negative_slope), p=self.dropout, training=self.training)
        out = ops.spmm_edge_values(g, alpha, h, H)
        if not self.concat:
            out = out.view(-1, H, Cc).mean(dim=1)
        return out + self.bias if self.bias is not None else out


class GINConvLayer(GINIDConvLayer):
    """pyg.nn.GINConv: nn((1 + eps) x + sum_j x_j), self loops left in place"""
    _loops = "none"

    def __init__(self, nn, eps=0, train_eps=False, **kwargs):
        super().__init__(nn, None, eps=eps, train_eps=train_eps)

    def forward(self, x, edge_index, holder=None):
        return super().forward(x, edge_index, None, holder=holder)


# ---- GraphGym batch wrappers: (dim_in, dim_out, bias=False, **kwargs); forward(batch) -> batch ----
def _mlp2(dim_in, dim_out):
    # idconv.py:432-435 / layer.py:168-170: Linear -> ReLU -> Linear (same module layout and state dict; the
    # Linears run on the engine's transform kernel)
    from .nn import Linear as _Lin
    return nn.Sequential(_Lin(dim_in, dim_out), nn.ReLU(), _Lin(dim_out, dim_out))


class _BatchLayer(nn.Module):
    uses_id = False

    def forward(self, batch):
        if self.uses_id:
            batch.node_feature = self.model(batch.node_feature, batch.edge_index, batch.node_id_index,
                                            holder=batch)
        else:
            batch.node_feature = self.model(batch.node_feature, batch.edge_index, holder=batch)
        return batch


class GCNConv(_BatchLayer):            # layer.py:135-142
    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GCNConvLayer(dim_in, dim_out, bias=bias)


class SAGEConv(_BatchLayer):           # layer.py:145-152
    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = SAGEConvLayer(dim_in, dim_out, bias=bias)


class GATConv(_BatchLayer):            # layer.py:155-162
    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GATConvLayer(dim_in, dim_out, bias=bias)


class GINConv(_BatchLayer):            # layer.py:165-174
    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GINConvLayer(_mlp2(dim_in, dim_out))


class GeneralConv(_BatchLayer):        # layer.py:188-196
    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GeneralConvLayer(dim_in, dim_out, bias=bias)


class GeneralIDConv(_BatchLayer):      # idconv.py:385-393
    uses_id = True

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GeneralIDConvLayer(dim_in, dim_out, bias=bias)


class GCNIDConv(_BatchLayer):          # idconv.py:396-404
    uses_id = True

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GCNIDConvLayer(dim_in, dim_out, bias=bias)


class SAGEIDConv(_BatchLayer):         # idconv.py:407-415
    uses_id = True

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = SAGEIDConvLayer(dim_in, dim_out, bias=bias, concat=True)


class GATIDConv(_BatchLayer):          # idconv.py:418-426
    uses_id = True

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GATIDConvLayer(dim_in, dim_out, bias=bias)


class GINIDConv(_BatchLayer):          # idconv.py:429-441
    uses_id = True

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GINIDConvLayer(_mlp2(dim_in, dim_out), _mlp2(dim_in, dim_out))


# =========================================================================================
# TF family — keras-style layers of TfgIDLayer.py / tf_geometric.layers
# =========================================================================================
def _unpack(inputs, with_id):
    """[x, edge_index, id_index(, edge_weight)] or [x, edge_index(, edge_weight)] (TfgIDLayer.py:80-84)"""
    if with_id:
        if len(inputs) == 4:
            return inputs
        x, edge_index, id_index = inputs
        return x, edge_index, id_index, None
    if len(inputs) == 3:
        x, edge_index, edge_weight = inputs
        return x, edge_index, None, edge_weight
    x, edge_index = inputs
    return x, edge_index, None, None


class _KerasLike(nn.Module):
    """parameters are created by build(num_features) — at construction when in_features is given,
    else on the first call, like keras' build(input_shape)"""
    with_id = True

    def _maybe_build(self, x):
        if not self._built:
            self.build(x.size(-1))
            self.to(x.device)

    def forward(self, inputs, cache=None, training=None, mask=None, holder=None):
        return self.call(inputs, cache=cache, training=training, mask=mask, holder=holder)


class IDGCN(_KerasLike):
    """TfgIDLayer.py:391-475 (+ gcn_id :478-525, gcn_norm_adj :528-566)"""

    def __init__(self, units, activation=None, use_bias=True, renorm=True, improved=False,
                 in_features=None, order="auto", **kwargs):
        super().__init__()
        self.units, self.activation, self.use_bias = units, activation, use_bias
        self.renorm, self.improved, self.order = renorm, improved, order
        self.kernel = self.kernel_id = self.bias = None
        self._built = False
        if in_features is not None:
            self.build(in_features)

    def build(self, num_features):
        self.in_features = num_features
        self.kernel = Parameter(torch.empty(num_features, self.units))
        glorot(self.kernel)
        if self.with_id:
            self.kernel_id = Parameter(torch.empty(num_features, self.units))
            glorot(self.kernel_id)
        if self.use_bias:
            self.bias = Parameter(torch.zeros(self.units))
        self._built = True

    def _normed_graph(self, holder, edge_index, n, edge_weight):
        fill = 2.0 if self.improved else 1.0
        if self.renorm:
            return get_graph(holder, edge_index, n, dst_row=0, loops="add", norm="row", fill=fill,
                             edge_weight=edge_weight)
        # D^-1/2 A D^-1/2 without loops, then loops of weight `fill` appended un-normalised (:560-561)
        g0 = get_graph(holder, edge_index, n, dst_row=0, loops="none", norm="row", edge_weight=edge_weight)
        g1 = get_graph(holder, edge_index, n, dst_row=0, loops="add", fill=fill, edge_weight=edge_weight)
        v1 = g1.val if g1.val is not None else torch.ones(g1.nnz, device=g1.device)
        normed = g0.dinv[g1.row_ids().long()] * v1 * g0.dinv[g1.col.long()]
        return g1.with_values(torch.where(g1.eid < 0, torch.full_like(normed, fill), normed))

    def prepare(self, inputs, holder):
        """the per-batch graph work of call(), ahead of time (CSRGraph.warm): normalised operator, plans, transpose,
        identity-branch operators — cached on `holder`, where call() finds them"""
        x, edge_index, id_index, edge_weight = _unpack(inputs, self.with_id)
        g = self._normed_graph(holder, edge_index, x.size(0), edge_weight)
        g.warm(id_index if _pick_order(self.order, x.size(1), self.units) == "aggregate_first" else None)
        return g

    def call(self, inputs, cache=None, training=None, mask=None, holder=None):
        x, edge_index, id_index, edge_weight = _unpack(inputs, self.with_id)
        self._maybe_build(x)
        g = self._normed_graph(holder, edge_index, x.size(0), edge_weight)
        relu = _is_relu(self.activation)
        order = _pick_order(self.order, self.in_features, self.units)
        if order == "aggregate_first":
            if id_index is not None:
                h = ops.agg_dense_id(g, x, self.kernel, self.kernel_id, id_index, self.bias, relu=relu)
                if h is None:
                    P, Q = ops.idgnn_aggregate(g, id_index, x)
                    h = ops.dense_fused(P, self.kernel, Q, self.kernel_id, self.bias, relu=relu)
            else:
                h = ops.agg_dense(g, x, self.kernel, bias=self.bias, relu=relu)
            return h if relu else _apply_act(h, self.activation)
        h = ops.dense_fused(x, self.kernel)
        if id_index is not None:
            h = _id_branch(h, x, id_index, self.kernel_id)
        h = ops.spmm(g, h, "sum", bias=self.bias, relu=relu)      # bias + activation fused (:519-523)
        return h if relu else _apply_act(h, self.activation)


class GCN(IDGCN):
    """tf_geometric.layers.GCN [3P] — IDGCN without the identity branch (main_zd.py:33-35)"""
    with_id = False


class IDSAGE(_KerasLike):
    """TfgIDLayer.py:15-120"""

    def __init__(self, units, activation=torch.relu, use_bias=True, concat=True, normalize=False,
                 in_features=None, **kwargs):
        super().__init__()
        if concat and (units % 2 != 0):
            raise Exception("units must be a event number if concat is True")
        self.units, self.activation, self.use_bias = units, activation, use_bias
        self.concat, self.normalize = concat, normalize
        self.self_kernel = self.id_kernel = self.neighbor_kernel = self.bias = None
        self._built = False
        if in_features is not None:
            self.build(in_features)

    def build(self, num_features):
        ku = self.units // 2 if self.concat else self.units
        self.self_kernel = Parameter(torch.empty(num_features, ku))
        glorot(self.self_kernel)
        if self.with_id:
            self.id_kernel = Parameter(torch.empty(num_features, ku))
            glorot(self.id_kernel)
        self.neighbor_kernel = Parameter(torch.empty(num_features, ku))
        glorot(self.neighbor_kernel)
        if self.use_bias:
            self.bias = Parameter(torch.zeros(self.units))
        self._built = True

    def call(self, inputs, cache=None, training=None, mask=None, holder=None):
        x, edge_index, id_index, edge_weight = _unpack(inputs, self.with_id)
        self._maybe_build(x)
        g = get_graph(holder, edge_index, x.size(0), dst_row=0, loops="none", edge_weight=edge_weight)
        if id_index is None and self.concat and (self.activation is None or _is_relu(self.activation)):
            # [x W_s ‖ mean W_n] + b -> act written as one buffer by two kernel launches (no cat / bias / act passes)
            h = ops.sage_concat(g, x, self.self_kernel, self.neighbor_kernel, self.bias,
                                relu=self.activation is not None)
            return F.normalize(h, p=2, dim=-1) if self.normalize else h
        neighbor_msg = ops.dense_fused(ops.spmm(g, x, "mean"), self.neighbor_kernel)
        h = ops.dense_fused(x, self.self_kernel)
        if id_index is not None:
            h = _id_branch(h, x, id_index, self.id_kernel)
        h = torch.cat([h, neighbor_msg], dim=1) if self.concat else h + neighbor_msg
        if self.bias is not None:
            h = h + self.bias
        h = _apply_act(h, self.activation)
        if self.normalize:
            h = F.normalize(h, p=2, dim=-1)
        return h


class MeanGraphSage(IDSAGE):
    """tf_geometric.layers.MeanGraphSage [3P] (main_zd.py:131-132)"""
    with_id = False


class IDGIN(_KerasLike):
    """TfgIDLayer.py:123-167"""

    def __init__(self, mlp_model, mlpid_model=None, eps=0, train_eps=False, **kwargs):
        super().__init__()
        self.mlp_model = mlp_model
        self.mlp_id = mlpid_model
        self.train_eps = train_eps
        if train_eps:
            self.eps = Parameter(torch.zeros(()))
        else:
            self.eps = eps
        self._built = True

    def prepare(self, inputs, holder):
        """the per-batch graph work of call(), ahead of time (CSRGraph.warm), cached on `holder`"""
        x, edge_index, id_index, _ = _unpack(inputs, self.with_id)
        g = get_graph(holder, edge_index, x.size(0), dst_row=0, loops="none")
        g.warm()
        if id_index is not None:
            g.select_rows(id_index).warm()
        return g

    def call(self, inputs, cache=None, training=None, mask=None, holder=None):
        x, edge_index, id_index, _ = _unpack(inputs, self.with_id)   # edge weights ignored (:150-151)
        g = get_graph(holder, edge_index, x.size(0), dst_row=0, loops="none")
        if not self.train_eps:
            out = _fused_mlp_head(self.mlp_model, g, x, 1.0 + float(self.eps))
            if out is not None:
                return out if id_index is None else _id_mlp_rows(out, g, x, id_index, self.mlp_id, 1.0 + float(self.eps))
        if self.train_eps:
            h = x * (1.0 + self.eps) + ops.spmm(g, x, "sum")
        else:
            h = ops.spmm(g, x, "sum", self_scale=1.0 + float(self.eps))
        out = self.mlp_model(h)
        if id_index is not None:
            out = ops.index_add_rows(out, id_index, self.mlp_id(ops.gather_rows(h, id_index)))
        return out


class GIN(IDGIN):
    """tf_geometric.layers.GIN [3P] (main_zd.py:178-187)"""
    with_id = False

    def __init__(self, mlp_model, eps=0, train_eps=False, **kwargs):
        super().__init__(mlp_model, None, eps=eps, train_eps=train_eps)


class IDGAT(_KerasLike):
    """TfgIDLayer.py:170-388: scaled dot-product attention over in-edges (+ self loops)"""

    def __init__(self, units, attention_units=None, activation=None, use_bias=True, num_heads=1,
                 split_value_heads=True, query_activation=torch.relu, key_activation=torch.relu,
                 drop_rate=0.0, in_features=None, **kwargs):
        super().__init__()
        self.units = units
        self.attention_units = units if attention_units is None else attention_units
        self.activation, self.use_bias = activation, use_bias
        self.num_heads, self.split_value_heads = num_heads, split_value_heads
        self.query_activation, self.key_activation = query_activation, key_activation
        self.drop_rate = drop_rate
        self.query_kernel = self.query_bias = self.key_kernel = self.key_bias = None
        self.kernel = self.kernel_id = self.bias = None
        self._built = False
        if in_features is not None:
            self.build(in_features)

    def build(self, num_features):
        au = self.attention_units
        self.query_kernel = Parameter(torch.empty(num_features, au)); glorot(self.query_kernel)
        self.query_bias = Parameter(torch.zeros(au))
        self.key_kernel = Parameter(torch.empty(num_features, au)); glorot(self.key_kernel)
        self.key_bias = Parameter(torch.zeros(au))
        self.kernel = Parameter(torch.empty(num_features, self.units)); glorot(self.kernel)
        if self.with_id:
            self.kernel_id = Parameter(torch.empty(num_features, self.units)); glorot(self.kernel_id)
        if self.use_bias:
            self.bias = Parameter(torch.zeros(self.units))
        self._built = True

    def call(self, inputs, cache=None, training=None, mask=None, holder=None):
        x, edge_index, id_index, _ = _unpack(inputs, self.with_id)     # edge_weight unused (:248-249)
        self._maybe_build(x)
        H = self.num_heads
        g = get_graph(holder, edge_index, x.size(0), dst_row=0, loops="add")
        def proj(kernel, bias, act):   # act(x @ kernel + bias) in one kernel when act is relu / None
            if act is None or _is_relu(act):
                return ops.dense_fused(x, kernel, bias=bias, relu=act is not None)
            return _apply_act(ops.dense_fused(x, kernel, bias=bias), act)
        Q = proj(self.query_kernel, self.query_bias, self.query_activation)
        K = proj(self.key_kernel, self.key_bias, self.key_activation)
        V = ops.dense_fused(x, self.kernel)
        if id_index is not None:
            V = _id_branch(V, x, id_index, self.kernel_id)
        scale = 1.0 / math.sqrt(self.attention_units // H)
        att = ops.edge_softmax(g, ops.sddmm_dot(g, Q, K, H, scale))
        if training and self.drop_rate > 0.0:
            att = F.dropout(att, p=self.drop_rate, training=True)
        if self.split_value_heads:
            h = ops.spmm_edge_values(g, att, V, H)
        else:
            h = ops.spmm_edge_values(g, att.mean(dim=1, keepdim=True), V, 1)
        if self.bias is not None:
            h = h + self.bias
        return _apply_act(h, self.activation)


class GAT(IDGAT):
    """tf_geometric.layers.GAT [3P] (main_zd.py:82-84)"""
    with_id = False


# ---- GraphGym-style wrappers for the TF-family keys of config/*_tf/*.yaml:29 ----
def _tf_gin_mlp(dim_in, dim_out):
    # main_zd.py:181-186 / 214-225: Dense(d, relu) -> Dense(d) -> BatchNorm -> relu
    from .nn import BatchNorm1d as _BN, Linear as _Lin
    return nn.Sequential(_Lin(dim_in, dim_out), nn.ReLU(), _Lin(dim_out, dim_out),
                         _BN(dim_out, eps=1e-3, momentum=0.01, relu=True))


class _TfBatchLayer(nn.Module):
    uses_id = False

    def forward(self, batch):
        inputs = [batch.node_feature, batch.edge_index]
        if self.uses_id:
            inputs.append(batch.node_id_index)
        batch.node_feature = self.model(inputs, training=self.training, holder=batch)
        return batch


class TfgGCNConv(_TfBatchLayer):
    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GCN(dim_out, use_bias=bias, in_features=dim_in)


class TfgIDGCN(_TfBatchLayer):
    uses_id = True

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = IDGCN(dim_out, use_bias=bias, in_features=dim_in)


class TfgSAGEConv(_TfBatchLayer):
    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = MeanGraphSage(dim_out, activation=None, use_bias=bias, in_features=dim_in)


class TfgIDSAGE(_TfBatchLayer):
    uses_id = True

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = IDSAGE(dim_out, activation=None, use_bias=bias, in_features=dim_in)


class TfgGINConv(_TfBatchLayer):
    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GIN(_tf_gin_mlp(dim_in, dim_out))


class TfgIDGIN(_TfBatchLayer):
    uses_id = True

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = IDGIN(_tf_gin_mlp(dim_in, dim_out), _tf_gin_mlp(dim_in, dim_out))


class TfgGATConv(_TfBatchLayer):
    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = GAT(dim_out, use_bias=bias, in_features=dim_in)


class TfgIDGAT(_TfBatchLayer):
    uses_id = True

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = IDGAT(dim_out, use_bias=bias, in_features=dim_in)

"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by graphgym_amd/.

Functional restatements (explicit parameter tensors, plain torch on CPU) of the
GNN layers on the hot path, in the operation order the reference uses.
PARITY UNPINNED (see ref_ops.py).  [3P] marks layers whose code lives in
tf_geometric / torch_geometric, restated from their published semantics.

TF flavour  (edge_index[0] = destination row, [1] = source col):
    gcn_id / IDGCN      TfgIDLayer.py:461-525     tfg_gcn            [3P] tfg.layers.GCN
    idsage / IDSAGE     TfgIDLayer.py:74-120      tfg_mean_graph_sage [3P]
    idgin / IDGIN       TfgIDLayer.py:143-167     tfg_gin            [3P]
    gat_id / IDGAT      TfgIDLayer.py:246-388     tfg_gat            [3P]
PyG flavour (edge_index[0] = source j, [1] = destination i):
    gcnid_conv          idconv.py:104-189         pyg_gcn_conv       [3P] GCNConv
    sageid_conv         idconv.py:192-263         pyg_sage_conv      [3P] SAGEConv
    gatid_conv          idconv.py:266-347         pyg_gat_conv       [3P] GATConv
    ginid_conv          idconv.py:350-382         pyg_gin_conv       [3P] GINConv
    generalid_conv      idconv.py:16-101          general_conv       generalconv.py:12-114
"""
import torch
import torch.nn.functional as F

from . import ref_ops as R


def _act(h, activation):
    if activation is None:
        return h
    if activation == "relu":
        return torch.relu(h)
    raise ValueError(activation)


def _id_add(h, id_index, h_id):
    # tf.tensor_scatter_nd_add (TfgIDLayer.py:107,165,330,515) == index_add_ (idconv.py:67,155,...)
    return h.index_add(0, id_index, h_id)


# ------------------------------- TF flavour -------------------------------- #
def gcn_id(x, edge_index, id_index, edge_weight, kernel, kernel_id, bias=None, activation=None,
           renorm=True, improved=False):
    """TfgIDLayer.py:478-525"""
    n = x.size(0)
    sparse_adj = R.SparseAdj(edge_index, edge_weight, [n, n])
    normed = R.gcn_norm_adj(sparse_adj, renorm, improved)
    h = x @ kernel
    if id_index is not None:
        h = _id_add(h, id_index, x[id_index] @ kernel_id)
    h = normed @ h
    if bias is not None:
        h = h + bias
    return _act(h, activation)


def tfg_gcn(x, edge_index, edge_weight, kernel, bias=None, activation=None, renorm=True, improved=False):
    """tfg.nn.gcn [3P] = gcn_id without the identity branch"""
    return gcn_id(x, edge_index, None, edge_weight, kernel, None, bias, activation, renorm, improved)


def idsage(x, edge_index, id_index, edge_weight, self_kernel, id_kernel, neighbor_kernel, bias=None,
           activation=None, concat=True, normalize=False):
    """TfgIDLayer.py:74-120"""
    n = x.size(0)
    row, col = edge_index[0], edge_index[1]
    neighbor_x = x[col]
    if edge_weight is not None:
        neighbor_x = neighbor_x * edge_weight[:, None]          # gcn_mapper [3P]
    neighbor_reduced = R.mean_reducer(neighbor_x, row, n)
    neighbor_msg = neighbor_reduced @ neighbor_kernel
    h = x @ self_kernel
    if id_index is not None:
        h = _id_add(h, id_index, x[id_index] @ id_kernel)
    h = torch.cat([h, neighbor_msg], dim=1) if concat else h + neighbor_msg
    if bias is not None:
        h = h + bias
    h = _act(h, activation)
    if normalize:
        h = F.normalize(h, p=2, dim=-1)                          # tf.nn.l2_normalize
    return h


def tfg_mean_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_kernel, bias=None,
                        activation=None, concat=True, normalize=False):
    return idsage(x, edge_index, None, edge_weight, self_kernel, None, neighbor_kernel, bias,
                  activation, concat, normalize)


def idgin(x, edge_index, id_index, mlp, mlp_id, eps=0.0):
    """TfgIDLayer.py:143-167; mlp / mlp_id are callables (main_zd.py:214-225)"""
    n = x.size(0)
    neighbor_h = R.SparseAdj(edge_index, None, [n, n]) @ x
    h = x * (1.0 + eps) + neighbor_h
    if id_index is None:
        return mlp(h)
    h_id = mlp_id(h[id_index])
    return _id_add(mlp(h), id_index, h_id)


def tfg_gin(x, edge_index, mlp, eps=0.0):
    return idgin(x, edge_index, None, mlp, None, eps)


def gat_id(x, edge_index, id_index, query_kernel, query_bias, key_kernel, key_bias, kernel, kernel_id,
           bias=None, activation=None, num_heads=1, split_value_heads=True,
           query_activation="relu", key_activation="relu"):
    """TfgIDLayer.py:269-388 (drop_rate = 0 as in every shipped config)"""
    n = x.size(0)
    sa = R.SparseAdj(edge_index, None, [n, n]).add_self_loop()          # :298
    edge_index = sa.edge_index
    row, col = edge_index[0], edge_index[1]
    Q = _act(x @ query_kernel + query_bias, query_activation)[row]      # :307-311
    K = _act(x @ key_kernel + key_bias, key_activation)[col]            # :316-320
    V = x @ kernel
    if id_index is not None:
        V = _id_add(V, id_index, x[id_index] @ kernel_id)               # :325-330
    Q_ = torch.cat(torch.chunk(Q, num_heads, dim=-1), dim=0)            # :333-334
    K_ = torch.cat(torch.chunk(K, num_heads, dim=-1), dim=0)
    qk_edge_index_ = torch.cat([edge_index + i * n for i in range(num_heads)], dim=1)
    scale = float(Q_.size(-1)) ** 0.5
    att_score_ = (Q_ * K_).sum(-1) / scale                              # :338-339
    att = R.SparseAdj(qk_edge_index_, att_score_, [n * num_heads, n * num_heads]).softmax(axis=-1)
    if split_value_heads:
        V_ = torch.cat(torch.chunk(V, num_heads, dim=-1), dim=0)
    else:
        V_ = V
        att = R.SparseAdj(edge_index.repeat(1, num_heads), att.edge_weight, [n, n])
    h_ = att @ V_
    h = torch.cat(torch.chunk(h_, num_heads, dim=0), dim=-1) if split_value_heads else h_ / num_heads
    if bias is not None:
        h = h + bias
    return _act(h, activation)


def tfg_gat(x, edge_index, query_kernel, query_bias, key_kernel, key_bias, kernel, bias=None,
            activation=None, num_heads=1, split_value_heads=True):
    return gat_id(x, edge_index, None, query_kernel, query_bias, key_kernel, key_bias, kernel, None,
                  bias, activation, num_heads, split_value_heads)


# ------------------------------- PyG flavour ------------------------------- #
def gcnid_conv(x, edge_index, id_index, weight, weight_id, bias=None, improved=False, normalize=True,
               edge_weight=None):
    """GCNIDConvLayer.forward (idconv.py:150-177)"""
    h = x @ weight
    if id_index is not None:
        h = _id_add(h, id_index, x[id_index] @ weight_id)
    if normalize:
        edge_index, norm = R.pyg_gcn_norm(edge_index, h.size(0), edge_weight, improved)
    else:
        norm = edge_weight
    out = R.propagate(edge_index, h, "add", norm)
    return out + bias if bias is not None else out


def generalid_conv(x, edge_index, id_index, weight, weight_id, bias=None, agg="add", normalize_adj=False,
                   improved=False, edge_weight=None):
    """GeneralIDConvLayer.forward (idconv.py:62-97): aggr = cfg.gnn.agg, normalise iff cfg.gnn.normalize_adj"""
    h = x @ weight
    if id_index is not None:
        h = _id_add(h, id_index, x[id_index] @ weight_id)
    if normalize_adj:
        edge_index, norm = R.pyg_gcn_norm(edge_index, h.size(0), edge_weight, improved)
    else:
        norm = edge_weight
    out = R.propagate(edge_index, h, agg, norm)
    return out + bias if bias is not None else out


def general_conv(x, edge_index, weight, weight_self=None, bias=None, agg="add", normalize_adj=False,
                 self_msg="concat", improved=False, edge_weight=None, edge_feature=None):
    """GeneralConvLayer.forward (generalconv.py:62-97)"""
    if self_msg == "concat":
        x_self = x @ weight_self
    h = x @ weight
    if normalize_adj:
        edge_index, norm = R.pyg_gcn_norm(edge_index, h.size(0), edge_weight, improved)
    else:
        norm = edge_weight
    x_msg = R.propagate(edge_index, h, agg, norm, edge_feature)
    if bias is not None:
        x_msg = x_msg + bias                                    # update(), generalconv.py:107-110
    if self_msg == "none":
        return x_msg
    if self_msg == "add":
        return x_msg + h
    if self_msg == "concat":
        return x_msg + x_self
    raise ValueError("self_msg {} not defined".format(self_msg))


def sageid_conv(x, edge_index, id_index, weight, weight_id, bias=None, concat=True, normalize=False,
                edge_weight=None):
    """SAGEIDConvLayer (idconv.py:221-259); the GraphGym wrapper uses concat=True (idconv.py:410)"""
    if not concat:
        edge_index, edge_weight = R.add_remaining_self_loops(edge_index, edge_weight, 1.0, x.size(0))
    aggr_out = R.propagate(edge_index, x, "mean", edge_weight)
    if concat:
        aggr_out = torch.cat([x, aggr_out], dim=-1)
    out = aggr_out @ weight
    if id_index is not None:
        out = _id_add(out, id_index, aggr_out[id_index] @ weight_id)
    if bias is not None:
        out = out + bias
    if normalize:
        out = F.normalize(out, p=2, dim=-1)
    return out


def gatid_conv(x, edge_index, id_index, weight, weight_id, att, bias=None, heads=1, concat=True,
               negative_slope=0.2):
    """GATIDConvLayer (idconv.py:299-342); att [1, heads, 2*out_channels]"""
    n = x.size(0)
    edge_index, _ = R.remove_self_loops(edge_index)
    edge_index, _ = R.add_self_loops(edge_index, num_nodes=n)
    h = x @ weight
    if id_index is not None:
        h = _id_add(h, id_index, x[id_index] @ weight_id)
    out_channels = h.size(1) // heads
    x_j = h[edge_index[0]].view(-1, heads, out_channels)
    x_i = h[edge_index[1]].view(-1, heads, out_channels)
    alpha = (torch.cat([x_i, x_j], dim=-1) * att).sum(dim=-1)
    alpha = F.leaky_relu(alpha, negative_slope)
    alpha = R.softmax(alpha, edge_index[1], n)
    msg = x_j * alpha.view(-1, heads, 1)
    aggr = R.scatter(msg, edge_index[1], n, "add")
    out = aggr.view(-1, heads * out_channels) if concat else aggr.mean(dim=1)
    return out + bias if bias is not None else out


def ginid_conv(x, edge_index, id_index, nn_fn, nn_id_fn, eps=0.0):
    """GINIDConvLayer.forward (idconv.py:367-376)"""
    edge_index, _ = R.remove_self_loops(edge_index)
    h = (1 + eps) * x + R.propagate(edge_index, x, "add")
    if id_index is None:
        return nn_fn(h)
    return _id_add(nn_fn(h), id_index, nn_id_fn(h[id_index]))


def pyg_gcn_conv(x, edge_index, weight, bias=None, edge_weight=None, improved=False):
    """torch_geometric.nn.GCNConv [3P]: transform, add remaining self loops, symmetric norm with the
    degree taken over destinations, sum-aggregate, bias"""
    n = x.size(0)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1))
    ei, ew = R.add_remaining_self_loops(edge_index, edge_weight, 2.0 if improved else 1.0, n)
    deg = R.scatter_add(ew, ei[1], n)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    norm = dis[ei[0]] * ew * dis[ei[1]]
    out = R.propagate(ei, x @ weight, "add", norm)
    return out + bias if bias is not None else out


def pyg_sage_conv(x, edge_index, weight_l, bias_l, weight_r):
    """torch_geometric.nn.SAGEConv [3P]: lin_l(mean_j x_j) + lin_r(x_i); weights stored [out, in]"""
    out = R.propagate(edge_index, x, "mean") @ weight_l.t()
    if bias_l is not None:
        out = out + bias_l
    return out + x @ weight_r.t()


def pyg_gat_conv(x, edge_index, weight, att_i, att_j, bias=None, negative_slope=0.2):
    """torch_geometric.nn.GATConv [3P], heads = 1: additive attention == gatid_conv without the id branch"""
    att = torch.cat([att_i, att_j], dim=-1).view(1, 1, -1)
    return gatid_conv(x, edge_index, None, weight, None, att, bias, 1, True, negative_slope)


def pyg_gin_conv(x, edge_index, nn_fn, eps=0.0):
    """torch_geometric.nn.GINConv [3P]: nn((1+eps) x + sum_j x_j), self loops left in place"""
    return nn_fn((1 + eps) * x + R.propagate(edge_index, x, "add"))


# ------------------------ ego-net expansion (K-next) ------------------------ #
def ego_nets(G, radius=2, return_map=False):
    """graphgym/models/transform.py:11-38 on a networkx graph with nodes 0..n-1.
    Returns (G_ego, node_id_index): centres keep ids 0..n-1, other ego members get fresh ids.
    The order of the fresh ids inside one ego is whatever order networkx iterates the ego's node
    set in (a Python set for small egos), i.e. implementation-defined; with return_map=True the
    function also returns (orig, ego_of): original id and owning centre of every new node, so that
    callers can compare expansions up to that relabelling."""
    import networkx as nx
    n = G.number_of_nodes()
    egos = [G if radius > 4 else nx.ego_graph(G, i, radius=radius) for i in range(n)]
    H = G.__class__()
    id_bias = n
    orig, ego_of = {}, {}
    for i in range(n):
        H.add_node(i, **egos[i].nodes(data=True)[i])
        orig[i], ego_of[i] = i, i
    for i in range(n):
        keys = list(egos[i].nodes)
        keys.remove(i)
        id_cur = egos[i].number_of_nodes() - 1
        mapping = dict(zip(keys, range(id_bias, id_bias + id_cur)))
        id_bias += id_cur
        for k, v in mapping.items():
            orig[v], ego_of[v] = k, i
        ego = nx.relabel_nodes(egos[i], mapping, copy=True)
        H.add_nodes_from(ego.nodes(data=True))
        H.add_edges_from(ego.edges(data=True))
    if return_map:
        N = H.number_of_nodes()
        return (H, torch.arange(n), torch.tensor([orig[k] for k in range(N)]),
                torch.tensor([ego_of[k] for k in range(N)]))
    return H, torch.arange(n)


def compute_identity(edge_index, n, k):
    """graphgym/contrib/transform/identity.py:7-35 restated: add_remaining_self_loops, symmetric normalisation with
    the degree scattered on edge_index[0], DENSE adjacency, diag of its powers 1..k -> [n, k]"""
    ei, w = R.add_remaining_self_loops(edge_index, None, 1.0, n)
    if w is None:
        w = torch.ones(ei.size(1))
    row, col = ei[0], ei[1]
    deg = torch.zeros(n).index_add_(0, row, w)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    val = dis[row] * w * dis[col]
    adj = torch.zeros(n, n).index_put_((row, col), val, accumulate=True)     # to_dense() sums duplicates
    diag_all = [torch.diag(adj)]
    power = adj
    for _ in range(1, k):
        power = power @ adj
        diag_all.append(torch.diag(power))
    return torch.stack(diag_all, dim=1)


# ------------- GraphGym's assembled GNN in eval mode, driven by a reference state dict ------------- #
def graphgym_gnn_eval(state, x, edge_index, node_id_index, *, layers_pre_mp, layers_mp, bn_eps=1e-5, l2norm=True,
                      task="node", ego=True, node_label_index=None, batch=None, num_graphs=None):
    """GNN.forward (graphgym/models/gnn.py:165-168) in eval mode for layer_type 'gcnidconv', stage 'stack',
    layers_post_mp = 1, evaluated from a state dict WITH THE REFERENCE'S OWN KEYS (the checkpoints under
    run/results/node*/1/ckpt hold exactly these):

        pre_mp.Layer_i      GeneralLayer('linear'): nn.Linear without bias -> BatchNorm1d(eval) -> ReLU   layer.py:16-47,70-82
        mp.layer{i}         GeneralLayer('gcnidconv'): GCNIDConvLayer (idconv.py:150-177, no bias under BN) -> BN -> ReLU
        (stage end)         F.normalize(p=2, dim=-1) if cfg.gnn.l2norm                                     gnn.py:79-80
        post_mp             node:  Linear(+bias) on all rows, then rows node_label_index                   head.py:19-37
                            graph: global_add_pool over the centre rows of each graph (transform 'ego':
                                   index_select by node_id_index first), then Linear(+bias)                head.py:96-119,
                                                                                                            pooling.py:12-17
    All arithmetic in the dtype of `x` (tests pass float64)."""
    t = lambda k: torch.as_tensor(state[k]).to(x.dtype)

    def bn_relu(h, prefix):
        h = (h - t(prefix + ".running_mean")) / torch.sqrt(t(prefix + ".running_var") + bn_eps)
        return torch.relu(h * t(prefix + ".weight") + t(prefix + ".bias"))

    h = x
    for i in range(layers_pre_mp):
        h = bn_relu(h @ t(f"pre_mp.Layer_{i}.layer.model.weight").t(), f"pre_mp.Layer_{i}.post_layer.0")
    for i in range(layers_mp):
        h = gcnid_conv(h, edge_index, node_id_index, t(f"mp.layer{i}.layer.model.weight"),
                       t(f"mp.layer{i}.layer.model.weight_id"), bias=None)
        h = bn_relu(h, f"mp.layer{i}.post_layer.0")
    if l2norm:
        h = F.normalize(h, p=2, dim=-1)
    Wp, bp = t("post_mp.layer_post_mp.model.0.model.weight"), t("post_mp.layer_post_mp.model.0.model.bias")
    if task == "node":
        return (h @ Wp.t() + bp)[node_label_index]
    if ego:
        h, batch = h.index_select(0, node_id_index), batch.index_select(0, node_id_index)
    size = int(batch.max()) + 1 if num_graphs is None else num_graphs
    emb = torch.zeros(size, h.size(1), dtype=h.dtype).index_add_(0, batch, h)      # scatter(reduce='add')
    return emb @ Wp.t() + bp

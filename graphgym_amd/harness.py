"""The callers on either side of the hot path, restated just far enough to drive it:
model shapes of the reference's two entry points and its training step order.

    TfgNodeModel   the eight keras models of main_zd.py:28-243 (n conv layers -> Flatten ->
                   Dense(256, relu) -> Dense(num_labels)); GCN / GAT families hard-code 3 layers
                   (main_zd.py:33-35,82-84), SAGE / GIN use cfg.gnn.layers_mp (:131-132,178-187)
    GNNStack       graphgym/models/gnn.py:123-168 with stage 'stack': pre_mp linears ->
                   layers_mp x GeneralLayer (conv -> BN -> dropout -> act, layer.py:16-47) ->
                   row L2-normalise (gnn.py:79-80) -> node head MLP + label gather (head.py:19-37)
    train_step     zero_grad -> forward -> loss -> backward -> [gradient all-reduce] -> step
                   (graphgym/train.py:18-25, 47-56)
    tfg_loss       mean softmax-CE over node_label_index + 5e-4 * sum ||kernel||^2 / 2
                   (graphgym/loss.py:53-68)
"""
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L
from . import nn as mpnn
from .config import cfg
from .registry import layer_dict


class Batch(types.SimpleNamespace):
    """the fields of a DeepSNAP batch the path reads (idconv.py:390-441, head.py:27-32)"""

    def to(self, device):
        for k, v in list(vars(self).items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self


# ---- TF path: main_zd.py model shapes ----------------------------------------------------
_TF_LAYER = {
    "gcn": (L.GCN, False), "idgcn": (L.IDGCN, True), "gat": (L.GAT, False), "idgat": (L.IDGAT, True),
    "sage": (L.MeanGraphSage, False), "idsage": (L.IDSAGE, True), "gin": (L.GIN, False), "idgin": (L.IDGIN, True),
}


def _keras_gin_mlp(dim_in, dim):
    # main_zd.py:181-186: Dense(d, relu) -> Dense(d) -> BatchNormalization -> relu
    # BatchNormalization + relu run as one engine op (graphgym_amd.nn.BatchNorm1d(relu=True))
    # Dense(d, relu): the ReLU rides in the transform kernel's store; nn.Identity keeps the module indices (and so the
    # state-dict keys) of Linear -> ReLU -> Linear -> BatchNorm
    return nn.Sequential(mpnn.Linear(dim_in, dim, relu=True), nn.Identity(), mpnn.Linear(dim, dim),
                         mpnn.BatchNorm1d(dim, eps=1e-3, momentum=0.01, relu=True))


class TfgNodeModel(nn.Module):
    def __init__(self, kind, dim_in, dim_inner, num_labels, layers_mp=3):
        super().__init__()
        cls, self.with_id = _TF_LAYER[kind]
        self.kind = kind
        n_layers = 3 if kind in ("gcn", "idgcn", "gat", "idgat") else layers_mp
        convs = []
        for i in range(n_layers):
            d_in = dim_in if i == 0 else dim_inner
            if kind in ("gin", "idgin"):
                mlps = [_keras_gin_mlp(d_in, dim_inner)] + ([_keras_gin_mlp(d_in, dim_inner)] if self.with_id else [])
                convs.append(cls(*mlps))
            else:
                convs.append(cls(dim_inner, activation=torch.relu, in_features=d_in))
        self.convs = nn.ModuleList(convs)
        self.mlp = nn.Sequential(nn.Flatten(), mpnn.Linear(dim_inner, 256, relu=True), nn.Identity(),
                                 mpnn.Linear(256, num_labels))

    def kernel_parameters(self):
        """the variables compute_loss_Tfg regularises: every keras variable whose name contains "kernel"
        (loss.py:65) — conv kernels and Dense kernels, not biases / BN"""
        out = []
        for name, p in self.named_parameters():
            leaf = name.split(".")[-1]
            if "kernel" in leaf or (leaf == "weight" and p.dim() == 2):
                out.append(p)
        return out

    def forward(self, inputs, holder=None):
        x, edge_index = inputs[0], inputs[1]
        id_index = inputs[2] if self.with_id else None
        h = x
        for conv in self.convs:
            args = [h, edge_index] + ([id_index] if self.with_id else [])
            h = conv(args, training=self.training, holder=holder)
        return self.mlp(h)


def tfg_loss(logits, node_label_index, labels, kernel_params, ego=False):
    """graphgym/loss.py:53-68"""
    if ego:
        labels = labels[node_label_index]
    ce = mpnn.softmax_cross_entropy(logits, labels, node_label_index)   # == F.cross_entropy(logits[idx], labels)
    l2 = sum((p * p).sum() / 2 for p in kernel_params)     # tf.nn.l2_loss = sum(t^2) / 2
    return ce + 5e-4 * l2


# ---- torch path: GraphGym's GNN with the 'stack' stage --------------------------------------
class GeneralLayer(nn.Module):
    """graphgym/models/layer.py:16-47"""

    def __init__(self, name, dim_in, dim_out, has_act=True, has_bn=True, has_l2norm=False, **kwargs):
        super().__init__()
        self.has_l2norm = has_l2norm
        has_bn = has_bn and cfg.gnn.batchnorm
        self.layer = layer_dict[name](dim_in, dim_out, bias=not has_bn, **kwargs)
        post = []
        fuse_relu = has_bn and has_act and cfg.gnn.act == "relu" and not cfg.gnn.dropout > 0
        if has_bn:   # BN (+ the ReLU right behind it) on the engine's HBM-bound passes
            post.append(mpnn.BatchNorm1d(dim_out, eps=cfg.bn.eps, momentum=cfg.bn.mom, relu=fuse_relu))
        if cfg.gnn.dropout > 0:
            post.append(nn.Dropout(p=cfg.gnn.dropout, inplace=cfg.mem.inplace))
        if has_act and not fuse_relu:
            post.append(nn.ReLU() if cfg.gnn.act == "relu" else getattr(nn, cfg.gnn.act)())
        self.post_layer = nn.Sequential(*post)

    def forward(self, batch):
        batch = self.layer(batch)
        batch.node_feature = self.post_layer(batch.node_feature)
        if self.has_l2norm:
            batch.node_feature = F.normalize(batch.node_feature, p=2, dim=1)
        return batch


class _LinearLayer(nn.Module):
    """graphgym/models/layer.py:70-82 ('linear' key)"""

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = nn.Linear(dim_in, dim_out, bias=bias)

    def forward(self, batch):
        batch.node_feature = self.model(batch.node_feature)
        return batch


layer_dict.setdefault("linear", _LinearLayer)


class GNNStack(nn.Module):
    """gnn.py:123-168 with stage_type='stack' and the node head (head.py:19-37)"""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        d = dim_in
        mods = []
        for _ in range(cfg.gnn.layers_pre_mp):
            mods.append(GeneralLayer("linear", d, cfg.gnn.dim_inner))
            d = cfg.gnn.dim_inner
        self.pre_mp = nn.Sequential(*mods)
        self.mp = nn.ModuleList()
        for i in range(cfg.gnn.layers_mp):
            self.mp.append(GeneralLayer(cfg.gnn.layer_type, d if i == 0 else cfg.gnn.dim_inner, cfg.gnn.dim_inner))
        d = cfg.gnn.dim_inner if cfg.gnn.layers_mp > 0 else d
        post = []
        for _ in range(max(cfg.gnn.layers_post_mp, 1) - 1):
            post.append(GeneralLayer("linear", d, d))
        self.post_mp = nn.Sequential(*post)
        self.out = nn.Linear(d, dim_out, bias=True)

    def forward(self, batch):
        batch = self.pre_mp(batch)
        for layer in self.mp:
            batch = layer(batch)
        if cfg.gnn.l2norm:
            batch.node_feature = F.normalize(batch.node_feature, p=2, dim=-1)
        batch = self.post_mp(batch)
        pred = self.out(batch.node_feature)
        idx = batch.node_label_index
        label = batch.node_label if idx.shape[0] == batch.node_label.shape[0] else batch.node_label[idx]
        return pred[idx], label


def train_step(model, optimizer, forward_loss, bucket=None):
    """train.py:18-25 / 47-56 with the data-parallel exchange added between backward and step"""
    optimizer.zero_grad()
    loss = forward_loss()
    loss.backward()
    if bucket is not None:
        bucket.all_reduce_mean()
    optimizer.step()
    return loss.detach()


class GraphedTrainStep:
    """The whole step (forward, loss, backward, optimizer) captured once into a HIP graph and replayed.

    The reference trains full-batch on small graphs (batch_size >= dataset in config/*_tf/*.yaml), where a
    step is ~100 short kernels and launch overhead dominates; shapes repeat every epoch, so one capture
    serves the run.  Requirements: the batch's CSR / plan / transpose are already cached (run a few eager
    steps first — done here), the optimizer is capture-safe (torch.optim.Adam(capturable=True)), and
    forward_loss() reads only tensors that stay at fixed addresses.  Engine ops are capture-safe: they
    enqueue on torch's current stream and never synchronise once the graph structures are cached."""

    def __init__(self, model, optimizer, forward_loss, warmup=3):
        self.optimizer = optimizer
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                train_step(model, optimizer, forward_loss)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = forward_loss()
            self.loss.backward()
            optimizer.step()

    def __call__(self):
        self.graph.replay()
        return self.loss

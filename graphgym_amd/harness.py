"""The callers on either side of the hot path, restated just far enough to drive it:
model shapes of the reference's two entry points and its training step order.

    TfgNodeModel   the eight keras models of main_zd.py:28-243 (n conv layers -> Flatten ->
                   Dense(256, relu) -> Dense(num_labels)); GCN / GAT families hard-code 3 layers
                   (main_zd.py:33-35,82-84), SAGE / GIN use cfg.gnn.layers_mp (:131-132,178-187)
    GNN            graphgym/models/gnn.py:123-168 with stage 'stack', module for module (the reference's own
                   state dicts load with strict=True): pre_mp.Layer_i -> mp.layer{i} (GeneralLayer: conv -> BN ->
                   dropout -> act, layer.py:16-47) -> row L2-normalise (gnn.py:79-80) -> post_mp.layer_post_mp
                   (node head + label gather head.py:19-37, or graph head with ego add-pool head.py:96-119)
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

    def prepare(self, inputs, holder):
        """build the batch's graph structures ahead of the step (every conv layer's prepare(); layers of one model share
        them through `holder`'s cache).  Returns False when a layer family has no prepare(): the step then builds lazily."""
        ok = True
        feats, edge_index = inputs[0], inputs[1]
        rest = [inputs[2]] if self.with_id else []
        for conv in self.convs:
            if not hasattr(conv, "prepare"):
                ok = False
                continue
            conv.prepare([feats, edge_index] + rest, holder)
            width = getattr(conv, "units", None)
            if width is not None and width != feats.size(1):     # later layers see the hidden width (order selection);
                feats = torch.empty((feats.size(0), width), device="meta")      # only its shape is read
        return ok

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


# ---- torch path: GraphGym's GNN, module for module ------------------------------------------
# The module tree (attribute names, nesting, which layers carry a bias) is the reference's, so that its state dicts load
# with strict=True: tests/test_reference_checkpoint.py loads the two trained gcnidconv checkpoints the reference holds
# (run/results/node*/1/ckpt/999.ckpt -> tests/golden/ref_ckpt_*.npz) into this class and into nothing else.
#
#   pre_mp.Layer_{i}.layer.model.weight            GeneralMultiLayer('linear', ...)      layer.py:50-67, gnn.py:23-25
#   pre_mp.Layer_{i}.post_layer.0.*                BatchNorm1d of GeneralLayer           layer.py:26-35
#   mp.layer{i}.layer.model.{weight,weight_id}     GNNStackStage -> GeneralLayer -> conv  gnn.py:65-74, idconv.py:396-404
#   post_mp.layer_post_mp.model.{j}...             GNNNodeHead / GNNGraphHead -> MLP      head.py:19-37,96-119, layer.py:107-132
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
        if isinstance(batch, torch.Tensor):                      # layer.py:39-42 (the head's MLP passes tensors)
            batch = self.post_layer(batch)
            return F.normalize(batch, p=2, dim=1) if self.has_l2norm else batch
        batch.node_feature = self.post_layer(batch.node_feature)
        if self.has_l2norm:
            batch.node_feature = F.normalize(batch.node_feature, p=2, dim=1)
        return batch


class GeneralMultiLayer(nn.Module):
    """graphgym/models/layer.py:50-67: children named Layer_{i}"""

    def __init__(self, name, num_layers, dim_in, dim_out, dim_inner=None, final_act=True, **kwargs):
        super().__init__()
        dim_inner = dim_in if dim_inner is None else dim_inner
        for i in range(num_layers):
            d_in = dim_in if i == 0 else dim_inner
            d_out = dim_out if i == num_layers - 1 else dim_inner
            has_act = final_act if i == num_layers - 1 else True
            self.add_module('Layer_{}'.format(i), GeneralLayer(name, d_in, d_out, has_act, **kwargs))

    def forward(self, batch):
        for layer in self.children():
            batch = layer(batch)
        return batch


class Linear(nn.Module):
    """graphgym/models/layer.py:70-82 (the 'linear' key): the nn.Linear sits under `.model`"""

    def __init__(self, dim_in, dim_out, bias=False, **kwargs):
        super().__init__()
        self.model = mpnn.Linear(dim_in, dim_out, bias=bias)     # nn.Linear's parameters and state dict, engine kernels

    def forward(self, batch):
        if isinstance(batch, torch.Tensor):
            return self.model(batch)
        batch.node_feature = self.model(batch.node_feature)
        return batch


layer_dict.setdefault("linear", Linear)


class MLP(nn.Module):
    """graphgym/models/layer.py:107-132: model = Sequential([GeneralMultiLayer('linear', n - 1), Linear]) or
    Sequential([Linear]) for n <= 1"""

    def __init__(self, dim_in, dim_out, bias=True, dim_inner=None, num_layers=2, **kwargs):
        super().__init__()
        dim_inner = dim_in if dim_inner is None else dim_inner
        layers = []
        if num_layers > 1:
            layers.append(GeneralMultiLayer('linear', num_layers - 1, dim_in, dim_inner, dim_inner, final_act=True))
            layers.append(Linear(dim_inner, dim_out, bias))
        else:
            layers.append(Linear(dim_in, dim_out, bias))
        self.model = nn.Sequential(*layers)

    def forward(self, batch):
        if isinstance(batch, torch.Tensor):
            return self.model(batch)
        batch.node_feature = self.model(batch.node_feature)
        return batch


def GNNLayer(dim_in, dim_out, has_act=True):       # gnn.py:19-20
    return GeneralLayer(cfg.gnn.layer_type, dim_in, dim_out, has_act)


def GNNPreMP(dim_in, dim_out):                     # gnn.py:23-25
    return GeneralMultiLayer('linear', cfg.gnn.layers_pre_mp, dim_in, dim_out, dim_inner=dim_out, final_act=True)


class GNNStackStage(nn.Module):
    """graphgym/models/gnn.py:65-81: children named layer{i}; row L2-normalisation behind the last one"""

    def __init__(self, dim_in, dim_out, num_layers):
        super().__init__()
        for i in range(num_layers):
            self.add_module('layer{}'.format(i), GNNLayer(dim_in if i == 0 else dim_out, dim_out))
        self.dim_out = dim_out

    def forward(self, batch):
        for layer in self.children():
            batch = layer(batch)
        if cfg.gnn.l2norm:
            batch.node_feature = F.normalize(batch.node_feature, p=2, dim=-1)
        return batch


class GNNNodeHead(nn.Module):
    """graphgym/models/head.py:19-37"""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.layer_post_mp = MLP(dim_in, dim_out, num_layers=cfg.gnn.layers_post_mp, bias=True)

    def _apply_index(self, batch):
        idx = batch.node_label_index
        if idx.shape[0] == batch.node_label.shape[0]:
            return batch.node_feature[idx], batch.node_label
        return batch.node_feature[idx], batch.node_label[idx]

    def forward(self, batch):
        batch = self.layer_post_mp(batch)
        return self._apply_index(batch)


class GNNGraphHead(nn.Module):
    """graphgym/models/head.py:96-119: pool (ego batches: the centre rows only, pooling.py:12-17), then the MLP"""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        from .pooling import pooling_dict
        self.layer_post_mp = MLP(dim_in, dim_out, num_layers=cfg.gnn.layers_post_mp, bias=True)
        self.pooling_fun = pooling_dict[cfg.model.graph_pooling]

    def forward(self, batch):
        if cfg.dataset.transform == 'ego':
            graph_emb = self.pooling_fun(batch.node_feature, batch.batch, batch.node_id_index)
        else:
            graph_emb = self.pooling_fun(batch.node_feature, batch.batch)
        batch.graph_feature = self.layer_post_mp(graph_emb)
        return batch.graph_feature, batch.graph_label


head_dict = {'node': GNNNodeHead, 'graph': GNNGraphHead}     # head.py:122-127 ('edge' / 'link_pred' are off the path)


class GNN(nn.Module):
    """graphgym/models/gnn.py:123-168 with stage_type 'stack' (the skip stages concatenate / add around the same
    layers and are not on the path the BASELINE configs drive).  The feature-augmentation `preprocess` module of the
    reference holds no parameters and is outside the path: inputs arrive already assembled."""

    def __init__(self, dim_in, dim_out, **kwargs):
        super().__init__()
        if getattr(cfg.gnn, "stage_type", "stack") != "stack":
            raise ValueError("harness.GNN restates GNNStackStage only (cfg.gnn.stage_type = 'stack')")
        d_in = dim_in
        if cfg.gnn.layers_pre_mp > 0:
            self.pre_mp = GNNPreMP(d_in, cfg.gnn.dim_inner)
            d_in = cfg.gnn.dim_inner
        if cfg.gnn.layers_mp > 0:
            self.mp = GNNStackStage(dim_in=d_in, dim_out=cfg.gnn.dim_inner, num_layers=cfg.gnn.layers_mp)
            d_in = self.mp.dim_out
        self.post_mp = head_dict[getattr(cfg.dataset, "task", "node")](dim_in=d_in, dim_out=dim_out)

    def forward(self, batch):
        for module in self.children():
            batch = module(batch)
        return batch


GNNStack = GNN      # the name rounds 1-2 used


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

"""Drop-in registration of the engine's layers under GraphGym's ``layer_type`` keys.

Importing this module is the whole integration:

  * the ID-GNN keys 'idconv', 'gcnidconv', 'sageidconv', 'gatidconv', 'ginidconv'
    (registered by graphgym/contrib/layer/idconv.py:444-448 in the reference),
  * the TF-family keys 'Tfg-gcnconv', 'Tfg-sageconv', 'Tfg-gatconv', 'Tfg-ginconv',
    'Tfg-idgcn', 'Tfg-idsage', 'Tfg-idgat', 'Tfg-idgin' (config/*_tf/*.yaml:29, dispatched
    by main_zd.py:299-308 in the reference),
  * the built-in keys 'gcnconv', 'sageconv', 'gatconv', 'ginconv', 'generalconv'
    (graphgym/models/layer.py:224-235).

``register_layer`` raises KeyError on a duplicate (register.py:6-10), and built-ins shadow
registered keys (layer.py:238), so taking over an existing key is done by assignment into the
dictionaries — ``install(override=True)``, the default.
"""
from . import layers as L
from . import registry as R

ID_KEYS = {
    'idconv': L.GeneralIDConv,
    'gcnidconv': L.GCNIDConv,
    'sageidconv': L.SAGEIDConv,
    'gatidconv': L.GATIDConv,
    'ginidconv': L.GINIDConv,
}
TF_KEYS = {
    'Tfg-gcnconv': L.TfgGCNConv,
    'Tfg-sageconv': L.TfgSAGEConv,
    'Tfg-gatconv': L.TfgGATConv,
    'Tfg-ginconv': L.TfgGINConv,
    'Tfg-idgcn': L.TfgIDGCN,
    'Tfg-idsage': L.TfgIDSAGE,
    'Tfg-idgat': L.TfgIDGAT,
    'Tfg-idgin': L.TfgIDGIN,
}
BUILTIN_KEYS = {
    'gcnconv': L.GCNConv,
    'sageconv': L.SAGEConv,
    'gatconv': L.GATConv,
    'ginconv': L.GINConv,
    'generalconv': L.GeneralConv,
}
ALL_KEYS = {**ID_KEYS, **TF_KEYS, **BUILTIN_KEYS}


def install(override=True):
    """Register every key; returns the list of keys now served by the engine.

    override=False keeps the reference's semantics strictly: keys that already exist are
    left alone (register_layer would raise KeyError)."""
    taken = []
    dicts = [R.layer_dict]
    try:  # the resolved dictionaries GeneralLayer actually indexes (layer.py:24,238)
        import sys
        for name in ("graphgym.models.layer", "graphgym.models.layer_pyg"):
            mod = sys.modules.get(name)
            if mod is not None and hasattr(mod, "layer_dict"):
                dicts.append(mod.layer_dict)
    except Exception:
        pass
    for key, cls in ALL_KEYS.items():
        for d in dicts:
            if key in d and d[key] is not cls:
                if override:
                    d[key] = cls
            elif key not in d:
                if d is R.layer_dict:
                    R.register_layer(key, cls)
                else:
                    d[key] = cls
        if R.layer_dict.get(key) is cls:
            taken.append(key)
    return taken


installed_keys = install(override=True)


# ---- the post-ops of GraphGym's layer wrapper on the engine -------------------------------------------------
def _swap_post_layer(post):
    """graphgym/models/layer.py:26-35 builds [BatchNorm1d, (Dropout), act] as plain torch modules inside GeneralLayer —
    outside what a layer_dict entry can replace.  Swap the BatchNorm1d (and the ReLU right behind it) for the engine's
    BatchNorm1d(relu=...) IN PLACE: same Parameter / buffer objects (an optimizer built earlier keeps working), same
    positions in the Sequential (state_dict keys unchanged; the fused ReLU's slot becomes nn.Identity)."""
    import torch.nn as nn
    from . import nn as mpnn
    swapped = 0
    mods = list(post)
    for i, m in enumerate(mods):
        if type(m) is not nn.BatchNorm1d:
            continue
        new = mpnn.BatchNorm1d(m.num_features, eps=m.eps, momentum=m.momentum, affine=m.affine,
                               track_running_stats=m.track_running_stats, relu=False)
        for name in ("weight", "bias"):
            new._parameters[name] = m._parameters.get(name)
        for name in ("running_mean", "running_var", "num_batches_tracked"):
            new._buffers[name] = m._buffers.get(name)
        new.train(m.training)
        if i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU:
            new.relu = True
            post[i + 1] = nn.Identity()
        post[i] = new
        swapped += 1
    return swapped


def _folded_forward(self, batch):
    """GeneralLayer.forward (layer.py:37-47) in eval mode for the GCN-type layers: transform, then ONE aggregation whose
    row flush carries conv bias + BatchNorm1d(eval) affine + ReLU + row L2-normalise (mp_spmm_csr_epilogue_f32)"""
    import torch
    import torch.nn as nn
    from . import nn as mpnn, ops
    conv = getattr(self.layer, "model", None)
    post = list(self.post_layer)
    foldable = (not self.training and not torch.is_grad_enabled() and isinstance(conv, (L.GCNConvLayer, L.GCNIDConvLayer))
                and not isinstance(batch, torch.Tensor) and getattr(conv, "_agg", "add") in ("add", "sum")
                and all(isinstance(m, (mpnn.BatchNorm1d, nn.BatchNorm1d, nn.Identity, nn.ReLU, nn.Dropout)) for m in post))
    bns = [m for m in post if isinstance(m, nn.BatchNorm1d)]
    if not foldable or len(bns) > 1 or (bns and not bns[0].track_running_stats):
        return self._mp_orig_forward(batch)
    x, ei = batch.node_feature, batch.edge_index
    g = conv.graph_for(x, ei, holder=batch)
    h = ops.dense_fused(x, conv.weight)
    if isinstance(conv, L.GCNIDConvLayer):
        h = L._id_branch(h, x, batch.node_id_index, conv.weight_id)
    d = h.size(1)
    scale = torch.ones(d, device=h.device)
    shift = torch.zeros(d, device=h.device) if conv.bias is None else conv.bias.detach().clone()
    relu = any(type(m) is nn.ReLU for m in post)
    if bns:
        bn = bns[0]
        relu = relu or bool(getattr(bn, "relu", False))
        s = torch.rsqrt(bn.running_var + bn.eps) * (bn.weight if bn.weight is not None else 1.0)
        shift = (shift - bn.running_mean) * s + (bn.bias if bn.bias is not None else 0.0)
        scale = s
    batch.node_feature = ops.spmm_fused_eval(g, h, "sum", col_scale=scale, col_shift=shift, relu=relu,
                                             l2norm=bool(self.has_l2norm))
    return batch


def accelerate(model, fold_eval=True):
    """Put the post-ops of a BUILT GraphGym model on the engine.  Walks the model for GraphGym's layer wrapper
    (GeneralLayer: `.layer`, `.post_layer`, `.has_l2norm`; graphgym/models/layer.py:16-47) and

      * swaps post_layer's nn.BatchNorm1d (+ the ReLU behind it) for graphgym_amd.nn.BatchNorm1d(relu=...) — training
        mode then runs the statistics, normalisation, ReLU and the whole backward as HBM-bound engine passes instead
        of torch's kernels (its batch_norm_backward_reduce takes 0.5 s per call on a [10^7, 256] activation);
      * in eval mode (under no_grad) folds conv bias + BatchNorm affine + ReLU + the row L2-normalisation into the
        aggregation's row flush for the GCN-type layers (gcnconv, gcnidconv).

    Returns the number of wrappers touched.  Call it once after create_model(); state_dict keys, parameter objects
    and numerics (to fp32 rounding) are unchanged."""
    import types
    import torch.nn as nn
    touched = 0
    for mod in model.modules():
        post = getattr(mod, "post_layer", None)
        if not isinstance(post, nn.Sequential) or not hasattr(mod, "layer"):
            continue
        n = _swap_post_layer(post)
        if fold_eval and not hasattr(mod, "_mp_orig_forward"):
            mod._mp_orig_forward = mod.forward
            mod.forward = types.MethodType(_folded_forward, mod)
        touched += 1 if (n or fold_eval) else 0
    return touched

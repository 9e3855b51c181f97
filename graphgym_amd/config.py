"""Minimal stand-in for the keys of GraphGym's global ``cfg`` that the hot path reads
(graphgym/config.py:313-372,409-420).  When the real ``graphgym.config`` is importable
(a deployment with yacs installed) that object is used instead, so the layers see the
experiment's actual configuration; otherwise the same YAML files are read with PyYAML.
"""
import types

import yaml


def _defaults():
    cfg = types.SimpleNamespace()
    cfg.device = "auto"                      # config.py:33
    cfg.num_threads = 6                      # config.py:57
    cfg.dataset = types.SimpleNamespace(transform="none", augment_feature=[], task="node")
    cfg.model = types.SimpleNamespace(graph_pooling="add", loss_fun="cross_entropy")      # config.py:285-301
    cfg.train = types.SimpleNamespace(batch_size=16)
    cfg.gnn = types.SimpleNamespace(
        layers_pre_mp=0, layers_mp=2, layers_post_mp=1, dim_inner=16,
        layer_type="generalconv", stage_type="stack", batchnorm=True, act="relu",
        dropout=0.0, agg="add", flow="source_to_target", normalize_adj=False,
        self_msg="concat", att_heads=1, l2norm=True)                     # config.py:313-369
    cfg.bn = types.SimpleNamespace(eps=1e-5, mom=0.1)                     # config.py:409-412
    cfg.mem = types.SimpleNamespace(inplace=False)                        # config.py:420
    cfg.optim = types.SimpleNamespace(base_lr=0.01, max_epoch=200, weight_decay=5e-4)
    return cfg


try:  # a real GraphGym install wins
    from graphgym.config import cfg  # noqa: F401
    HAVE_GRAPHGYM_CFG = True
except Exception:  # yacs / graphgym absent: same keys, plain namespace
    cfg = _defaults()
    HAVE_GRAPHGYM_CFG = False


def load_cfg(path, target=None):
    """merge a GraphGym YAML (e.g. config/gcnconv_tf/gcnconv_node_scalefree.yaml) into cfg"""
    target = cfg if target is None else target
    with open(path) as f:
        data = yaml.safe_load(f) or {}
    for k, v in data.items():
        if isinstance(v, dict):
            node = getattr(target, k, None)
            if node is None:
                node = types.SimpleNamespace()
                setattr(target, k, node)
            for kk, vv in v.items():
                setattr(node, kk, vv)
        else:
            setattr(target, k, v)
    return target

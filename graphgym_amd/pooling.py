"""Graph pooling on the engine — graphgym/models/pooling.py:12-33.

global_{add,mean,max}_pool(x, batch, id=None, size=None): with dataset.transform == 'ego' only the
centre rows (``id``) are pooled (pooling.py:14-16).  A pool is an aggregation whose operator has one
row per graph and one column per node, so it runs on the same kernel as the layer aggregation
(rectangular CSR: rows = graphs, cols = nodes); the operator is cached on a holder like any graph.
"""
import torch

from . import ops
from .config import cfg
from .graph import CSRGraph


def pool_operator(batch, num_nodes, id=None, size=None):
    """CSR [size, num_nodes] with entry (g, v) = 1 for every pooled node v of graph g"""
    nodes = torch.arange(num_nodes, device=batch.device) if id is None else id.to(torch.int64)
    owner = batch.to(torch.int64) if id is None else batch.to(torch.int64).index_select(0, nodes)
    size = int(batch.max().item()) + 1 if size is None else int(size)
    return CSRGraph.from_edge_index(torch.stack([nodes, owner]), size, num_cols=int(num_nodes))   # [source; destination]


def _pool(x, batch, id, size, reduce, operator=None):
    if cfg.dataset.transform != 'ego':
        id = None
    g = operator if operator is not None else pool_operator(batch, x.size(0), id, size)
    return ops.spmm(g, x, reduce)


def global_add_pool(x, batch, id=None, size=None, operator=None):
    return _pool(x, batch, id, size, "sum", operator)


def global_mean_pool(x, batch, id=None, size=None, operator=None):
    return _pool(x, batch, id, size, "mean", operator)


def global_max_pool(x, batch, id=None, size=None, operator=None):
    return _pool(x, batch, id, size, "max", operator)


pooling_dict = {'add': global_add_pool, 'mean': global_mean_pool, 'max': global_max_pool}

try:  # serve the same keys through GraphGym's registry when it is importable
    import graphgym.register as _reg
    for _k, _f in pooling_dict.items():
        _reg.pooling_dict[_k] = _f
except Exception:
    pass

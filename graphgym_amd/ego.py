"""Ego-net batches on the GPU — graphgym/models/transform.py:11-38 (ID-GNN "Full" sampler)
for a batch of centre nodes, through mp_ego_expand_* (csrc/ego.hip)."""
import ctypes as C

import torch

from ._lib import check, lib, ptr
from .graph import CSRGraph, _require_hip, _stream


def ego_batch(base, centres, radius):
    """Expand `centres` (LongTensor [B] on the GPU) of the symmetric graph `base` (CSRGraph) into the
    disjoint union of their radius-`radius` ego nets.

    Returns (edge_index [2, E] int64 in PyG convention, orig_node [N'] int64, node_id_index [B] int64,
    ego_of_node [N'] int32): node k of the expanded graph is original node orig_node[k]; the centres
    are nodes 0..B-1 (node_id_index = arange(B), transform.py:38)."""
    _require_hip(centres, "centres")
    L = lib()
    dev = base.device
    cen = centres.to(torch.int64).contiguous()
    B, N = cen.numel(), base.num_nodes
    with torch.cuda.device(dev):
        nb = C.c_size_t(0)
        check(L.mp_ego_ws_bytes(N, B, C.byref(nb)), "mp_ego_ws_bytes")
        ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        counts = (C.c_int64 * 2)()
        check(L.mp_ego_expand_count(ptr(base.rowptr), ptr(base.col), N, ptr(cen), B, int(radius), ptr(ws),
                                    nb.value, counts, _stream()), "mp_ego_expand_count")
        n_out, e_out = int(counts[0]), int(counts[1])
        ei = torch.empty((2, max(e_out, 1)), dtype=torch.int64, device=dev)
        orig = torch.empty(max(n_out, 1), dtype=torch.int64, device=dev)
        ego_of = torch.empty(max(n_out, 1), dtype=torch.int32, device=dev)
        check(L.mp_ego_expand_emit(ptr(base.rowptr), ptr(base.col), N, ptr(cen), B, ptr(ws), nb.value,
                                   ptr(ei[0]), ptr(ei[1]), ptr(orig), ptr(ego_of), _stream()),
              "mp_ego_expand_emit")
    return ei[:, :e_out], orig[:n_out], torch.arange(B, device=dev), ego_of[:n_out]

"""Ego-net batches on the GPU — graphgym/models/transform.py:11-38 (ID-GNN "Full" sampler)
for a batch of centre nodes, through mp_ego_expand (csrc/ego.hip)."""
import ctypes as C

import torch

from ._lib import ALLOC_FN, FREE_FN, EgoResult, EngineError, check, lib, ptr
from .graph import CSRGraph, _require_hip, _stream  # noqa: F401

TAG_SCRATCH, TAG_EDGES, TAG_ORIG, TAG_EGO_OF, TAG_ROWPTR, TAG_COL, TAG_EID = range(7)
FLAG_CSR, FLAG_CSR_SELF_LOOPS = 1, 2
last_stats = {}          # the last call's sizes (bench / tests read them): nodes, edges, candidates, scratch_peak_bytes


def ego_batch(base, centres, radius, csr=None):
    """Expand `centres` (LongTensor [B] on the GPU) of the symmetric graph `base` (CSRGraph) into the
    disjoint union of their radius-`radius` ego nets.

    Returns (edge_index [2, E] int64 in PyG convention, orig_node [N'] int64, node_id_index [B] int64,
    ego_of_node [N'] int32): node k of the expanded graph is original node orig_node[k]; the centres
    are nodes 0..B-1 (node_id_index = arange(B), transform.py:38).

    csr = "none" | "add": a fifth value, the batch's CSRGraph itself — what CSRGraph.from_edge_index(edge_index, N')
    (csr="none") or from_edge_index(..., add_self_loops=True) (csr="add") would build, entry for entry, written by the
    expansion in the engine's CSR order instead of being sorted out of the COO list again; the graph is flagged
    symmetric (an induced subgraph of a symmetric graph: its transpose is itself).  Needs a base graph without explicit
    self loops and radius <= 4 (otherwise the fifth value is None and the caller builds the CSR the usual way).

    Work and memory follow the ego nets (members + candidate neighbours of one level at a time; nothing is sized by the
    base graph): the engine asks for its buffers through a callback that hands out torch tensors on the current stream
    and returns scratch to torch's cache as each level finishes.  Synchronises the current stream."""
    _require_hip(centres, "centres")
    L = lib()
    dev = base.device
    cen = centres.to(torch.int64).contiguous()
    B, N = cen.numel(), base.num_nodes
    held, outs, failure = {}, {}, []

    def _alloc(nbytes, tag, _user):
        try:
            t = torch.empty(max(int(nbytes), 8), dtype=torch.uint8, device=dev)
        except Exception as e:          # (an exception must not cross the C frame: NULL -> MP_ERR_WORKSPACE)
            failure.append(e)
            return None
        if tag == TAG_SCRATCH:
            held[t.data_ptr()] = t
        else:
            outs[tag] = t
        return t.data_ptr()

    def _free(p, _user):
        held.pop(p, None)

    alloc_cb, free_cb = ALLOC_FN(_alloc), FREE_FN(_free)
    res = EgoResult()
    flags = 0
    if csr is not None:
        if csr not in ("none", "add"):
            raise ValueError("csr must be None, 'none' or 'add'")
        if int(radius) <= 4 and not base.has_self_loops():
            flags = FLAG_CSR | (FLAG_CSR_SELF_LOOPS if csr == "add" else 0)
    with torch.cuda.device(dev):
        status = L.mp_ego_expand(ptr(base.rowptr), ptr(base.col), N, ptr(cen), B, int(radius), flags, alloc_cb, free_cb,
                                 None, C.byref(res), _stream())
    held.clear()
    # (the ctypes callback objects and the closures they wrap form reference cycles: without this the `outs` dictionary —
    # and with it the output blocks — would stay alive until Python's cycle collector runs, long after the batch is dropped)
    tensors = dict(outs)
    outs.clear()
    del alloc_cb, free_cb
    outs = tensors
    if status != 0 and failure:
        raise EngineError(f"mp_ego_expand: allocation failed inside the expansion: {failure[0]!r}") from failure[0]
    check(status, "mp_ego_expand")
    n_out, e_out = int(res.n_nodes), int(res.n_edges)
    ei = outs[TAG_EDGES].view(torch.int64)[:2 * e_out].view(2, e_out)      # row 0 = source, row 1 = destination
    orig = outs[TAG_ORIG].view(torch.int64)[:n_out]
    ego_of = outs[TAG_EGO_OF].view(torch.int32)[:n_out]
    last_stats.clear()
    last_stats.update(nodes=n_out, edges=e_out, candidates=int(res.candidates),
                      scratch_peak_bytes=int(res.scratch_peak_bytes), centres=B)
    ids = torch.arange(B, device=dev)
    if csr is None:
        return ei, orig, ids, ego_of
    g = None
    if flags:
        nnz = int(res.nnz)
        g = CSRGraph(outs[TAG_ROWPTR].view(torch.int32)[:n_out + 1], outs[TAG_COL].view(torch.int32)[:nnz], None,
                     outs[TAG_EID].view(torch.int32)[:nnz], n_out, nnz)
        g.symmetric = True
        g._ego_ids = ((ids.data_ptr(), ids.numel(), ids._version), ids)      # id_branch(ids) takes the ego-batch shortcut
    return ei, orig, ids, ego_of, g

"""CSRGraph — the device-resident graph format of the engine.

One destination-sorted CSR per (batch, self-loop policy), built once on the GPU
from the batch's COO ``edge_index`` and cached; the reference instead re-derives
self-loops, degrees and the GCN normalisation from COO inside every layer call
(TfgIDLayer.py:500-503,546-558 with cache=None, main_zd.py:69-71;
idconv.py:165-173 with cached=False).

Row r holds the in-edges of destination r; ``col`` holds source ids.
"""
import collections
import ctypes as C
import itertools
import os
import threading
import weakref

import torch

from . import _lib
from ._lib import check, lib, ptr


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_hip(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.EngineError(
            f"{name} must live on a HIP device: the engine has no CPU path "
            "(the CPU oracle is test infrastructure only)")


BUILDS = collections.Counter()   # structures built so far, by kind: a step that runs on a PREPARED batch (CSRGraph.warm) adds none
BUILDS_BY_THREAD = collections.Counter()     # the same by building thread (the batch pipeline builds on a worker thread)


def _built(kind):
    BUILDS[kind] += 1
    BUILDS_BY_THREAD[threading.get_ident()] += 1


def builds_by_this_thread():
    return BUILDS_BY_THREAD[threading.get_ident()]


_HANDLES = weakref.WeakValueDictionary()     # int handle -> CSRGraph: how a graph crosses the torch.ops.mp.* boundary
_next_handle = itertools.count(1)


def from_handle(h):
    """the CSRGraph behind a handle passed to a torch.ops.mp.* operator"""
    try:
        return _HANDLES[int(h)]
    except KeyError:
        raise _lib.EngineError(f"graph handle {h} is not alive (the CSRGraph was garbage-collected)") from None


class CSRGraph:
    def __init__(self, rowptr, col, val, eid, num_nodes, nnz, num_cols=None):
        self.num_cols = int(num_nodes if num_cols is None else num_cols)   # != num_nodes for pooling operators
        self.rowptr = rowptr      # [N+1] int32
        self.col = col            # [nnz] int32 (view of a capacity-sized buffer)
        self.val = val            # [nnz] fp32 or None (= ones)
        self.eid = eid            # [nnz] int32 input position / -1-i for inserted loops, or None
        self.num_nodes = int(num_nodes)
        self.nnz = int(nnz)
        self._plan = None
        self._t = None            # transpose cache
        self._t_mean = None
        self._deg_cnt = None
        self.pos = None           # for a transposed graph: index into the source CSR
        self.dinv = None
        self.symmetric = False    # pattern AND values equal their transpose (ego batches of a symmetric graph): A^T is A

    # ---- construction ----------------------------------------------------
    @classmethod
    def from_edge_index(cls, edge_index, num_nodes, edge_weight=None, *, dst_row=1,
                        remove_self_loops=False, add_self_loops=False, keep_loop_weight=False,
                        fill=1.0, validate=True, num_cols=None):
        """edge_index: LongTensor [2, E] on the GPU.  dst_row=1 is PyG's
        source_to_target flow (edge_index[1] = destination i); dst_row=0 is the
        TF path's SparseAdj convention (edge_index[0] = row = destination).
        validate (default): node ids outside [0, num_nodes) raise ValueError, as indexing does in the reference,
        instead of becoming out-of-range column indices for the kernels (one E-sized pass; the CSR is cached per batch).
        num_cols: number of source nodes of a rectangular operator (pooling: rows = graphs, columns = nodes)."""
        _require_hip(edge_index, "edge_index")
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError("edge_index must be [2, E]")
        L = lib()
        dev = edge_index.device
        ei = edge_index.to(torch.int64)
        dst = ei[dst_row].contiguous()
        src = ei[1 - dst_row].contiguous()
        E, N = dst.numel(), int(num_nodes)
        w = None
        if edge_weight is not None:
            _require_hip(edge_weight, "edge_weight")
            w = edge_weight.detach().to(torch.float32).contiguous()
        flags = (_lib.COO_REMOVE_SELF_LOOPS if remove_self_loops else 0) | \
                (_lib.COO_ADD_SELF_LOOPS if add_self_loops else 0) | \
                (_lib.COO_KEEP_LOOP_WEIGHT if keep_loop_weight else 0)
        with torch.cuda.device(dev):
            NC = N if num_cols is None else int(num_cols)
            if NC != N and (remove_self_loops or add_self_loops):
                raise ValueError("self-loop edits need a square operator")
            if NC != N:
                flags |= _lib.COO_RECT            # column ids beyond the row count: the wide sort key
            if validate:
                bad = torch.zeros(2, dtype=torch.int32, device=dev)
                check(L.mp_check_edge_index(ptr(dst), ptr(dst), E, N, ptr(bad[0:1]), _stream()))
                check(L.mp_check_edge_index(ptr(src), ptr(src), E, NC, ptr(bad[1:2]), _stream()))
                if int(bad.sum().item()):
                    raise ValueError(f"edge_index has entries outside [0, {N}) x [0, {NC})")
            cap = E + (N if add_self_loops else 0)
            need = C.c_size_t(0)
            check(L.mp_csr_from_coo_ws_bytes(E, N, C.byref(need)))
            ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
            rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
            col = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
            weighted = w is not None or (add_self_loops and float(fill) != 1.0)
            val = torch.empty(max(cap, 1), dtype=torch.float32, device=dev) if weighted else None
            eid = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
            check(L.mp_csr_from_coo(ptr(dst), ptr(src), ptr(w), E, N, flags, float(fill), ptr(rowptr),
                                    ptr(col), ptr(val), ptr(eid), ptr(ws), need.value, _stream()),
                  "mp_csr_from_coo")
            nnz = int(rowptr[N].item())
        _built("csr")
        g = cls(rowptr, col[:nnz], None if val is None else val[:nnz], eid[:nnz], N, nnz, num_cols)
        return g

    @classmethod
    def from_csr(cls, rowptr, col, val, num_nodes):
        _require_hip(rowptr, "rowptr")
        return cls(rowptr.to(torch.int32).contiguous(), col.to(torch.int32).contiguous(),
                   None if val is None else val.to(torch.float32).contiguous(), None,
                   num_nodes, col.numel())

    def row_slice(self, r0, r1):
        """rows [r0, r1) as a rectangular operator over all columns (views of col / val, shifted rowptr):
        the local block of a destination-row partition"""
        e0, e1 = int(self.rowptr[r0].item()), int(self.rowptr[r1].item())
        rp = (self.rowptr[r0:r1 + 1] - e0).contiguous()
        return CSRGraph(rp, self.col[e0:e1], None if self.val is None else self.val[e0:e1], None,
                        r1 - r0, e1 - e0, self.num_cols)

    def select_rows(self, rows):
        """the rows `rows` (LongTensor, any order) as a rectangular operator [len(rows), num_cols];
        cached per index tensor (a batch's node_id_index is the same tensor every step)"""
        stamp = (rows.data_ptr(), rows.numel(), rows._version)
        cache = self.__dict__.setdefault("_row_subsets", {})
        if stamp not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[stamp] = (self._select_rows(rows), rows)    # holding `rows` keeps its address from being reused
        return cache[stamp][0]

    def _select_rows(self, rows):
        _built("row_subset")
        rows = rows.to(torch.int64)
        rp64 = self.rowptr.to(torch.int64)
        start = rp64.index_select(0, rows)
        deg = rp64.index_select(0, rows + 1) - start
        rp = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(deg, 0, out=rp[1:])
        total = int(rp[-1].item())
        idx = torch.arange(total, device=self.device) + torch.repeat_interleave(start - rp[:-1], deg)
        return CSRGraph(rp.to(torch.int32), self.col[idx].contiguous(),
                        None if self.val is None else self.val[idx].contiguous(), None,
                        rows.numel(), total, self.num_cols)

    def edge_operator(self):
        """[N, nnz] operator with one column per stored entry (entry e of row r -> column e, same value): reduces
        per-entry messages [nnz, d] by destination on the aggregation kernel (messages that are not x_j alone, e.g.
        x_j + edge_feature of generalconv.py:99-106)"""
        g = self.__dict__.get("_edge_op")
        if g is None:
            cols = torch.arange(self.nnz, dtype=torch.int32, device=self.device)
            g = CSRGraph(self.rowptr, cols, self.val, None, self.num_nodes, self.nnz, max(self.nnz, 1))
            self.__dict__["_edge_op"] = g
        return g

    def with_values(self, val):
        """same sparsity pattern (and plan / transpose pattern), other entry values"""
        g = CSRGraph(self.rowptr, self.col, val, self.eid, self.num_nodes, self.nnz, self.num_cols)
        g._plan = self._plan
        g._row_ids = getattr(self, "_row_ids", None)
        g._pattern_of = self if getattr(self, "_pattern_of", None) is None else self._pattern_of
        g._ego_ids = getattr(self, "_ego_ids", None)
        return g

    @property
    def device(self):
        return self.rowptr.device

    @property
    def handle(self):
        """an int naming this graph in torch.ops.mp.* calls (custom ops take tensors and scalars only); the
        registry holds weak references, the autograd context of an op keeps the graph itself"""
        h = self.__dict__.get("_handle")
        if h is None:
            h = next(_next_handle)
            self.__dict__["_handle"] = h
            _HANDLES[h] = self
        return h

    def variant(self, which):
        """0: this operator; 1: its transpose; 2: the transposed mean operator (backward of reduce='mean')"""
        return self if which == 0 else (self.transpose() if which == 1 else self.transpose_mean())

    # ---- plan --------------------------------------------------------------
    # segmentation tunables {seg_cost, row_cost, hub_deg, piece_edges} handed to mp_spmm_plan_build per call
    # (None = the library's defaults); tests shrink them to push small graphs through the hub path
    PLAN_CONFIG = None

    def plan(self):
        if self._plan is None:
            src = getattr(self, "_pattern_of", None)
            if src is not None and src is not self:
                self._plan = src.plan()       # same rowptr, same segmentation
                return self._plan
            L = lib()
            _built("plan")
            with torch.cuda.device(self.device):
                nb = C.c_size_t(0)
                cfg = None if CSRGraph.PLAN_CONFIG is None else (C.c_int32 * 4)(*CSRGraph.PLAN_CONFIG)
                check(L.mp_spmm_plan_bytes(self.num_nodes, self.nnz, cfg, C.byref(nb)))
                blob = torch.empty(nb.value // 4, dtype=torch.int32, device=self.device)
                counts = (C.c_int32 * 8)()
                check(L.mp_spmm_plan_build(ptr(self.rowptr), self.num_nodes, self.nnz, cfg, ptr(blob),
                                           nb.value, counts, _stream()), "mp_spmm_plan_build")
            self._plan = (blob, counts)
        return self._plan

    # ---- derived graphs ----------------------------------------------------
    def has_self_loops(self):
        """any stored entry with row == col (one device reduction + one host read, cached)"""
        cached = self.__dict__.get("_has_loops")
        if cached is None:
            cached = bool((self.row_ids() == self.col).any().item()) if self.nnz else False
            self.__dict__["_has_loops"] = cached
        return cached

    def transpose(self):
        """CSR of A^T (rows = sources) with values permuted; cached.  Graphs that share a sparsity
        pattern (with_values / gcn_norm) share one sorted transpose pattern and only permute values.
        A graph flagged `symmetric` is its own transpose."""
        if self.symmetric or (self._t is None and self.is_symmetric()):
            return self
        return self._transpose_sorted()

    def is_symmetric(self, run=None):
        """Does the stored operator equal its transpose, bit for bit?  Graphs flagged by their builder (ego batches) say
        yes at once.  run=True (or MP_SYM_CHECK=1 for every transpose() call) runs mp_csr_is_symmetric — one binary
        search per entry + one host read, cached; the reference's graphs are undirected, both directions stored
        (loader.py, transform.py:11-38), so the answer is usually yes and the backward pass then needs no second CSR.
        Off by default: the check costs what the sorted transpose costs (0.33 vs 0.33 ms at 6.6e6 entries, 8.8 vs 6.8 ms
        at 1.1e8 — random probes against three radix passes), so it buys memory (12 B per entry), not time."""
        if self.symmetric:
            return True
        known = self.__dict__.get("_sym_known")
        if known is None:
            known = False
            if run is None:
                run = os.environ.get("MP_SYM_CHECK", "0") == "1"
            if not run:
                return False                      # (not cached: a later explicit check may still run)
            if self.num_nodes == self.num_cols and self.nnz > 0:
                L = lib()
                _built("is_symmetric")
                flag = torch.empty(1, dtype=torch.int32, device=self.device)
                with torch.cuda.device(self.device):
                    check(L.mp_csr_is_symmetric(ptr(self.rowptr), ptr(self.col), ptr(self.val), self.num_nodes, self.nnz,
                                                ptr(flag), _stream()), "mp_csr_is_symmetric")
                known = int(flag.item()) == 0
            self.__dict__["_sym_known"] = known
            if known:
                self.symmetric = True
        return known

    def _transpose_sorted(self):
        if self._t is None:
            src = getattr(self, "_pattern_of", None)
            if src is not None and src is not self:
                t0 = src._transpose_sorted()
                t = CSRGraph(t0.rowptr, t0.col, None if self.val is None else self.val[t0.pos.long()], None,
                             t0.num_nodes, t0.nnz, t0.num_cols)
                t._plan = t0._plan if t0._plan is not None else None
                t.pos = t0.pos
                t._pattern_of = t0
                self._t = t
                return t
            L = lib()
            _built("transpose")
            R, N, nnz, dev = self.num_nodes, self.num_cols, self.nnz, self.device
            with torch.cuda.device(dev):
                nb = C.c_size_t(0)
                check(L.mp_csr_transpose_ws_bytes(nnz, N, C.byref(nb)))
                ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
                t_rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
                t_col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
                t_val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev) if self.val is not None else None
                pos = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
                check(L.mp_csr_transpose(ptr(self.rowptr), ptr(self.col), ptr(self.val), R, N, nnz,
                                         ptr(t_rowptr), ptr(t_col), ptr(t_val), ptr(pos), ptr(ws),
                                         nb.value, _stream()), "mp_csr_transpose")
            t = CSRGraph(t_rowptr, t_col[:nnz], None if t_val is None else t_val[:nnz], None, N, nnz, R)
            t.pos = pos[:nnz]
            self._t = t
        return self._t

    def transpose_with(self, val):
        """transpose pattern of this graph carrying other per-entry values"""
        t = self._transpose_sorted()           # (needs t.pos: also for a symmetric pattern, whose new values need not be)
        g = CSRGraph(t.rowptr, t.col, val[t.pos.long()] if val is not None else None, None,
                     t.num_nodes, t.nnz, t.num_cols)
        g._plan = t._plan
        g.pos = t.pos
        return g

    def row_ids(self):
        """row of every stored entry (cached: the attention kernels stream over it)"""
        if getattr(self, "_row_ids", None) is None:
            L = lib()
            _built("row_ids")
            out = torch.empty(max(self.nnz, 1), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                check(L.mp_csr_row_ids(ptr(self.rowptr), self.num_nodes, self.nnz, ptr(out), _stream()))
            self._row_ids = out[:self.nnz]
        return self._row_ids

    def entry_counts(self):
        """number of stored entries per row (the divisor of reduce='mean')"""
        if self._deg_cnt is None:
            self._deg_cnt = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float32)
        return self._deg_cnt

    def max_row_entries(self):
        """the longest row (one device reduction + one host read, cached): the one-kernel aggregate -> transform
        gives a row to the four waves of one workgroup, so extreme hubs are sent to the plan-based kernel"""
        cached = self.__dict__.get("_max_row")
        if cached is None:
            _built("max_row")
            cached = int((self.rowptr[1:] - self.rowptr[:-1]).max().item()) if self.num_nodes > 0 else 0
            self.__dict__["_max_row"] = cached
        return cached

    def transpose_mean(self):
        """A^T with entry (j <- i) = val / count(i): the backward operator of reduce='mean'"""
        if self._t_mean is None:
            inv = 1.0 / self.entry_counts().clamp(min=1.0)
            per_entry = inv[self.row_ids().long()]
            if self.val is not None:
                per_entry = per_entry * self.val
            self._t_mean = self.transpose_with(per_entry)
        return self._t_mean

    def degree(self, axis="row"):
        """weighted degree by destination ('row': SparseAdj.reduce_sum(axis=-1), sparse_adj.py:84-85) or by source
        ('col': scatter_add(edge_weight, edge_index[0]), idconv.py:56,144).  Both are row sums in a fixed order (the
        by-source sum runs over the cached transposed CSR): bitwise reproducible, no float atomics."""
        L = lib()
        if axis != "row":
            if self.num_cols != self.num_nodes:
                raise ValueError("degree('col') needs a square operator")
            return self.transpose().degree("row")
        deg = torch.empty(self.num_nodes, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(L.mp_csr_degree(ptr(self.rowptr), ptr(self.col), ptr(self.val), self.num_nodes, self.nnz,
                                  _lib.AXIS_ROW, ptr(deg), _stream()))
        return deg

    def warm(self, id_index=None, backward=True, mean=False, tiles=True):
        """Build and cache, NOW and on the current stream, every derived structure the aggregation launches through this
        operator ask for — longest row, segment plan, the transposed operator (+ its longest row and plan) for the
        backward pass, the identity-branch operators — so that the step which follows enqueues its launches without a
        single host synchronisation or build of its own (BUILDS stays put).  The batch pipeline
        (graphgym_amd/pipeline.py) calls this on a side stream, one batch ahead of the training step."""
        self.max_row_entries()
        self.plan()
        if id_index is not None:
            br = self.id_branch(id_index)
            br.t.max_row_entries()
            br.t.plan()
        if backward:
            t = self.transpose_mean() if mean else self.transpose()
            if t is not self:
                t.max_row_entries()
                t.plan()
        return self

    def gcn_norm(self, deg_axis="row"):
        """D^-1/2 A D^-1/2 on the stored entries (self-loops are a from_edge_index option).
        deg_axis='row': degree by destination (TfgIDLayer.py:549); 'col': by source (idconv.py:143-144)."""
        L = lib()
        _built("gcn_norm")
        N, nnz, dev = self.num_nodes, self.nnz, self.device
        if deg_axis != "row":
            # by-source degrees from the transposed CSR's row sums (deterministic), then one scaling pass
            deg = self.degree("col")
            dinv = deg.pow(-0.5)
            dinv = torch.where(torch.isfinite(dinv), dinv, torch.zeros_like(dinv))     # idconv.py:57-58,145-146
            g = self.scaled(row_scale=dinv, col_scale=dinv)
            g.dinv = dinv
            return g
        val_out = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)
        dinv = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.mp_gcn_norm_edges(ptr(self.rowptr), ptr(self.col), ptr(self.val), N, nnz,
                                      _lib.AXIS_ROW, ptr(val_out), ptr(dinv), _stream()), "mp_gcn_norm_edges")
        g = self.with_values(val_out[:nnz])
        g.dinv = dinv[:N]
        # D^-1/2 A D^-1/2 of a symmetric operator with the row degrees (= column degrees) on both sides is symmetric
        g.symmetric = self.is_symmetric()
        return g

    def scaled(self, row_scale=None, col_scale=None):
        """diag(row_scale) @ A @ diag(col_scale) on the stored entries (sparse_adj.py:110-119)"""
        L = lib()
        out = torch.empty(max(self.nnz, 1), dtype=torch.float32, device=self.device)
        rs = None if row_scale is None else row_scale.to(torch.float32).contiguous()
        cs = None if col_scale is None else col_scale.to(torch.float32).contiguous()
        with torch.cuda.device(self.device):
            check(L.mp_csr_scale_f32(ptr(self.rowptr), ptr(self.col), ptr(self.val), self.num_nodes, self.nnz,
                                     ptr(rs), ptr(cs), ptr(out), _stream()), "mp_csr_scale_f32")
        return self.with_values(out[:self.nnz])

    def mark_ids(self, id_index):
        """col with the sign bit set on entries whose source is an identity node; cached per index tensor
        (the mark depends only on the pattern, so graphs sharing a pattern share it)"""
        _require_hip(id_index, "id_index")
        owner = getattr(self, "_pattern_of", None) or self
        stamp = (id_index.data_ptr(), id_index.numel(), id_index._version)
        cache = owner.__dict__.setdefault("_id_marks", {})
        if stamp not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[stamp] = (self._mark_ids(id_index), id_index)   # holding the tensor keeps its address unique
        return cache[stamp][0]

    def id_branch(self, id_index):
        """The stored entries whose SOURCE is an identity node, as two compact operators (cached per index tensor):
        A_id = this operator restricted to the identity columns.  A S X W_id of gcn_id (TfgIDLayer.py:510-517) equals
        A_id Z with Z = X[id] W_id, so the identity branch is a small product plus these few entries.

        Returns an object with: rows [n_m] int32 (ascending rows that own such an entry), crp [n_m + 1] int32,
        slot [n_e] int32 (position of the entry's source in id_index), val [n_e] fp32 or None, defer [N] uint8
        (1 on `rows`), and t: A_id^T as a CSRGraph [n_id x N] (for the backward pass)."""
        _require_hip(id_index, "id_index")
        stamp = (id_index.data_ptr(), id_index.numel(), id_index._version)
        cache = self.__dict__.setdefault("_id_branch", {})
        if stamp not in cache:
            if len(cache) > 8:
                cache.clear()
            ego = getattr(self, "_ego_ids", None)
            if ego is not None and ego[0] == stamp and self.symmetric:
                cache[stamp] = (self._id_branch_of_ego_batch(id_index), id_index)
            else:
                cache[stamp] = (self._id_branch_build(id_index), id_index)
        return cache[stamp][0]

    def _id_branch_of_ego_batch(self, id_index):
        """id_branch for the CSR an ego expansion wrote (ego.ego_batch(csr=...)) with the batch's own identity nodes
        (the centres, new ids 0..B-1): every component holds exactly one identity node, whose id is the smallest of its
        component — so a row's entry from its centre, if it has one, is the row's FIRST entry, and by symmetry the
        transposed operator A_id^T is the block of the B centre rows.  A few gathers instead of the general build's
        passes over all entries (0.67 -> ~0.1 ms at 4096 centres)."""
        import types
        _built("id_branch")
        B, N, dev = id_index.numel(), self.num_nodes, self.device
        rp = self.rowptr.long()
        first = rp[:-1].clamp(max=max(self.nnz - 1, 0))
        src0 = self.col[first].long() if self.nnz else torch.zeros(N, dtype=torch.int64, device=dev)
        has = (rp[1:] > rp[:-1]) & (src0 < B)
        rows_m = torch.nonzero(has).view(-1)
        e_idx = first[rows_m]
        crp = torch.arange(rows_m.numel() + 1, dtype=torch.int32, device=dev)
        slot = src0[rows_m].to(torch.int32).contiguous()
        val = None if self.val is None else self.val[e_idx].contiguous()
        defer = has.to(torch.uint8)
        e1 = int(self.rowptr[B].item())
        t = CSRGraph(self.rowptr[:B + 1], self.col[:e1], None if self.val is None else self.val[:e1], None, B, e1,
                     self.num_cols)
        return types.SimpleNamespace(rows=rows_m.to(torch.int32).contiguous(), crp=crp, slot=slot, val=val,
                                     defer=defer, t=t, n_rows=int(rows_m.numel()))

    def _id_branch_build(self, id_index):
        import types
        _built("id_branch")
        dev = self.device
        ids = id_index.to(torch.int64)
        n_id, N = ids.numel(), self.num_nodes
        if n_id and (int(ids.min()) < 0 or int(ids.max()) >= self.num_cols):
            raise ValueError(f"id_index has entries outside [0, {self.num_cols})")
        if torch.unique(ids).numel() != n_id:
            raise ValueError("id_index has duplicate entries; the fused identity branch needs unique identity nodes "
                             "(use order='transform_first')")
        slot_of = torch.full((max(self.num_cols, 1),), -1, dtype=torch.int32, device=dev)
        slot_of[ids] = torch.arange(n_id, dtype=torch.int32, device=dev)
        s = slot_of[self.col.long()] if self.nnz else slot_of[:0]
        e_idx = torch.nonzero(s >= 0).view(-1)                      # ascending = CSR order
        rows_e = self.row_ids()[e_idx]
        rows_m, counts = torch.unique_consecutive(rows_e, return_counts=True)
        crp = torch.zeros(rows_m.numel() + 1, dtype=torch.int32, device=dev)
        crp[1:] = torch.cumsum(counts, 0)
        slot = s[e_idx].contiguous()
        val = None if self.val is None else self.val[e_idx].contiguous()
        defer = torch.zeros(max(N, 1), dtype=torch.uint8, device=dev)
        defer[rows_m.long()] = 1
        order = torch.sort(slot.long(), stable=True).indices
        t_rowptr = torch.zeros(n_id + 1, dtype=torch.int32, device=dev)
        t_rowptr[1:] = torch.cumsum(torch.bincount(slot.long(), minlength=n_id), 0)
        t = CSRGraph(t_rowptr, rows_e[order].contiguous().to(torch.int32),
                     None if val is None else val[order].contiguous(), None, n_id, int(e_idx.numel()), N)
        return types.SimpleNamespace(rows=rows_m.to(torch.int32).contiguous(), crp=crp, slot=slot, val=val,
                                     defer=defer, t=t, n_rows=int(rows_m.numel()))

    def _mark_ids(self, id_index):
        L = lib()
        _built("id_marks")
        ids = id_index.to(torch.int64).contiguous()
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.num_cols):
            raise ValueError(f"id_index has entries outside [0, {self.num_cols})")
        if torch.unique(ids).numel() != ids.numel():
            # the mark is a set: a node listed k times would count once here but k times in the reference's
            # tensor_scatter_nd_add / index_add_ (TfgIDLayer.py:515, idconv.py:155)
            raise ValueError("id_index has duplicate entries; the two-branch aggregation needs unique identity nodes "
                             "(use order='transform_first')")
        flag = torch.empty(max(self.num_cols, 1), dtype=torch.uint8, device=self.device)
        out = torch.empty(max(self.nnz, 1), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(L.mp_mark_id_sources(ptr(self.col), self.nnz, ptr(ids), ids.numel(), self.num_cols,
                                       ptr(flag), ptr(out), _stream()), "mp_mark_id_sources")
        return out[:self.nnz]

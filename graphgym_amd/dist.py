"""Multi-GPU execution of the path: one process per GPU, independent units per rank.

GraphGym batches are disjoint unions of graphs (graphgym/loader.py:247-251) and ID-GNN
ego nets are disjoint by construction (graphgym/models/transform.py:24-36), so the
aggregation itself needs no data-path collective: whole graphs / ego nets are dealt to
ranks by nnz (LPT), each rank builds its own CSR, and the only exchange of a training
step is the gradient all-reduce (RCCL over xGMI through torch.distributed's "nccl"
backend; "gloo" on CPU for the tests).  The reference has no distributed code at all.
"""
import os

import torch
import torch.distributed as dist


def lpt_partition(costs, world_size):
    """Greedy longest-processing-time assignment of units (cost = stored entries) to ranks.
    Returns a list of index lists, deterministic for equal inputs.  (The engine's host-side mp_lpt_partition_host
    computes the same assignment for the 10^4-10^5 units of a per-step global batch: lpt_owners.)"""
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    loads = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(costs[i])
    for p in parts:
        p.sort()
    return parts


def lpt_owners(costs, world_size):
    """the same assignment as lpt_partition as an owner array (numpy int32 [n]), through the engine's host function"""
    import ctypes as C
    import numpy as np
    from ._lib import check, lib
    c = np.ascontiguousarray(np.asarray(costs), dtype=np.int64)
    owner = np.empty(c.size, dtype=np.int32)
    check(lib().mp_lpt_partition_host(c.ctypes.data_as(C.c_void_p), c.size, int(world_size),
                                      owner.ctypes.data_as(C.c_void_p)), "mp_lpt_partition_host")
    return owner


def force_collectives():
    """MP_DIST_FORCE=1: run every collective even in a one-rank group.  A one-rank RCCL group executes the same code
    (communicator set-up, stream hand-off, the async work objects, reduce_scatter_tensor / all_gather_into_tensor) as an
    8-rank one, which is how the RCCL branches are exercised on the one GPU a test box has
    (tests/test_rccl_gpu.py, bench.py --mode step)."""
    return os.environ.get("MP_DIST_FORCE") == "1"


def init_from_env(device_type=None):
    """(rank, local_rank, world_size); initialises the process group when WORLD_SIZE > 1 (or MP_DIST_FORCE=1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type is None:
        device_type = "cuda" if torch.cuda.is_available() else "cpu"
    if device_type == "cuda":
        if os.environ.get("MP_SHARE_DEVICE") == "1":      # rehearsal of N ranks on a 1-GPU box (gloo only)
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("MP_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")
        # gloo and RCCL print banners to STDOUT (gloo when the group forms, RCCL when the first communicator is created);
        # a caller whose stdout is a protocol (bench.py: one JSON line) must not find them there: file descriptor 1 points
        # at stderr while the group forms and, under RCCL, while the first collective creates the communicator
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            dist.init_process_group(backend, rank=rank, world_size=world)
            if backend == "nccl":
                t = torch.zeros(1, device=torch.device("cuda", torch.cuda.current_device()))
                dist.all_reduce(t)
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return rank, local, world


def _coll_device(device):
    """collectives on scalars run on the GPU under RCCL and on the host under gloo"""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return device


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def all_reduce_max(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class GradBucket:
    """Flat fp32 gradient buckets, all-reduced once per step.

    The models on this path are small (0.2 M - 3.3 M parameters, <= 13 MB): the exchange is latency-bound over xGMI,
    so the point is to take it off the critical path, not to shrink it.  Two ways to use it:

    * ``attach()`` (the data-parallel step of bench.py --mode step): every ``p.grad`` becomes a VIEW into a bucket
      (no copy in or out), buckets follow reverse parameter order (the order backward produces gradients), and a
      post-accumulate hook launches a bucket's all-reduce asynchronously the moment its last gradient is written —
      the head's and the last layers' gradients travel while the first layers are still in backward.  ``finish()``
      waits and scales.  Use ``zero_grad()`` of the bucket (the views must survive).  ONE backward per finish():
      accumulating gradients over several backward passes needs the synchronous form below (in a single process,
      where nothing is exchanged, the hooks do nothing and accumulation works as usual).
    * ``all_reduce_mean()`` / ``all_reduce_sum()``: the synchronous form for an arbitrary loop (copies p.grad in and
      out).

    Averaging per-rank MEAN losses over ranks is a mean of means — not the full-batch gradient when shards hold
    different numbers of labelled nodes (LPT balances stored entries, not labels).  Normalise the loss by the GLOBAL
    count (``global_count``) and sum, as bench.py --mode step and tests/test_ddp_gpu.py do.
    The RCCL ("nccl") branches run under -m gpu in a one-rank group (tests/test_rccl_gpu.py: MP_DIST_FORCE=1) and the
    N > 1 logic under gloo with two ranks; no multi-GPU node was available to this build."""

    def __init__(self, params, n_buckets=1):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device if self.params else torch.device("cpu")
        order = list(reversed(self.params))                       # backward produces gradients roughly in this order
        n_buckets = max(1, min(int(n_buckets), len(order) or 1))
        total = sum(p.numel() for p in order)
        self.buckets = []                                          # (flat, [(param, view)])
        cur, cur_n, target = [], 0, (total + n_buckets - 1) // n_buckets
        groups = []
        for p in order:
            if cur and cur_n + p.numel() > target and len(groups) < n_buckets - 1:
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        for grp in groups:
            flat = torch.zeros(sum(p.numel() for p in grp), dtype=torch.float32, device=dev)
            views, off = [], 0
            for p in grp:
                views.append((p, flat[off:off + p.numel()].view_as(p)))
                off += p.numel()
            self.buckets.append((flat, views))
        self.flat = self.buckets[0][0] if len(self.buckets) == 1 else None     # (kept for callers of the one-bucket form)
        self._attached = False
        self._pending = []
        self._left = []
        self._hooks = []

    @staticmethod
    def _active():
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives())

    def _reduce(self, flat, async_op):
        if dist.get_backend() == "gloo" and flat.is_cuda:   # rehearsal mode (ranks sharing a GPU): stage through the host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
            return None
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)

    # ---- overlapped form ---------------------------------------------------------------
    def attach(self):
        """make every p.grad a view of its bucket and launch a bucket's all-reduce from the hook of its last gradient"""
        if self._attached:
            return self
        for bi, (flat, views) in enumerate(self.buckets):
            for p, v in views:
                p.grad = v
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self._attached = True
        self._reset()
        return self

    def _reset(self):
        self._left = [len(views) for _, views in self.buckets]
        self._pending = []

    def _check_views(self, bi=None):
        """p.grad must still BE the bucket view: optimizer.zero_grad() with set_to_none=True (torch's default) drops it,
        autograd then allocates a fresh p.grad, and the all-reduce would exchange a bucket nobody wrote — replicas
        would diverge silently.  Use bucket.zero_grad() (or optimizer.zero_grad(set_to_none=False))."""
        for i, (_, views) in enumerate(self.buckets):
            if bi is not None and i != bi:
                continue
            for p, v in views:
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    raise RuntimeError("GradBucket: a parameter's .grad is no longer the bucket view (was "
                                       "optimizer.zero_grad(set_to_none=True) called?); use GradBucket.zero_grad() "
                                       "or re-attach")

    def _make_hook(self, bi):
        def hook(param):
            if not self._active():
                return            # a single process exchanges nothing: plain gradient accumulation keeps working
            if self._left[bi] <= 0:
                raise RuntimeError("GradBucket: a gradient arrived for a bucket whose all-reduce was already launched "
                                   "(a second backward before finish()); call finish() / zero_grad() between steps — "
                                   "gradient accumulation over several backward passes needs the synchronous form "
                                   "(all_reduce_mean / all_reduce_sum after the last backward)")
            self._left[bi] -= 1
            if self._left[bi] == 0:
                self._check_views(bi)
                flat = self.buckets[bi][0]
                if flat.is_cuda and dist.get_backend() != "gloo":
                    # the gradient kernels run on the current stream; the collective is enqueued behind them by
                    # torch.distributed's own stream hand-off and overlaps the backward kernels that follow
                    self._pending.append(self._reduce(flat, async_op=True))
                else:
                    self._pending.append(self._reduce(flat, async_op=False))
        return hook

    def zero_grad(self):
        for flat, _ in self.buckets:
            flat.zero_()
        self._reset()

    def finish(self, scale=1.0):
        """wait for the launched all-reduces (launch any bucket whose hooks did not all fire: parameters unused this
        step), then multiply by `scale` (1 / world for a mean of per-rank gradients; 1 for globally normalised losses)"""
        if self._active():
            if self._attached:
                self._check_views()
            for bi, left in enumerate(self._left):
                if left > 0:
                    self._pending.append(self._reduce(self.buckets[bi][0], async_op=False))
            for w in self._pending:
                if w is not None:
                    w.wait()
        if scale != 1.0:
            for flat, _ in self.buckets:
                flat.mul_(scale)
        self._reset()

    # ---- synchronous form ----------------------------------------------------------------
    def _sync_reduce(self, scale):
        if not self._active():
            return
        for flat, views in self.buckets:
            if not self._attached:
                for p, v in views:
                    if p.grad is None:
                        v.zero_()
                    else:
                        v.copy_(p.grad)
            self._reduce(flat, async_op=False)
            if scale != 1.0:
                flat.mul_(scale)
            if not self._attached:
                for p, v in views:
                    if p.grad is None:
                        p.grad = v.clone()
                    else:
                        p.grad.copy_(v)

    def all_reduce_mean(self):
        """average gradients over ranks in place (no-op for a single process)"""
        self._sync_reduce(1.0 / dist.get_world_size() if self._active() else 1.0)

    def all_reduce_sum(self):
        """sum gradients over ranks in place: the exact full-batch gradient when every rank's loss is normalised by
        the global count (global_count)"""
        self._sync_reduce(1.0)


def global_count(local_count, device):
    """sum of a per-rank count over ranks (e.g. labelled nodes), as a float"""
    return all_reduce_sum(local_count, device)


# ---- one graph partitioned by destination rows ("one giant graph") ----------------
class RowPartition:
    """Contiguous, nnz-balanced destination-row ranges of one CSRGraph, one per rank.

    Rank p owns rows R_p of the adjacency and of every hidden matrix H.  An aggregation needs source
    rows it does not own (on a scale-free graph: nearly all of them), so each layer exchanges
    features: all-gather of H before the local aggregation, reduce-scatter of dH in backward.  At the
    headline size that is 8.96 GB received per GPU per layer over xGMI (>= 8.4 ms at 7 x 153 GB/s)
    against 2.6 ms of local aggregation — the step is xGMI-bound, which is why the independent-unit
    sharding above is the scaling mode of record."""

    def __init__(self, graph, rank=None, world=None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        n = graph.num_nodes
        targets = torch.arange(1, self.world, device=graph.device, dtype=torch.int64) * graph.nnz // self.world
        cuts = torch.searchsorted(graph.rowptr.to(torch.int64), targets).clamp(max=n).tolist()
        self.bounds = [0] + cuts + [n]
        for i in range(1, len(self.bounds)):            # keep the ranges monotone
            self.bounds[i] = max(self.bounds[i], self.bounds[i - 1])
        self.num_nodes = n
        self.local = graph.row_slice(self.bounds[self.rank], self.bounds[self.rank + 1])
        self.max_rows = max(self.bounds[i + 1] - self.bounds[i] for i in range(self.world))

    @property
    def rows(self):
        return self.bounds[self.rank], self.bounds[self.rank + 1]


def pad_rows(h_loc, max_rows):
    """[rows, d] -> [max_rows, d], zero rows appended (equal chunks for all_gather_into_tensor)"""
    if h_loc.size(0) == max_rows:
        return h_loc.contiguous()
    pad = torch.zeros((max_rows, h_loc.size(1)), dtype=h_loc.dtype, device=h_loc.device)
    pad[:h_loc.size(0)] = h_loc
    return pad


def unpad_gathered(out, bounds, max_rows):
    """[world * max_rows, d] of padded chunks -> [N, d]: chunk p contributes its first bounds[p+1] - bounds[p] rows"""
    world = len(bounds) - 1
    if all(bounds[p + 1] - bounds[p] == max_rows for p in range(world)):
        return out
    return torch.cat([out[p * max_rows: p * max_rows + bounds[p + 1] - bounds[p]] for p in range(world)], dim=0)


def pad_chunks(full, bounds, max_rows):
    """[N, d] -> [world, max_rows, d]: rows bounds[p]..bounds[p+1] in chunk p, zero rows behind them (equal chunks for
    reduce_scatter_tensor)"""
    world = len(bounds) - 1
    inp = torch.zeros((world, max_rows, full.size(1)), dtype=full.dtype, device=full.device)
    for p in range(world):
        a, b = bounds[p], bounds[p + 1]
        inp[p, :b - a] = full[a:b]
    return inp


def _gather_rows(part, h_loc):
    """[rows_p, d] on every rank -> [N, d] everywhere (all-gather with padding to the widest range; the index arithmetic
    is pad_rows / unpad_gathered, tested on CPU with ragged bounds in tests/test_host_logic.py)"""
    if part.world == 1 and not (force_collectives() and dist.is_initialized()):
        return h_loc
    d = h_loc.size(1)
    pad = pad_rows(h_loc, part.max_rows)
    if dist.get_backend() == "gloo":
        host = pad.is_cuda
        src = pad.cpu() if host else pad
        out = torch.empty((part.world * part.max_rows, d), dtype=h_loc.dtype)
        dist.all_gather_into_tensor(out, src)
        return unpad_gathered(out, part.bounds, part.max_rows).to(h_loc.device)
    # RCCL: one all_gather_into_tensor of equal (padded) chunks straight into the [world * max_rows, d] result
    out = torch.empty((part.world * part.max_rows, d), dtype=h_loc.dtype, device=h_loc.device)
    dist.all_gather_into_tensor(out, pad)
    return unpad_gathered(out, part.bounds, part.max_rows)


def _scatter_sum_rows(part, full):
    """sum of [N, d] partials over ranks, each rank keeping its own rows (reduce-scatter; pad_chunks lays out the equal
    chunks reduce_scatter_tensor wants)"""
    if part.world == 1 and not (force_collectives() and dist.is_initialized()):
        return full
    r0, r1 = part.rows
    d = full.size(1)
    if dist.get_backend() == "gloo":             # gloo has no reduce_scatter_tensor: all-reduce of the same padded layout
        host = full.is_cuda
        inp = pad_chunks(full.cpu() if host else full, part.bounds, part.max_rows)
        dist.all_reduce(inp, op=dist.ReduceOp.SUM)
        return inp[part.rank, :r1 - r0].to(full.device)
    # RCCL: every rank receives only its own rows (1 / world of the all-reduce's traffic)
    inp = pad_chunks(full, part.bounds, part.max_rows)
    out = torch.empty((part.max_rows, d), dtype=full.dtype, device=full.device)
    dist.reduce_scatter_tensor(out, inp.view(-1, d), op=dist.ReduceOp.SUM)
    return out[:r1 - r0]


class _HaloAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h_loc, part, reduce):
        from . import _lib as L_, ops
        h_full = _gather_rows(part, h_loc.contiguous())
        z_loc, _ = ops._raw_spmm(part.local, h_full, L_.REDUCE[reduce])
        ctx.part, ctx.reduce = part, reduce
        return z_loc

    @staticmethod
    def backward(ctx, dz_loc):
        from . import _lib as L_, ops
        part = ctx.part
        gt = part.local.transpose() if ctx.reduce != "mean" else part.local.transpose_mean()
        partial, _ = ops._raw_spmm(gt, dz_loc.contiguous(), L_.SUM)       # [N, d]: this rank's rows' contribution
        return _scatter_sum_rows(part, partial), None, None


def halo_aggregate(part, h_loc, reduce="sum"):
    """z[R_p] = reduce_j A[R_p, j] h[j] with h row-partitioned like the output ('sum' / 'mean')"""
    if reduce not in ("sum", "add", "mean"):
        raise ValueError("row-partitioned aggregation supports sum and mean")
    return _HaloAggregate.apply(h_loc, part, "sum" if reduce == "add" else reduce)

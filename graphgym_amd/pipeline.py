"""A fresh ego-net batch every step, built one batch AHEAD of the training step on a second HIP stream.

The reference draws a new batch every iteration (graphgym/train.py:18-25,39: `for batch in loader`) and builds it on the
host (graphgym/models/transform.py:11-38, one Python loop per node) while the device waits.  Here the whole build of
batch k + 1 — ego expansion (csrc/ego.hip), feature / label gather, COO -> CSR, GCN normalisation, segment plans, the
transposed operator for the backward pass, the identity-branch operators (CSRGraph.warm) — runs on a SIDE stream while
the training step of batch k runs on the main stream, so the step finds everything cached on the batch holder and
enqueues without building or synchronising anything itself.

The loop:   pipe.submit(c0, y0)
            for k: b = pipe.get(); step(b); pipe.done(); pipe.submit(c[k+1], y[k+1])
(a caller may run further ahead — submit d batches before the loop, then one per step: batches are handed out in the
order they were submitted, and a build then has d steps to finish in)
(the step is enqueued FIRST, then the next build is started: the host blocks only in the build's size reads, on the side
stream, while the main stream works through the step it already holds).

Memory discipline (torch's caching allocator is per stream): everything a batch owns is allocated on the side stream and
read by the main stream.  A batch's memory may be handed out again (to a later build, on the side stream) only behind
the event recorded after ITS step on the main stream.  The pipeline enforces that itself: it keeps a reference to every
batch it hands out; when build k + 1 is started (after step k was enqueued) the side stream first waits for the event
behind step k - 1 (`done()`; the event BEFORE the last one — waiting for step k itself would serialise build and step)
and only then are the batches up to k - 1 released; batch k stays alive until build k + 2 starts, whatever the caller does
with its own reference.  The main stream waits for a batch's `built` event before its first launch.  Centres and labels arrive as HOST tensors and are uploaded on the side stream (an upload on the main
stream would sit behind the queued step).  No record_stream bookkeeping, no device-wide synchronisation."""
import collections
import contextlib
import gc
import os
import time
import types

import torch

from . import graph as G
from .ego import ego_batch
from . import ego as _ego


class EgoBatch(types.SimpleNamespace):
    """x [n, F], edge_index [2, E], ids [B], y [B], holder (the per-batch graph cache), built (event), stats"""


class EgoBatchPipeline:
    def __init__(self, base, features, radius, prepare=None, device=None, threaded=True, csr=None):
        """base: CSRGraph of the (symmetric) base graph; features: [N, F] node features of the base graph;
        prepare(inputs, holder): builds the model's per-batch graph structures (e.g. TfgNodeModel.prepare);
        threaded: the build runs on a worker THREAD as well as on its own stream — the step's launches are host work too
        (an ID-GCN step on a 2 * 10^6-node batch is ~150 launches), and one Python thread would enqueue step and build one
        after the other; the worker spends most of its time inside ctypes / torch calls and size reads, which release the
        interpreter lock."""
        self.base, self.features, self.radius, self.prepare = base, features, int(radius), prepare
        self.csr = csr      # "none" | "add": let the expansion write the batch's CSR itself (ego.ego_batch(csr=...))
        self.device = device if device is not None else base.device
        # a HIGH-PRIORITY stream: the build is many short launches separated by size reads; queued at normal priority
        # behind the step's long HBM-bound launches every one of those reads waits for a slot (build 7.5 ms alone, ~18 ms
        # beside a step), at high priority its launches take the next free compute units
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        # The stream is this pipeline's own and lives as long as it does: keep ONE pipeline open for a whole run (the
        # caching allocator keeps a pool per stream: the first batches of a new stream go to the driver for their memory,
        # later ones find it cached — bench_step warms and times the same pipeline).  One process-wide stream for every
        # pipeline was measured and lost the overlap (20.1 against 16.3 ms per ID-GCN step when the stream had been created
        # before the process group's first collective; cause not established) — not done.
        self.side = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("MP_PIPE_PRIORITY", hi)))
        self._done = [None, None]          # events behind the last two steps on the main stream
        self._held = []                    # the batches handed out last: the pipeline keeps them alive until it is safe
        self._pending = collections.deque()  # submitted, not yet handed out (more than one: the caller runs ahead by `depth`)
        self._first = True
        self._pool = None
        # a batch's buffers live for one step: no (read, output) pair of the aggregation launches is ever seen twice, so
        # placement checks (graphgym_amd/placement.py) could only cost — they are off while a pipeline is open
        from . import placement
        placement.pause()
        self._paused = True
        if threaded:
            import concurrent.futures
            self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="mp-batch")

    def submit(self, centres, labels):
        """start building the batch around `centres` (LongTensor [B], HOST) with labels `labels` ([B], HOST) on the side
        stream (and, threaded, on the worker thread: submit returns at once)"""
        main = torch.cuda.current_stream(self.device)
        if self._first:
            self.side.wait_stream(main)                    # the base graph / features are ready
            self._first = False
        if self._done[0] is not None:
            self.side.wait_event(self._done[0])            # the step before last: its batch's memory may be reused now
        # Only now may the batches before the last one die: what the side stream enqueues from here on runs behind the
        # event of their last step.  The LAST batch handed out stays alive inside the pipeline whatever the caller does
        # with its own reference — its step may still be running on the main stream while this build allocates (on a
        # worker thread the build's allocations interleave with the caller's `del batch`: a block freed too early was
        # handed to the build and overwritten under the running step — a GPU memory fault, found the hard way).
        del self._held[:-1]
        if self._pool is not None and self.side != main:
            job = self._pool.submit(self._build, centres, labels)
        else:
            job = self._build(centres, labels)
        self._pending.append(job)
        return job

    def _build(self, centres, labels):
        torch.cuda.set_device(self.device)
        timing = {} if os.environ.get("MP_PIPE_TIMING") == "1" else None     # (study: wall time per phase, side stream drained)

        def lap(name, t0):
            if timing is None:
                return t0
            self.side.synchronize()
            t1 = time.perf_counter()
            timing[name] = (t1 - t0) * 1e3
            return t1

        with torch.cuda.stream(self.side):
            from .harness import Batch
            t = lap("wait", time.perf_counter())
            centres = centres.to(self.device, non_blocking=True)
            labels = labels.to(self.device, non_blocking=True)
            t = lap("upload", t)
            g = None
            if self.csr is not None:
                ei, orig, ids, _, g = ego_batch(self.base, centres, self.radius, csr=self.csr)
            else:
                ei, orig, ids, _ = ego_batch(self.base, centres, self.radius)
            t = lap("ego", t)
            x = self.features.index_select(0, orig)
            t = lap("features", t)
            holder = Batch()
            if g is not None:
                from .layers import seed_graph_cache
                seed_graph_cache(holder, ei, int(orig.numel()), g, self.csr)
            inputs = [x, ei, ids]
            prepared = bool(self.prepare(inputs, holder)) if self.prepare is not None else False
            t = lap("prepare", t)
            built = torch.cuda.Event()
            built.record(self.side)
        return EgoBatch(x=x, edge_index=ei, ids=ids, y=labels, holder=holder, built=built, prepared=prepared,
                        nodes=int(orig.numel()), edges=int(ei.size(1)), ego_stats=dict(_ego.last_stats), timing=timing)

    def get(self):
        """the oldest submitted batch, usable on the current (main) stream"""
        b = self._pending.popleft()
        if hasattr(b, "result"):
            b = b.result()                                 # (re-raises what the worker raised)
        torch.cuda.current_stream(self.device).wait_event(b.built)
        self._held.append(b)
        return b

    def close(self):
        for job in self._pending:                          # builds nobody asked for: let them finish before the streams go
            if hasattr(job, "result"):
                job.result()
        self._pending.clear()
        self.side.synchronize()
        torch.cuda.current_stream(self.device).synchronize()
        self._held.clear()
        if self._paused:
            from . import placement
            placement.resume()
            self._paused = False
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def done(self, batch=None):
        """call right after the step of a batch has been enqueued on the main stream"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._done = [self._done[1], ev]


def fit_allocator_to_changing_shapes(divisions=8):
    """Every fresh batch has its own sizes, and torch's caching allocator hands a cached block only to a request it
    covers: with exact sizes the pools keep missing and the allocator keeps going to the driver (hipMalloc is synchronous
    and stalls the device: ~1 per step after 100 batches, 53 GB reserved for a 6 GB working set — bench_step's
    `driver_allocs_in_fresh_steps`).  Rounding large requests up to 1/`divisions` steps between powers of two
    (`roundup_power2_divisions`) makes batches of similar size share blocks.  Returns what was set (None: the allocator
    does not take settings at run time)."""
    conf = f"roundup_power2_divisions:{int(divisions)}"
    if os.environ.get("MP_KEEP_ALLOCATOR") == "1":
        return None
    for setter in (getattr(torch._C, "_accelerator_setAllocatorSettings", None),
                   getattr(torch.cuda.memory, "_set_allocator_settings", None)):
        if setter is None:
            continue
        try:
            setter(conf)
            return conf
        except Exception:                                  # (an allocator backend without run-time settings)
            continue
    return None


def builds_snapshot():
    return dict(G.BUILDS)


@contextlib.contextmanager
def quiet_gc():
    """Keep Python's cycle collector out of the steps.  A full collection of a torch process's heap takes 40-80 ms
    (measured: every ~3rd batch build of a 10 ms loop carried one — scripts/pipe_probe.py), as long as four training
    steps.  Inside this context automatic collection is off and everything alive at entry is frozen (never traversed
    again); the loop calls the yielded `tick()` once per step, which collects the young generation only (~0.1 ms) and
    the middle one every 16th call.  What a step leaves behind in reference cycles (autograd graphs, ctypes callbacks)
    is reclaimed by these; tensors themselves are freed by reference counting and never wait for the collector."""
    gc.collect()
    gc.freeze()
    was = gc.isenabled()
    gc.disable()
    n = [0]

    def tick():
        n[0] += 1
        gc.collect(1 if n[0] % 16 == 0 else 0)
    try:
        yield tick
    finally:
        if was:
            gc.enable()
        gc.unfreeze()

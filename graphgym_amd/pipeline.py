"""A fresh ego-net batch every step, built one batch AHEAD of the training step on a second HIP stream.

The reference draws a new batch every iteration (graphgym/train.py:18-25,39: `for batch in loader`) and builds it on the
host (graphgym/models/transform.py:11-38, one Python loop per node) while the device waits.  Here the whole build of
batch k + 1 — ego expansion (csrc/ego.hip), feature / label gather, COO -> CSR, GCN normalisation, segment plans, the
transposed operator for the backward pass, the identity-branch operators (CSRGraph.warm) — runs on a SIDE stream while
the training step of batch k runs on the main stream, so the step finds everything cached on the batch holder and
enqueues without building or synchronising anything itself.

The loop:   pipe.submit(c0, y0)
            for k: b = pipe.get(); step(b); pipe.done(); pipe.submit(c[k+1], y[k+1])
(the step is enqueued FIRST, then the next build is started: the host blocks only in the build's size reads, on the side
stream, while the main stream works through the step it already holds).

Memory discipline (torch's caching allocator is per stream): everything a batch owns is allocated on the side stream and
read by the main stream.  Batch k - 1's memory goes back to the side stream's pool when the caller lets go of it — at
`get()` of batch k at the earliest — and may be handed out again while batch k + 1 is built, so that build's side-stream
work first waits for the event recorded behind step k - 1 on the main stream (`done()`; the event BEFORE the last one:
waiting for step k itself would serialise build and step), and the main stream waits for a batch's `built` event before
its first launch.  Centres and labels arrive as HOST tensors and are uploaded on the side stream (an upload on the main
stream would sit behind the queued step).  No record_stream bookkeeping, no device-wide synchronisation."""
import types

import torch

from . import graph as G
from .ego import ego_batch
from . import ego as _ego


class EgoBatch(types.SimpleNamespace):
    """x [n, F], edge_index [2, E], ids [B], y [B], holder (the per-batch graph cache), built (event), stats"""


class EgoBatchPipeline:
    def __init__(self, base, features, radius, prepare=None, device=None):
        """base: CSRGraph of the (symmetric) base graph; features: [N, F] node features of the base graph;
        prepare(inputs, holder): builds the model's per-batch graph structures (e.g. TfgNodeModel.prepare)"""
        self.base, self.features, self.radius, self.prepare = base, features, int(radius), prepare
        self.device = device if device is not None else base.device
        self.side = torch.cuda.Stream(device=self.device)
        self._done = [None, None]          # events behind the last two steps on the main stream
        self._pending = None
        self._first = True

    def submit(self, centres, labels):
        """start building the batch around `centres` (LongTensor [B], HOST) with labels `labels` ([B], HOST) on the side
        stream"""
        main = torch.cuda.current_stream(self.device)
        if self._first:
            self.side.wait_stream(main)                    # the base graph / features are ready
            self._first = False
        if self._done[0] is not None:
            self.side.wait_event(self._done[0])            # the step before last: its batch's memory may be reused now
        with torch.cuda.stream(self.side):
            from .harness import Batch
            centres = centres.to(self.device, non_blocking=True)
            labels = labels.to(self.device, non_blocking=True)
            ei, orig, ids, _ = ego_batch(self.base, centres, self.radius)
            x = self.features.index_select(0, orig)
            holder = Batch()
            inputs = [x, ei, ids]
            prepared = bool(self.prepare(inputs, holder)) if self.prepare is not None else False
            built = torch.cuda.Event()
            built.record(self.side)
        self._pending = EgoBatch(x=x, edge_index=ei, ids=ids, y=labels, holder=holder, built=built, prepared=prepared,
                                 nodes=int(orig.numel()), edges=int(ei.size(1)), ego_stats=dict(_ego.last_stats))
        return self._pending

    def get(self):
        """the submitted batch, usable on the current (main) stream"""
        b, self._pending = self._pending, None
        torch.cuda.current_stream(self.device).wait_event(b.built)
        return b

    def done(self, batch=None):
        """call right after the step of a batch has been enqueued on the main stream"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._done = [self._done[1], ev]


def builds_snapshot():
    return dict(G.BUILDS)

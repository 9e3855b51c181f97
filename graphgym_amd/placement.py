"""Engine-owned placement of the large matrices of the aggregation path.

On MI355X one HBM-bound launch that reads X and writes Y is up to ~15 % slower depending on which
physical memory backs the two (measured map: profiles/r02_placement_map.log).  torch's
allocator cannot be steered, so large outputs of the engine's operators come from an arena the engine
owns (csrc/arena.hip):

  * the arena is one slab of device memory, created on the first large request;
  * ``calibrate`` times a streaming copy between a 1 GiB chunk of every 4 GiB granule and every other
    (mp_probe_copy_ms; about a second, once per process) -> a symmetric matrix of relative slow-downs;
  * ``empty(shape, reads=[x, ...])`` prices every granule by how much it conflicts with the granules the
    launch reads and takes the cheapest free range (mp_arena_alloc_placed);
  * the buffer becomes an ordinary torch tensor through DLPack; when its last reference dies the deleter
    gives the range back to the arena.  Engine operators enqueue on torch's current stream, so a range
    reused by a later operator is ordered behind the earlier one's kernels.

Tensors the engine did not allocate (a batch's input features) are priced by probing them against the free
part of the arena once (cached per buffer).  Small outputs (< ``MIN_BYTES``) stay with torch: they live in
L2 / Infinity Cache and placement does not matter.

Environment: MP_PLACEMENT=off disables the arena; MP_ARENA_GB sets its size (default: 50 % of the free
device memory at creation, at most 160 GiB); MP_PLACE_MIN_MB the size threshold (default 1024).
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from ._lib import check, lib

GiB = 1 << 30
GRANULE = 4 * GiB          # resolution of the conflict map
CHUNK = 1 * GiB            # bytes copied per calibration probe (well past the 256 MiB Infinity Cache)
FOREIGN_CHUNK = 512 << 20  # per probe of a tensor the engine did not allocate
CALIBRATION_TRIALS = 3     # timed copies per pair of granules (the fastest counts)
VERIFY_CHUNK = 512 << 20   # bytes per timed copy when candidate positions are verified
MIN_BYTES = int(os.environ.get("MP_PLACE_MIN_MB", "1024")) << 20

_lock = threading.Lock()
_arenas = {}               # device index -> Arena | False (creation failed / disabled)
_CAPSULE_NAME = b"dltensor"

_PyCapsule_New = C.pythonapi.PyCapsule_New
_PyCapsule_New.restype = C.py_object
_PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]

_DL_TYPES = {torch.float32: (2, 32, 4), torch.int32: (0, 32, 4), torch.uint8: (1, 8, 1), torch.int64: (0, 64, 8),
             torch.float64: (2, 64, 8)}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _probe(src, dst, nbytes, reps=2):
    ms = C.c_float(0)
    check(lib().mp_probe_copy_ms(C.c_void_p(src), C.c_void_p(dst), nbytes, reps, C.byref(ms), _stream()),
          "mp_probe_copy_ms")
    return ms.value


class Arena:
    def __init__(self, nbytes):
        L = lib()
        check(L.mp_arena_create(nbytes), "mp_arena_create")
        base, nb = C.c_void_p(), C.c_size_t()
        check(L.mp_arena_info(C.byref(base), C.byref(nb), None, None))
        self.base, self.bytes = base.value, nb.value
        self.n_gran = (self.bytes + GRANULE - 1) // GRANULE
        self.conflict = np.zeros((self.n_gran, self.n_gran), dtype=np.float32)
        self.t_min_ms = None
        self._foreign = {}
        self.calibrate()

    # ---- the conflict map ---------------------------------------------------------------
    def calibrate(self):
        """conflict[g, h] = t(copy between granule g and granule h) / t_min - 1, symmetric.  Runs on an empty
        arena (the probes write into it)."""
        n = self.n_gran
        chunk = CHUNK
        T = np.full((n, n), np.inf, dtype=np.float64)
        torch.cuda.synchronize()
        for _ in range(CALIBRATION_TRIALS):      # interference only ever adds time: keep the fastest trial per cell
            for g in range(n):
                for h in range(g, n):
                    src = self.base + g * GRANULE
                    dst = self.base + h * GRANULE + (chunk if g == h else 0)      # same granule: its second chunk
                    t = _probe(src, dst, chunk, 1)
                    T[g, h] = T[h, g] = min(T[g, h], t)
        tmin = np.nanmin(T)
        M = T / tmin - 1.0
        M[np.isnan(M)] = np.nanmax(M)
        self.conflict = M.astype(np.float32)
        self.t_min_ms = float(tmin)
        self.chunk = chunk

    # ---- pricing ------------------------------------------------------------------------
    def owns(self, t):
        p = t.data_ptr()
        return self.base <= p < self.base + self.bytes

    def _footprint(self, ptr, nbytes):
        """fraction of [ptr, ptr + nbytes) in every granule"""
        f = np.zeros(self.n_gran, dtype=np.float64)
        off, end = ptr - self.base, ptr - self.base + nbytes
        g = off // GRANULE
        while off < end and g < self.n_gran:
            stop = min((g + 1) * GRANULE, end)
            f[g] += (stop - off) / nbytes
            off = stop
            g += 1
        return f

    def price_row(self, t):
        """penalty of writing into each granule while `t` is read: [n_gran] float"""
        nbytes = t.numel() * t.element_size()
        if nbytes == 0:
            return np.zeros(self.n_gran)
        if self.owns(t):
            return self._footprint(t.data_ptr(), nbytes) @ self.conflict
        if nbytes < FOREIGN_CHUNK:
            return np.zeros(self.n_gran)
        key = (t.data_ptr(), nbytes)
        row = self._foreign.get(key)
        if row is None:
            row = self._probe_foreign(t.data_ptr(), nbytes)
            if len(self._foreign) > 64:
                self._foreign.clear()
            self._foreign[key] = row
        return row

    def _probe_foreign(self, ptr, nbytes):
        """read sample chunks of a buffer outside the arena, write into a free chunk of every granule"""
        L = lib()
        n = self.n_gran
        k = max(1, min(4, nbytes // GRANULE + 1))
        step = (nbytes - FOREIGN_CHUNK) // max(k - 1, 1) if k > 1 else 0
        samples = [ptr + (i * step) // 256 * 256 for i in range(k)]
        T = np.full((k, n), np.nan)
        torch.cuda.synchronize()
        for h in range(n):
            pen = np.ones(n, dtype=np.float32)
            pen[h] = 0.0
            out = C.c_void_p()
            st = L.mp_arena_alloc_placed(FOREIGN_CHUNK, pen.ctypes.data_as(C.c_void_p), n, GRANULE, C.byref(out))
            if st != 0:
                continue
            try:
                off = out.value - self.base
                if off // GRANULE == h and (off + FOREIGN_CHUNK - 1) // GRANULE == h:
                    for i, s in enumerate(samples):
                        T[i, h] = _probe(s, out.value, FOREIGN_CHUNK)
            finally:
                L.mp_arena_release(out)
        if np.all(np.isnan(T)):
            return np.zeros(n)
        tmin = np.nanmin(T)
        have = ~np.all(np.isnan(T), axis=0)
        row = np.full(n, np.nan)
        row[have] = np.nanmean(T[:, have] / tmin - 1.0, axis=0)
        row[~have] = np.nanmax(row)                  # granules without a free chunk: priced as the worst
        return row

    # ---- allocation ---------------------------------------------------------------------
    def empty(self, shape, dtype=torch.float32, reads=(), weights=None, verify=0):
        """a tensor placed to conflict least with `reads` (weights default to their byte sizes); None when the
        arena cannot hold it.  verify=k > 1: the k best predicted positions — verify="all": every free granule-aligned
        position — are timed against sample chunks of the read tensors (a few ms per candidate) and the fastest is
        kept: for long-lived buffers (a resident output the same launch writes every step), where 0.1 s of set-up buys the
        last per cent over the prediction."""
        if verify and (verify == "all" or verify > 1) and reads:
            return self._empty_verified(shape, dtype, reads, weights, verify)
        code, bits, esize = _DL_TYPES[dtype]
        shape = tuple(int(s) for s in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * esize
        if nbytes == 0 or len(shape) > 4:
            return None
        pen = None
        reads = [r for r in reads if r is not None]
        if reads:
            w = [float(r.numel() * r.element_size()) for r in reads] if weights is None else list(weights)
            tot = sum(w) or 1.0
            acc = np.zeros(self.n_gran, dtype=np.float64)
            for r, wi in zip(reads, w):
                acc += (wi / tot) * self.price_row(r)
            pen = np.ascontiguousarray(acc, dtype=np.float32)
        L = lib()
        out = C.c_void_p()
        st = L.mp_arena_alloc_placed(nbytes, None if pen is None else pen.ctypes.data_as(C.c_void_p), self.n_gran,
                                     GRANULE, C.byref(out))
        if st == 3:       # MP_ERR_WORKSPACE: no free run of that size
            return None
        check(st, "mp_arena_alloc_placed")
        t = self._wrap(out.value, shape, dtype)
        if pen is not None:
            t._mp_predicted_conflict = float(self._footprint(out.value, nbytes) @ pen)
        return t

    def _wrap(self, ptr, shape, dtype):
        code, bits, _ = _DL_TYPES[dtype]
        managed = C.c_void_p()
        sh = (C.c_int64 * len(shape))(*shape)
        st = lib().mp_arena_dlpack(C.c_void_p(ptr), len(shape), sh, code, bits, C.byref(managed))
        if st != 0:
            lib().mp_arena_release(C.c_void_p(ptr))
            check(st, "mp_arena_dlpack")
        return torch.from_dlpack(_PyCapsule_New(managed, _CAPSULE_NAME, None))

    def empty_at(self, shape, granule, dtype=torch.float32):
        """a tensor whose range starts in granule `granule` (None when that part of the arena is taken): for studies
        and tests that compare positions"""
        shape = tuple(int(s) for s in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * _DL_TYPES[dtype][2]
        span = max(1, -(-nbytes // GRANULE))
        pen = np.ones(self.n_gran, dtype=np.float32)
        pen[granule:granule + span] = 0.0
        out = C.c_void_p()
        st = lib().mp_arena_alloc_placed(nbytes, pen.ctypes.data_as(C.c_void_p), self.n_gran, GRANULE, C.byref(out))
        if st != 0:
            return None
        if (out.value - self.base) // GRANULE != granule:
            lib().mp_arena_release(out)
            return None
        return self._wrap(out.value, shape, dtype)

    def _pair_ms(self, reads, t):
        """timed copies between sample chunks of the read tensors and of the candidate: every (read sample, candidate
        sample) pair, since a gather kernel reads all of its input while it writes each part of its output"""
        tb = t.numel() * t.element_size()
        pb = min(VERIFY_CHUNK, tb // 256 * 256)
        if pb < (16 << 20):
            return 0.0
        def samples(base, nb):
            k = 1 if nb < 2 * pb else 3
            return [base + ((nb - pb) * i // max(k - 1, 1)) // 256 * 256 for i in range(k)]
        total = 0.0
        for r in reads:
            nb = r.numel() * r.element_size()
            if nb < pb:
                continue
            for src in samples(r.data_ptr(), nb):
                for dst in samples(t.data_ptr(), tb):
                    total += min(_probe(src, dst, pb, 1) for _ in range(2))
        return total

    def _empty_verified(self, shape, dtype, reads, weights, k):
        reads = [r for r in reads if r is not None]
        if k == "all":
            timed = []                       # one candidate at a time: neighbouring positions overlap
            for g in range(self.n_gran):
                t = self.empty_at(shape, g, dtype)
                if t is not None:
                    timed.append((self._pair_ms(reads, t), g))
                    del t
            if not timed:
                return None
            best = self.empty_at(shape, min(timed)[1], dtype)
            if best is not None:
                best._mp_verified_candidates_ms = [round(ms, 4) for ms, _ in timed]
            return best
        cands = []
        held = []
        for _ in range(int(k)):     # the k best predicted positions: each candidate is held while the next is placed
            t = self.empty(shape, dtype, reads, weights)
            if t is None:
                break
            cands.append((self._pair_ms(reads, t), t))
            held.append(t)
        del held
        if not cands:
            return None
        best = min(cands, key=lambda c: c[0])[1]
        best._mp_verified_candidates_ms = [round(c[0], 4) for c in cands]
        del cands
        return best

    def stats(self):
        iu, lf = C.c_size_t(), C.c_size_t()
        check(lib().mp_arena_info(None, None, C.byref(iu), C.byref(lf)))
        return {"bytes": self.bytes, "in_use": iu.value, "largest_free": lf.value, "granule_bytes": GRANULE,
                "probe_ms_min": self.t_min_ms, "conflict_max": float(self.conflict.max()),
                "conflict_median": float(np.median(self.conflict))}


def enabled():
    return os.environ.get("MP_PLACEMENT", "auto") != "off"


def arena(device=None, create=True):
    """the calling device's arena (created and calibrated on first use), or None when placement is off or
    the device has too little free memory for one"""
    if not enabled() or not torch.cuda.is_available():
        return None
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    a = _arenas.get(idx)
    if a is None and create:
        with _lock:
            a = _arenas.get(idx)
            if a is None:
                a = False
                with torch.cuda.device(idx):
                    free, _ = torch.cuda.mem_get_info()
                    want = os.environ.get("MP_ARENA_GB")
                    nbytes = int(float(want) * GiB) if want else min(int(free * 0.5), 160 * GiB)
                    nbytes = nbytes // GRANULE * GRANULE
                    if nbytes >= 4 * GRANULE and nbytes <= free - 2 * GiB:
                        try:
                            a = Arena(nbytes)
                        except Exception:
                            a = False
                _arenas[idx] = a
    return a or None


def empty(shape, dtype=torch.float32, device=None, reads=(), force=False, verify=0):
    """placed allocation for an engine output; None when placement does not apply (small output, arena off or
    full) — the caller then uses torch.empty"""
    n = 1
    for s in shape:
        n *= int(s)
    if not force and n * _DL_TYPES.get(dtype, (0, 0, 4))[2] < MIN_BYTES:
        return None
    if dtype not in _DL_TYPES:
        return None
    a = arena(device)
    if a is None:
        return None
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        return a.empty(shape, dtype, reads, verify=verify)


def empty_or_torch(shape, device, reads=(), dtype=torch.float32, verify=0):
    t = empty(shape, dtype, device, reads, verify=verify)
    return t if t is not None else torch.empty(shape, dtype=dtype, device=device)

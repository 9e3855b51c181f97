"""Placement of the large matrices of the aggregation path: check, and re-allocate on conflict.

On MI355X one HBM-bound launch that reads X and writes Y is up to ~13 % slower depending on which physical memory
backs the two: the high address bits are hashed into the DRAM bank / channel selection, and a read
stream and a write stream that hash alike pay bus turnarounds.  Successive allocations of torch's caching allocator
land in this structure wherever the driver put them — measured on the headline aggregation (X, Y 10 GB each,
profiles/r03_placement_retry.log): five of eight successive positions of Y run at 20.8-20.9 ms, three at 23.2-23.5 ms.
A short probe with the aggregation's own access pattern (mp_probe_gather_ms: every 1 KiB row written into a 256 MiB
sample of the candidate is the sum of 10 pseudo-random rows of the read tensor) tells the two apart at the size the
aggregation feels it; a plain 1:1 copy shows only +4 % where the aggregation loses 12 %.

So every large output of an engine operator is allocated BY TORCH (ordinary tensors: torch's lifetime, stream
semantics, memory accounting and out-of-memory handling all apply — the engine owns no device memory), checked
against the tensors the launch reads with that probe, and, when it conflicts, held while the next candidate is
allocated (the caching allocator then hands out a different block).  The fastest of at most `MP_PLACE_TRIES`
candidates is kept; the others go back to torch's cache, where anything may reuse them.  Probe results are remembered
per (read buffers, candidate buffer), so the steady state of a training loop — the same blocks every step — pays
dictionary look-ups only: the first sight of a pair costs ~6 ms of timed probes (and a synchronisation).

Round 2 solved the same problem with an engine-owned arena (one hipMalloc of half the free memory + a calibrated
conflict map).  It placed as well but took memory torch could not see or reclaim, calibrated for 1.8 s per process and
reused ranges without stream tracking; it is gone.

Environment: MP_PLACEMENT=off disables the check; MP_PLACE_MIN_MB the size from which outputs are checked (default
1024: smaller ones live in L2 / Infinity Cache); MP_PLACE_TRIES the candidates per allocation at most (default 4);
MP_PLACE_EXPLORE the candidates the first two read sets of a process may look at while no two candidates have differed
yet (default 8: on one box 7 of 10 successive blocks conflicted with the read tensor, and four candidates that all look
alike can all be slow; every further candidate is held while the next is allocated and churns torch's cache — in a
training loop that means new buffer pairs and new probes in later steps — so the wide search is for the first
allocations only);
MP_PLACE_ACCEPT the accepted slow-down of the probe against the fastest probe seen (default 0.05: good positions
measure +0-4 %, conflicting ones +6-12 %); MP_PLACE_BUDGET_MS the probe time a process may spend per device in all
(default 250 ms — bench.py's one long-lived output takes ~40) and MP_PLACE_BUDGET_SEARCHES the allocations that may
probe anything new in all (default 12): a training loop on fresh batches meets new buffers every step and would otherwise
probe — and synchronise — in every one of them for ever (measured on the ID-GIN ego-batch step: 13 probes and +30 ms per
step); past the budget outputs are torch's blocks as they come;
MP_PLACE_HOLD_FRAC the share of free memory rejected candidates may hold while a search runs (default 0.25).
"""
import collections
import ctypes as C
import os
import threading

import torch

from ._lib import check, lib

MiB = 1 << 20
CHUNK = 256 * MiB                         # bytes of the candidate written per probe (reads: FAN x as much, past every cache)
FAN = 10                                  # rows read per row written: the mean degree of the path's graphs
MIN_BYTES = int(os.environ.get("MP_PLACE_MIN_MB", "1024")) * MiB
TRIES = int(os.environ.get("MP_PLACE_TRIES", "4"))
EXPLORE_TRIES = int(os.environ.get("MP_PLACE_EXPLORE", "8"))   # candidates for the first read sets of a process, see below
EXPLORE_SETS = 2
ACCEPT = float(os.environ.get("MP_PLACE_ACCEPT", "0.05"))
BUDGET_MS = float(os.environ.get("MP_PLACE_BUDGET_MS", "250"))   # probe launches a process may spend per device, in all
BUDGET_SEARCHES = int(os.environ.get("MP_PLACE_BUDGET_SEARCHES", "12"))   # allocations that may probe anything new, in all
HOLD_FRAC = float(os.environ.get("MP_PLACE_HOLD_FRAC", "0.25"))   # rejected candidates held at once: at most this share of free memory
MEMO_ENTRIES = 4096

_lock = threading.RLock()             # memo / yardsticks are also touched from the autograd thread
_state = {}                               # device index -> {"t_min": {chunk bytes: ms}, "memo": OrderedDict, "stats": {...}}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _probe(src, dst, nbytes, reps=1):
    """mean ms of `reps` streaming copies src -> dst after one untimed copy (mp_probe_copy_ms; SYNCHRONISES)"""
    ms = C.c_float(0)
    check(lib().mp_probe_copy_ms(C.c_void_p(src), C.c_void_p(dst), nbytes, reps, C.byref(ms), _stream()),
          "mp_probe_copy_ms")
    return ms.value


def _probe_gather(src, src_bytes, dst, dst_bytes, reps=1):
    """mean ms of `reps` launches writing dst_bytes of dst, every 1 KiB row the sum of FAN random rows of ALL of src
    (mp_probe_gather_ms; one untimed launch first; SYNCHRONISES)"""
    ms = C.c_float(0)
    check(lib().mp_probe_gather_ms(C.c_void_p(src), src_bytes, C.c_void_p(dst), dst_bytes, FAN, reps, C.byref(ms),
                                   _stream()), "mp_probe_gather_ms")
    return ms.value


def _segments_freed(device):
    try:
        return torch.cuda.memory_stats(device).get("segment.all.freed", 0)
    except Exception:
        return 0


_paused = [0]


def pause():
    """stop checking placements (outputs are plain torch allocations) until resume(): for loops whose large buffers live for
    ONE step — a fresh batch every step (graphgym_amd/pipeline.py) — where no (read, output) pair is ever seen twice, a
    check can only cost (a search holds candidates: new 1 GiB blocks from the driver, ~20 ms each)"""
    _paused[0] += 1


def resume():
    _paused[0] = max(0, _paused[0] - 1)


def enabled():
    return os.environ.get("MP_PLACEMENT", "auto") != "off" and _paused[0] == 0


def _dev_state(device):
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    st = _state.get(idx)
    if st is None:
        with _lock:
            st = _state.setdefault(idx, {"t_min": {}, "memo": collections.OrderedDict(), "contrast": False, "explore_left": EXPLORE_SETS,
                                         "stats": {"allocations": 0, "probed_pairs": 0, "memo_hits": 0, "retries": 0,
                                                   "probe_ms_total": 0.0}})
    return st


def stats(device=None):
    """counters of this process's placement work on `device` (bench.py reports them)"""
    st = _dev_state(device if device is not None else torch.cuda.current_device())
    return dict(st["stats"], read_sets_seen=len(st["t_min"]))


def _samples(ptr, nbytes, chunk):
    """start addresses of up to three sample chunks of the candidate: both ends and the middle"""
    k = 1 if nbytes < 2 * chunk else 3
    return [ptr + (((nbytes - chunk) * i // max(k - 1, 1)) // 1024) * 1024 for i in range(k)]


def pair_cost_ms(reads, t, st=None):
    """time of the gather probe writing a `chunk` of the candidate from ALL of a read tensor, averaged over the
    candidate's sample positions (each the faster of two timed launches) and over the read tensors by size; remembered
    per (read buffers, candidate buffer).  Returns (ms, chunk) or (None, None) when nothing is big enough to matter."""
    st = st if st is not None else _dev_state(t.device)
    tb = t.numel() * t.element_size()
    big = [(r.data_ptr(), r.numel() * r.element_size()) for r in reads
           if r is not None and r.is_cuda and r.numel() * r.element_size() >= CHUNK and r.data_ptr() % 16 == 0]
    if not big or tb < CHUNK or t.data_ptr() % 16:
        return None, None
    chunk = CHUNK
    memo = st["memo"]
    total, wsum, cnt = 0.0, 0.0, 0
    for ptr, nb in big:                      # remembered per (read buffer, candidate buffer): launches share buffers
        key = (ptr, nb, t.data_ptr(), tb)
        acc = memo.get(key)
        if acc is not None:
            memo.move_to_end(key)
            st["stats"]["memo_hits"] += 1
        elif st["stats"]["probe_ms_total"] >= BUDGET_MS or st["stats"].get("searches", 0) > BUDGET_SEARCHES:
            # a loop whose buffers move every step (a fresh batch per step: new sizes, new addresses) would probe —
            # and synchronise — in every step for ever: a process checks placements until its budget of probes is
            # spent (MP_PLACE_BUDGET_MS of probe launches in all), then takes torch's blocks as they come
            st["stats"]["budget_refusals"] = st["stats"].get("budget_refusals", 0) + 1
            return None, None
        else:
            acc = 0.0
            pos = _samples(t.data_ptr(), tb, chunk)
            for dst in pos:
                acc += min(_probe_gather(ptr, nb // 1024 * 1024, dst, chunk, 1) for _ in range(2))
            acc /= len(pos)
            cnt += len(pos)
            memo[key] = acc
            if len(memo) > MEMO_ENTRIES:
                memo.popitem(last=False)
        total += nb * acc
        wsum += nb
    ms = total / wsum
    if cnt and not st.get("_in_search"):
        st["stats"]["searches"] = st["stats"].get("searches", 0) + 1     # an allocation that met a new pair
        st["_in_search"] = True
    st["stats"]["probed_pairs"] += cnt
    st["stats"]["probe_ms_total"] += 4.0 * ms * cnt          # two trials, each one untimed + one timed launch (the
    #                                                          budget check above reads this running total)
    return ms, chunk


def empty_or_torch(shape, device, reads=(), dtype=torch.float32, tries=None, accept=None, streaming=False):
    """torch.empty(shape) for an engine operator's output, placed against the tensors `reads` the launch reads: see the
    module docstring.  tries: candidates to consider (default MP_PLACE_TRIES; a long-lived buffer may ask for more);
    with tries == 1 the allocation is only priced (its probe result is on the tensor as `_mp_place`); accept: the
    probe slow-down that ends the search (default MP_PLACE_ACCEPT; negative: time all `tries` candidates).
    streaming: the launch reads and writes 1 : 1 (transform, BatchNorm, masks) — such a pair loses at most ~4 % in a
    conflicting position (the copy probe of profiles/r03_placement_retry.log), which does not pay for candidates held in
    memory: the output is a plain torch.empty.  Checked are the outputs of the gather launches (aggregation, one-kernel
    layer, two-branch aggregation, attention-weighted aggregation), which lose 12 %."""
    nbytes = torch.empty((), dtype=dtype).element_size()
    for s in shape:
        nbytes *= int(s)
    t = torch.empty(shape, dtype=dtype, device=device)
    if (streaming or nbytes < MIN_BYTES or not enabled() or not t.is_cuda or not reads
            or torch.cuda.is_current_stream_capturing()):     # a probe synchronises: never under HIP-graph capture
        return t
    tries_given = tries
    tries = TRIES if tries is None else int(tries)
    accept = ACCEPT if accept is None else float(accept)
    st = _dev_state(t.device)
    st["stats"]["allocations"] += 1
    st["_in_search"] = False
    with _lock, torch.cuda.device(t.device):
        # a segment torch returned to the driver since the last look (empty_cache, an out-of-memory retry) may come back
        # at the same virtual address on different physical memory: remembered probe results would be stale
        freed = _segments_freed(t.device)
        if freed != st.get("segments_freed"):
            st["memo"].clear()
            st["segments_freed"] = freed
        ms, chunk = pair_cost_ms(reads, t, st)
        if ms is None:
            return t
        # Yardsticks: the fastest probe seen FOR THESE READ TENSORS (another read tensor sits elsewhere and has another
        # best) and the fastest seen for ANY read tensor in this process.  The search ends when the best candidate is
        # within `accept` of the first and 2 x `accept` of the second — for a read set seen for the first time after
        # three candidates at least.  While this process has not yet seen two candidates of one read set differ by more
        # than `accept` (a "contrast": both bands of the address hash observed, so the yardsticks are known to come
        # from the fast one), its first EXPLORE_SETS new read sets keep looking until they see one, up to EXPLORE_TRIES.
        rkey = (chunk,) + tuple((r.data_ptr(), r.numel() * r.element_size()) for r in reads if r is not None)
        gkey = ("any", chunk)
        known = rkey in st["t_min"]
        st["t_min"][rkey] = min(st["t_min"].get(rkey, ms), ms)
        st["t_min"][gkey] = min(st["t_min"].get(gkey, ms), ms)
        explore = (not known and not st["contrast"] and st["explore_left"] > 0 and tries_given is None and accept >= 0.0)
        if explore:
            st["explore_left"] -= 1
            tries = max(tries, EXPLORE_TRIES)
        best, best_ms, seen = t, ms, [ms]
        held = []
        # the search holds every rejected candidate while it allocates the next (so that torch hands out a DIFFERENT
        # block): bounded to HOLD_FRAC of the memory free when the search starts, whatever `tries` says
        hold_budget = int(HOLD_FRAC * torch.cuda.mem_get_info(t.device)[0])
        while len(seen) < tries:
            if (len(held) + 1) * nbytes > hold_budget:
                st["stats"]["hold_cap_hits"] = st["stats"].get("hold_cap_hits", 0) + 1
                break
            spread = accept >= 0.0 and max(seen) > (1.0 + accept) * min(seen)
            if spread:
                st["contrast"] = True
            good = (best_ms <= (1.0 + accept) * st["t_min"][rkey]
                    and best_ms <= (1.0 + 2.0 * max(accept, 0.0)) * st["t_min"][gkey])
            if good and (known or (len(seen) >= 3 and (spread or not explore))):
                break
            held.append(t)
            try:
                t = torch.empty(shape, dtype=dtype, device=device)
            except torch.OutOfMemoryError:
                break                                       # no room for another candidate: keep the best so far
            ms, _ = pair_cost_ms(reads, t, st)
            if ms is None:                                  # the probe budget ran out in mid-search: keep the best so far
                break
            st["stats"]["retries"] += 1
            seen.append(ms)
            st["t_min"][rkey] = min(st["t_min"][rkey], ms)
            st["t_min"][gkey] = min(st["t_min"][gkey], ms)
            if ms < best_ms:
                best, best_ms = t, ms
        st["stats"]["held_bytes_peak"] = max(st["stats"].get("held_bytes_peak", 0), len(held) * nbytes)
        n_held = len(held)
        del held, t
        if n_held > 2 and os.environ.get("MP_PLACE_RELEASE", "1") != "0":
            # a wide search leaves several full-size blocks in torch's cache that nothing may ever ask for again: hand the
            # cached blocks of THIS search back to the driver.  torch offers no per-block release outside a private pool
            # (and a pool would take the chosen block out of the cache every later step reuses), so this is
            # empty_cache() — issued only after a search that held more than two candidates, i.e. a process's first
            # allocations, never in the steady state of a loop (there the memo answers and nothing is held)
            torch.cuda.empty_cache()
            st["stats"]["cache_releases"] = st["stats"].get("cache_releases", 0) + 1
            keep = best.data_ptr()                      # the released blocks' addresses may come back on other memory
            for k in [k for k in st["memo"] if k[2] != keep]:
                del st["memo"][k]
            st["segments_freed"] = _segments_freed(best.device)
        if len(st["t_min"]) > MEMO_ENTRIES:
            st["t_min"].clear()
        tmin = st["t_min"].get(rkey, best_ms)
    best._mp_place = {"candidates_ms": [round(v, 4) for v in seen], "chosen_ms": round(best_ms, 4),
                      "probe_rel": best_ms / tmin - 1.0, "chunk_bytes": chunk}
    return best

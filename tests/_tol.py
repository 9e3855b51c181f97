"""The ONE tolerance regime of the parity tests.

north_star's bar is "fp32 aggregation within 1e-5".  It is applied PER ROW: the error of an output row is
measured against that row's own magnitude (max |reference| over the row), not the matrix's, so a small row
cannot hide behind a large one.  The reference for fp32 results is a float64 evaluation of the same
formula on the same fp32 inputs (the oracle functions are dtype-generic: `both(fn)` evaluates an oracle closure
in float64 and in float32).  Where a result passes through fp32 arithmetic whose own rounding can exceed
1e-5 of a row (long dot products, cancellation, softmax denominators), the allowance is not a wider constant but the
distance of the reference's OWN fp32 CPU evaluation (the oracle in float32) from the float64 result, times two:
the engine must be as close to the exact answer as the reference's CPU path is.

That allowance is policed, not just granted: every call records how many rows needed it, and fails when
  * more than `max_ref32_frac` of the rows needed it (default 0.5: an operator whose typical row misses 1e-5 is
    not "within 1e-5 up to fp32 GEMM rounding", it is a different accuracy class and must say so at the call site), or
  * over the rows that needed it, the MEDIAN of err / (oracle32's own err) exceeds `max_median_ratio` (default 1.5):
    the 2x is slack for single rows; a kernel that is 1.9x worse than the CPU path on every row does not pass.
The per-call statistics are appended to STATS and written to gpurun_out/tol_stats.json at the end of the session
(tests/conftest.py), so DESIGN.md can quote how often the second term was the binding one.

Parameter gradients and losses (reductions over all rows) use `assert_close_all`: one scale for the whole tensor.
"""
import numpy as np
import torch

STATS = []          # one record per call: {what, rows, needed_ref32, frac, median_ratio, worst_rel}


def _t64(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().double()
    return torch.from_numpy(np.asarray(v)).double()


def to64(t):
    return t.double() if isinstance(t, torch.Tensor) and t.is_floating_point() else t


def to32(t):
    return t.float() if isinstance(t, torch.Tensor) and t.is_floating_point() else t


def both(fn):
    """(fn(to64), fn(to32)): an oracle closure `fn(c)` — c casts each float input — evaluated in float64 and in the
    float32 the reference runs in.  fn may return a tensor or a tuple / list of tensors (e.g. outputs and gradients).
    The float64 evaluation runs under torch.set_default_dtype(float64) so that tensors the oracle creates itself
    (ones for missing weights, zeros for accumulators) have the evaluation's dtype."""
    old = torch.get_default_dtype()
    try:
        torch.set_default_dtype(torch.float64)
        r64 = fn(to64)
    finally:
        torch.set_default_dtype(old)
    r32 = fn(to32)
    return r64, r32


def _record(what, n_rows, needed, ratios, worst):
    STATS.append({"what": what, "rows": int(n_rows), "needed_ref32": int(needed),
                  "frac": float(needed) / max(int(n_rows), 1),
                  "median_ratio": None if ratios is None else float(ratios),
                  "worst_rel": float(worst)})


def assert_close_rows(a, ref64, tol=1e-5, ref32=None, what="", max_ref32_frac=0.5, max_median_ratio=1.5):
    """rows of a 2-D result (a 1-D result is one row)"""
    a, r = _t64(a), _t64(ref64)
    assert a.shape == r.shape, (what, a.shape, r.shape)
    if a.numel() == 0:
        return
    if a.dim() == 1:
        a, r = a[None], r[None]
    a, r = a.reshape(a.size(0), -1), r.reshape(r.size(0), -1)
    err = (a - r).abs().amax(dim=1)
    scale = r.abs().amax(dim=1)
    base = tol * scale
    allow = base
    e32 = None
    if ref32 is not None:
        o = _t64(ref32)
        o = (o[None] if o.dim() == 1 else o).reshape(r.shape)
        e32 = (o - r).abs().amax(dim=1)
        allow = torch.maximum(base, 2.0 * e32)
    # an all-zero reference row must be reproduced exactly (empty neighbourhoods, masked rows)
    bad = err > allow
    worst = float((err / scale.clamp(min=1e-300)).where(scale > 0, torch.zeros_like(err)).max())
    if bool(bad.any()):
        i = int(torch.nonzero(bad)[0])
        raise AssertionError(f"{what}: row {i}: err {float(err[i]):.3e} > allowed {float(allow[i]):.3e} "
                             f"(row max {float(r[i].abs().max()):.3e}); {int(bad.sum())} of {bad.numel()} rows off")
    needed = err > base                    # rows that passed only through the ref32 term
    n_need = int(needed.sum())
    med = None
    if n_need:
        med = float((err[needed] / e32[needed].clamp(min=1e-300)).median())
    _record(what, err.numel(), n_need, med, worst)
    if n_need:
        frac = n_need / err.numel()
        assert frac <= max_ref32_frac, (f"{what}: {n_need} of {err.numel()} rows ({frac:.1%}) are outside {tol:g} of "
                                        f"float64 and pass only through the fp32 oracle's own error "
                                        f"(limit {max_ref32_frac:.0%})")
        assert med <= max_median_ratio, (f"{what}: over the {n_need} rows outside {tol:g}, the median error is "
                                         f"{med:.2f}x the fp32 oracle's own (limit {max_median_ratio})")


def assert_close_all(a, ref64, tol=1e-5, ref32=None, what=""):
    """one scale for the whole tensor (parameter gradients, losses: reductions over all rows)"""
    a, r = _t64(a), _t64(ref64)
    assert a.shape == r.shape, (what, a.shape, r.shape)
    if a.numel() == 0:
        return
    err = float((a - r).abs().max())
    scale = float(r.abs().max())
    allow = tol * scale
    e32 = None
    if ref32 is not None:
        e32 = float((_t64(ref32).reshape(r.shape) - r).abs().max())
        allow = max(allow, 2.0 * e32)
    assert err <= allow, f"{what}: err {err:.3e} > allowed {allow:.3e} (max |ref| {scale:.3e})"
    need = err > tol * scale
    _record(what + " [all]", 1, int(need), (err / max(e32, 1e-300)) if need else None, err / max(scale, 1e-300))


def close(a, refs, tol=1e-5, what="", **kw):
    """assert_close_rows against `refs` = (ref64, ref32) from both(...), or a float64 reference alone"""
    if isinstance(refs, tuple):
        return assert_close_rows(a, refs[0], tol, ref32=refs[1], what=what, **kw)
    return assert_close_rows(a, refs, tol, what=what, **kw)


def close_all(a, refs, tol=1e-5, what=""):
    if isinstance(refs, tuple):
        return assert_close_all(a, refs[0], tol, ref32=refs[1], what=what)
    return assert_close_all(a, refs, tol, what=what)

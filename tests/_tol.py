"""Tolerance helpers shared by the parity tests.

north_star's bar is "fp32 aggregation within 1e-5".  It is applied PER ROW: the error of an output row is
measured against that row's own magnitude (max |reference| over the row), not the matrix's, so a small row
cannot hide behind a large one.  The reference for fp32 results is a float64 evaluation of the same
formula on the same fp32 inputs.  Where a result passes through fp32 GEMMs whose own rounding can exceed
1e-5 of a row (long dot products, cancellation), the allowance is not a wider constant but the distance of
the reference's OWN fp32 CPU evaluation (the oracle in float32) from the float64 result, times two:
the engine must be as close to the exact answer as the reference's CPU path is.
"""
import numpy as np
import torch


def _t64(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().double()
    return torch.from_numpy(np.asarray(v)).double()


def assert_close_rows(a, ref64, tol=1e-5, ref32=None, what=""):
    """rows of a 2-D result (a 1-D result is one row)"""
    a, r = _t64(a), _t64(ref64)
    assert a.shape == r.shape, (what, a.shape, r.shape)
    if a.numel() == 0:
        return
    if a.dim() == 1:
        a, r = a[None], r[None]
    a, r = a.reshape(a.size(0), -1), r.reshape(r.size(0), -1)
    err = (a - r).abs().amax(dim=1)
    allow = tol * r.abs().amax(dim=1)
    if ref32 is not None:
        o = _t64(ref32)
        o = (o[None] if o.dim() == 1 else o).reshape(r.shape)
        allow = torch.maximum(allow, 2.0 * (o - r).abs().amax(dim=1))
    # an all-zero reference row must be reproduced exactly (empty neighbourhoods, masked rows)
    bad = err > allow
    if bool(bad.any()):
        i = int(torch.nonzero(bad)[0])
        raise AssertionError(f"{what}: row {i}: err {float(err[i]):.3e} > allowed {float(allow[i]):.3e} "
                             f"(row max {float(r[i].abs().max()):.3e}); {int(bad.sum())} of {bad.numel()} rows off")


def assert_close_all(a, ref64, tol=1e-5, ref32=None, what=""):
    """one scale for the whole tensor (parameter gradients, losses: reductions over all rows)"""
    a, r = _t64(a), _t64(ref64)
    assert a.shape == r.shape, (what, a.shape, r.shape)
    if a.numel() == 0:
        return
    err = float((a - r).abs().max())
    allow = tol * float(r.abs().max())
    if ref32 is not None:
        allow = max(allow, 2.0 * float((_t64(ref32).reshape(r.shape) - r).abs().max()))
    assert err <= allow, f"{what}: err {err:.3e} > allowed {allow:.3e} (max |ref| {float(r.abs().max()):.3e})"

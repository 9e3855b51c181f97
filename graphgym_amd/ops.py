"""Differentiable operators over CSRGraph, each a thin autograd wrapper around one
C-ABI entry point of libmpengine.so (include/mp_engine.h).

    spmm(g, x, reduce)           SparseAdj.matmul (sparse_adj.py:91-97); PyG propagate
    idgnn_aggregate(g, id, x)    two-branch form of gcn_id (TfgIDLayer.py:510-517)
    index_add_rows(h, id, u)     tensor_scatter_nd_add / index_add_ (K10)
    edge_softmax / sddmm_*       GAT pieces (TfgIDLayer.py:333-355; idconv.py:317-332)
"""
import ctypes as C
import os

import torch

from . import _lib, placement
from ._lib import check, lib, ptr
from .graph import CSRGraph, _require_hip, _stream


def _f32c(t, name):
    _require_hip(t, name)
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the path aggregates in fp32), got {t.dtype}")
    return t if t.stride(-1) == 1 and t.dim() == 2 else t.contiguous()


def _raw_spmm(g, x, reduce, S=None, self_scale=0.0, bias=None, relu=False, want_argmax=False,
              col_override=None, out=None):
    """one launch of mp_spmm_csr_f32 — or, for plain sum / mean / max (values only, no argmax) at d = 128 / 256 / 512 on
    a large operator, of mp_agg_rows_tiles_f32 (the same aggregation on the producer/consumer tile structure: 2-5 %
    faster; MP_AGG_TILES=0 keeps the plan-based kernel) —; x [n_src, d] -> y [N, d] (written into `out` when given)"""
    L = lib()
    N, d = g.num_nodes, x.size(1)
    y = out if out is not None else placement.empty_or_torch((N, d), x.device, reads=(x,))
    if (reduce in (_lib.SUM, _lib.MEAN, _lib.MAX) and d in AGG_TILES_WIDTHS and N >= AGG_TILES_MIN_ROWS and bias is None
            and not relu and not want_argmax and col_override is None and not (reduce != _lib.SUM and S is not None)
            and os.environ.get("MP_AGG_TILES", "1") != "0"
            and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0
            and (S is None or (S.stride(0) % 4 == 0 and S.data_ptr() % 16 == 0))
            and g.nnz > 0 and g.max_row_entries() <= FUSED_MAX_ROW):
        global AGG_TILES_CALLS
        AGG_TILES_CALLS += 1
        with torch.cuda.device(x.device):
            check(L.mp_agg_rows_tiles_f32(ptr(g.rowptr), ptr(g.col), ptr(g.val), N, reduce, ptr(x), x.stride(0), d,
                                          ptr(S), S.stride(0) if S is not None else 0, float(self_scale),
                                          ptr(y), y.stride(0), _stream()), "mp_agg_rows_tiles_f32")
        return y, None
    argmax = torch.empty((N, d), dtype=torch.int32, device=x.device) if want_argmax else None
    plan, counts = g.plan()
    with torch.cuda.device(x.device):
        nb = C.c_size_t(0)
        check(L.mp_spmm_ws_bytes(counts, d, reduce, 0, C.byref(nb)))
        ws = torch.empty(nb.value, dtype=torch.uint8, device=x.device) if nb.value else None
        col = g.col if col_override is None else col_override
        check(L.mp_spmm_csr_f32(ptr(g.rowptr), ptr(col), ptr(g.val), N, ptr(plan), counts,
                                ptr(x), x.stride(0), ptr(y), y.stride(0), d, reduce,
                                ptr(S), S.stride(0) if S is not None else 0, float(self_scale),
                                ptr(bias), _lib.ACT_RELU if relu else _lib.ACT_NONE, ptr(argmax),
                                ptr(ws), nb.value, _stream()), "mp_spmm_csr_f32")
    return y, argmax


# (spmm: registered operator mp::spmm, below)


def spmm_fused_eval(g, x, reduce="sum", self_scale=0.0, col_scale=None, col_shift=None, relu=False,
                    l2norm=False, l2_eps=1e-12):
    """Inference-only aggregation with the layer's post-ops folded into the row flush:
    act((agg + self_scale * x) * col_scale + col_shift), then optional row L2 normalisation
    (graphgym/models/layer.py:26-47, gnn.py:79-80 with BatchNorm in eval mode).  No autograd."""
    x = _f32c(x.detach(), "x")
    L = lib()
    N, d = g.num_nodes, x.size(1)
    y = torch.empty((N, d), dtype=torch.float32, device=x.device)
    plan, counts = g.plan()
    red = _lib.REDUCE[reduce]
    S = x if self_scale != 0.0 else None
    with torch.cuda.device(x.device):
        nb = C.c_size_t(0)
        check(L.mp_spmm_ws_bytes(counts, d, red, 0, C.byref(nb)))
        ws = torch.empty(nb.value, dtype=torch.uint8, device=x.device) if nb.value else None
        cs = None if col_scale is None else col_scale.detach().contiguous()
        ct = None if col_shift is None else col_shift.detach().contiguous()
        st = L.mp_spmm_csr_epilogue_f32(ptr(g.rowptr), ptr(g.col), ptr(g.val), N, ptr(plan), counts, ptr(x),
                                        x.stride(0), ptr(y), y.stride(0), d, red, ptr(S),
                                        S.stride(0) if S is not None else 0, float(self_scale), ptr(cs), ptr(ct),
                                        _lib.ACT_RELU if relu else _lib.ACT_NONE, 1 if l2norm else 0,
                                        float(l2_eps), ptr(ws), nb.value, _stream())
        if st == 2 and l2norm:   # row wider than one wave: normalise in a second pass
            check(L.mp_spmm_csr_epilogue_f32(ptr(g.rowptr), ptr(g.col), ptr(g.val), N, ptr(plan), counts, ptr(x),
                                             x.stride(0), ptr(y), y.stride(0), d, red, ptr(S),
                                             S.stride(0) if S is not None else 0, float(self_scale), ptr(cs),
                                             ptr(ct), _lib.ACT_RELU if relu else _lib.ACT_NONE, 0, float(l2_eps),
                                             ptr(ws), nb.value, _stream()), "mp_spmm_csr_epilogue_f32")
            return torch.nn.functional.normalize(y, p=2, dim=-1, eps=l2_eps)
        check(st, "mp_spmm_csr_epilogue_f32")
    return y


def _raw_dense_fused(P, W, Q, W_id, bias, relu):
    """out = act(P @ W [+ Q @ W_id] + bias) on the engine's MFMA kernel; None if the shapes are outside
    what the kernel covers (the caller then composes library ops)"""
    L = lib()
    M, F = P.shape
    d = W.size(1)
    if Q is None and dense_x3_supported(P, F, d):
        return _raw_dense_x3(P, W, bias, relu)
    out = placement.empty_or_torch((M, d), P.device, reads=(P, Q), streaming=True)
    Wc = W.contiguous()
    Wi = None if W_id is None else W_id.contiguous()
    b = None if bias is None else bias.contiguous()
    with torch.cuda.device(P.device):
        st = L.mp_dense_fused_f32(ptr(P), P.stride(0), ptr(Wc), ptr(Q), Q.stride(0) if Q is not None else 0,
                                  ptr(Wi), ptr(b), _lib.ACT_RELU if relu else _lib.ACT_NONE, ptr(out),
                                  out.stride(0), M, F, d, _stream())
    if st in (2, 5):
        return None
    check(st, "mp_dense_fused_f32")
    return out


X3_WIDTHS = (64, 128, 256)
X3_MIN_ROWS = int(os.environ.get("MP_X3_MIN_ROWS", 1 << 17))   # below this the persistent 256-row blocks do not fill the chip


def _split_w(W, trans=False):
    """W -> the engine's three-way bf16 split of W (or of W^T): [3, K / 8, n, 8] bf16 (mp_split_w_bf16x3; one
    small launch, so nothing is cached across calls)"""
    Wc = W.detach()
    if Wc.stride(1) != 1 and Wc.stride(0) == 1:      # a transposed view (nn.Linear's weight.t()): split its base
        Wc, trans = Wc.t(), not trans
    elif Wc.stride(1) != 1:
        Wc = Wc.contiguous()
    K, n = (Wc.size(1), Wc.size(0)) if trans else (Wc.size(0), Wc.size(1))
    sp = torch.empty((3, K // 8, n, 8), dtype=torch.bfloat16, device=Wc.device)
    with torch.cuda.device(Wc.device):
        check(lib().mp_split_w_bf16x3(ptr(Wc), Wc.stride(0), K, n, 1 if trans else 0, ptr(sp), _stream()),
              "mp_split_w_bf16x3")
    return sp


def dense_x3_supported(P, K, n, out=None):
    """shapes the streaming transform takes (mp_dense_x3_f32): [M, K] @ [K, n] with n = 64 / 128 / 256, K % 32 == 0,
    K >= 64, 16-byte-aligned rows on both sides, and enough rows to fill the chip; MP_X3=0 turns it off (A/B timing)"""
    return (os.environ.get("MP_X3", "1") != "0" and P.dim() == 2 and P.size(0) >= X3_MIN_ROWS and n in X3_WIDTHS
            and K % 32 == 0 and K >= 64 and P.size(1) == K and P.stride(1) == 1 and P.stride(0) % 4 == 0
            and P.data_ptr() % 16 == 0 and P.dtype == torch.float32
            and (out is None or (out.stride(1) == 1 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0)))


def _raw_dense_x3(P, W, bias=None, relu=False, trans=False, out=None):
    """act(P @ W + bias) (trans: P @ W^T) on the streaming kernel; the caller has checked dense_x3_supported"""
    M, K = P.shape
    n = W.size(0) if trans else W.size(1)
    sp = _split_w(W, trans)
    if out is None:
        out = placement.empty_or_torch((M, n), P.device, reads=(P,), streaming=True)
    b = None if bias is None else bias.detach().contiguous()
    if b is not None and b.data_ptr() % 16:
        b = b.clone()                        # a slice of a longer bias: the kernel reads it in 16-byte groups
    with torch.cuda.device(P.device):
        check(lib().mp_dense_x3_f32(ptr(P), P.stride(0), ptr(sp), ptr(b), _lib.ACT_RELU if relu else _lib.ACT_NONE,
                                    ptr(out), out.stride(0), M, K, n, _stream()), "mp_dense_x3_f32")
    return out


def times_wt(g, W):
    """g @ W^T — the input gradient of a transform: the streaming kernel at its shapes, else the library GEMM"""
    if dense_x3_supported(g, W.size(1), W.size(0)):
        return _raw_dense_x3(g, W, trans=True)
    return torch.mm(g, W.detach().t())


FUSED_WIDTHS = (64, 128, 256, 512)
AGG_TILES_CALLS = 0                   # launches of mp_agg_rows_tiles_f32 by this process (tests assert the dispatch)
AGG_TILES_WIDTHS = (128, 256, 512)   # widths of mp_agg_rows_tiles_f32 (d = 128: 10.04 -> 9.85 ms; d = 64 stays on the plan-based kernel)
AGG_TILES_MIN_ROWS = 1 << 21    # crossover against the plan-based kernel on BA graphs (d = 256): 1e6 rows 2.15 vs 1.73 ms, 2e6 3.72 vs 3.82, 3e6 5.45 vs 5.82, 1e7 19.6 vs 20.6
FUSED_MAX_ROW = 1 << 18      # longer rows (star-like hubs) go to the plan-based kernel, which spreads them over many waves


def agg_dense_supported(g, x, W):
    """shapes the one-kernel aggregate -> transform takes (mp_agg_dense_f32); MP_FUSED=0 in the environment
    turns the path off (A/B timing against the two-kernel order)"""
    if os.environ.get("MP_FUSED", "1") == "0":
        return False
    return (x.size(1) in FUSED_WIDTHS and W.size(1) % 2 == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and (x.size(1) < 512 or W.size(1) <= 512)
            and g.nnz > 0 and g.max_row_entries() <= FUSED_MAX_ROW)


BF16X3_MIN_ROWS = 1 << 14   # below this the split of W (one ~5 us launch per call) costs more than the shorter MFMA phase saves
                            # (round 2: 2^19, set on the one-role kernel; an ego batch of 1.5 * 10^5 rows at F = 512 ran the
                            # exact-f32 product at 1.24 ms per layer, MFMA-bound)


def _split_bf16_t(W):
    """W [F, d] fp32 -> [3, F / 8, d, 8] bf16: W split three ways, plane s = bf16(W - sum of the planes before it)
    (24 mantissa bits in all), in the layout the bf16x3 product of mp_agg_dense_f32 loads (a lane's 8 k-values of a
    column contiguous, neighbouring columns neighbours).  Split afresh on every call (mp_split_w_bf16x3, one ~5 us
    launch): nothing observable from Python says W is unchanged — `w.data.uniform_()` (the reference's own
    reset_parameters idiom, idconv.py:125-128) rewrites a weight without touching `_version`, and a cached split
    would then multiply by the old weights silently."""
    return _split_w(W)


def _raw_agg_dense(g, x, W, bias=None, relu=False, S=None, self_scale=0.0, want_P=False, reduce=_lib.SUM,
                   out=None, defer_act=None, bf16x3=None, residual=None):
    """out = act((reduce_j w_ij x[j] + self_scale * S) W + bias) in one launch (into the view `out` when given);
    returns (out, P or None) with P the aggregated rows; defer_act [N] uint8: rows stored without the activation.
    bf16x3: run the product on the bf16 matrix pipe with three-way split operands (fp32-accurate, 3/8 of the MFMA
    cycles); default: on for N >= BF16X3_MIN_ROWS unless MP_BF16X3=0."""
    L = lib()
    N, F, d = g.num_nodes, x.size(1), W.size(1)
    Wc = W.contiguous()
    if bf16x3 is None:
        bf16x3 = N >= BF16X3_MIN_ROWS and os.environ.get("MP_BF16X3", "1") != "0"
    Wsp = _split_bf16_t(Wc) if bf16x3 else None
    b = None if bias is None else bias.contiguous()
    if out is None:
        out = placement.empty_or_torch((N, d), x.device, reads=(x,))
    P = placement.empty_or_torch((N, F), x.device, reads=(x,)) if want_P else None
    with torch.cuda.device(x.device):
        args = (ptr(g.rowptr), ptr(g.col), ptr(g.val), N, reduce, ptr(x), x.stride(0), F,
                ptr(S), S.stride(0) if S is not None else 0, float(self_scale), ptr(Wc),
                Wc.stride(0), d, ptr(b), _lib.ACT_RELU if relu else _lib.ACT_NONE, ptr(defer_act),
                ptr(P), P.stride(0) if P is not None else 0, ptr(out), out.stride(0), ptr(Wsp))
        if residual is None:
            check(L.mp_agg_dense_f32(*args, _stream()), "mp_agg_dense_f32")
        else:   # out = act(... + residual); the residual may be `out` itself
            check(L.mp_agg_dense_add_f32(*args, ptr(residual), residual.stride(0), _stream()), "mp_agg_dense_add_f32")
    return out, P


def _raw_dense_wgrad(P, G, want_bias=False):
    """P^T @ G on the engine's split-K MFMA kernel (None when the shape is outside it); with want_bias the pair
    (P^T @ G, column sums of G) — the bias gradient comes out of the same pass over G"""
    L = lib()
    M, F = P.shape
    d = G.size(1)
    out = torch.empty((F, d), dtype=torch.float32, device=P.device)
    db = torch.empty(d, dtype=torch.float32, device=P.device) if want_bias else None
    with torch.cuda.device(P.device):
        nb = C.c_size_t(0)
        check(L.mp_dense_wgrad_ws_bytes(M, F, d, C.byref(nb)))
        ws = torch.empty(max(nb.value, 1), dtype=torch.uint8, device=P.device)
        st = L.mp_dense_wgrad_f32(ptr(P), P.stride(0), ptr(G), G.stride(0), M, F, d, ptr(out), ptr(db), ptr(ws),
                                  nb.value, _stream())
    if st in (2, 5):
        return (None, None) if want_bias else None
    check(st, "mp_dense_wgrad_f32")
    return (out, db) if want_bias else out


def _raw_dense_wgrad_relu(P, G, Y, want_bias=False, want_gm=True, gm_out=None):
    """(P^T (G * [Y > 0]), its column sums or None, G * [Y > 0]) in one pass (mp_dense_wgrad_relu_f32): the weight-gradient
    kernel masks the incoming gradient by the forward's ReLU pattern as it reads it and writes the masked gradient out
    for the input-gradient launch; None when the shape is outside the kernel"""
    L = lib()
    M, F = P.shape
    d = G.size(1)
    out = torch.empty((F, d), dtype=torch.float32, device=P.device)
    db = torch.empty(d, dtype=torch.float32, device=P.device) if want_bias else None
    # the masked gradient is written only when an input-gradient launch will read it (a first layer has none)
    # (gm_out: a [M, d] view to receive it, e.g. one half of a concatenated gradient)
    gm = gm_out if gm_out is not None else (
        placement.empty_or_torch((M, d), P.device, reads=(G, Y, P), streaming=True) if want_gm else None)
    with torch.cuda.device(P.device):
        nb = C.c_size_t(0)
        check(L.mp_dense_wgrad_ws_bytes(M, F, d, C.byref(nb)))
        ws = torch.empty(max(nb.value, 1), dtype=torch.uint8, device=P.device)
        st = L.mp_dense_wgrad_relu_f32(ptr(P), P.stride(0), ptr(G), G.stride(0), ptr(Y), Y.stride(0), ptr(gm),
                                       gm.stride(0) if gm is not None else 0, M, F, d, ptr(out), ptr(db), ptr(ws),
                                       nb.value, _stream())
    if st in (2, 5):
        return None
    check(st, "mp_dense_wgrad_relu_f32")
    return out, db, gm


def _wgrad_and_bias(X, g, need_w, need_b):
    """(X^T g, sum_m g[m]) for a transform's backward pass: one kernel when both are wanted"""
    if need_w and need_b:
        dW, db = _raw_dense_wgrad(X, g, want_bias=True)
        if dW is not None:
            return dW, db
    dW = None
    if need_w:
        dW = _raw_dense_wgrad(X, g)
        if dW is None:
            dW = X.t() @ g
    return dW, (g.sum(0) if need_b else None)


def _dense_into(out_view, P, W, bias, relu):
    """out_view[:, :] = act(P @ W + bias) written in place through the kernel's output leading dimension"""
    L = lib()
    M, F = P.shape
    d = W.size(1)
    if dense_x3_supported(P, F, d, out_view):
        _raw_dense_x3(P, W, bias, relu, out=out_view)
        return
    Wc = W.contiguous()
    b = None if bias is None else bias.contiguous()
    with torch.cuda.device(P.device):
        check(L.mp_dense_fused_f32(ptr(P), P.stride(0), ptr(Wc), None, 0, None, ptr(b),
                                   _lib.ACT_RELU if relu else _lib.ACT_NONE, ptr(out_view), out_view.stride(0), M, F, d,
                                   _stream()), "mp_dense_fused_f32")


class _ConcatDense(torch.autograd.Function):
    """out = act([x @ Ws ‖ m @ Wn] + bias): both halves written straight into one buffer (no cat, no separate
    bias / activation passes) — the combine step of tfg MeanGraphSage / IDSAGE.call (TfgIDLayer.py:100-117)"""
    @staticmethod
    def forward(ctx, x, m, Ws, Wn, bias, relu):
        x, m = _f32c(x, "x"), _f32c(m, "m")
        ku, kn = Ws.size(1), Wn.size(1)
        out = placement.empty_or_torch((x.size(0), ku + kn), x.device, reads=(x, m), streaming=True)
        b = None if bias is None else bias.detach()
        _dense_into(out[:, :ku], x, Ws.detach(), None if b is None else b[:ku], relu)
        _dense_into(out[:, ku:], m, Wn.detach(), None if b is None else b[ku:], relu)
        ctx.relu, ctx.ku, ctx.has_bias = relu, ku, bias is not None
        ctx.save_for_backward(x, m, Ws, Wn, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, m, Ws, Wn, out = ctx.saved_tensors
        ku = ctx.ku
        g = g.contiguous()
        need_in = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        done = False
        if ctx.relu and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]:
            # the ReLU mask rides in the two weight-gradient passes (one per half of the output); the masked halves are
            # written only when an input gradient will read them (a first layer has none)
            mg = placement.empty_or_torch(tuple(g.shape), g.device, reads=(g, out), streaming=True) if need_in else None
            rs = _raw_dense_wgrad_relu(x, g[:, :ku], out[:, :ku], want_bias=ctx.has_bias, want_gm=need_in,
                                       gm_out=None if mg is None else mg[:, :ku])
            rn = None if rs is None else _raw_dense_wgrad_relu(m, g[:, ku:], out[:, ku:], want_bias=ctx.has_bias,
                                                               want_gm=need_in,
                                                               gm_out=None if mg is None else mg[:, ku:])
            if rs is not None and rn is not None:
                (dWs, dbs, _), (dWn, dbn, _) = rs, rn
                g, done = mg, True
        if not done:
            if ctx.relu:
                g = torch.ops.aten.threshold_backward(g, out, 0.0)
            dWs, dbs = _wgrad_and_bias(x, g[:, :ku], ctx.needs_input_grad[2], ctx.has_bias)
            dWn, dbn = _wgrad_and_bias(m, g[:, ku:], ctx.needs_input_grad[3], ctx.has_bias)
        dx = times_wt(g[:, :ku], Ws) if ctx.needs_input_grad[0] else None     # strided views: the kernels take leading dimensions
        dm = times_wt(g[:, ku:], Wn) if ctx.needs_input_grad[1] else None
        db = torch.cat([dbs, dbn]) if ctx.has_bias else None
        return dx, dm, dWs, dWn, db, None


def concat_dense(x, m, Ws, Wn, bias=None, relu=False):
    return _ConcatDense.apply(x, m, Ws, Wn, bias, bool(relu))


class _SageConcatFused(torch.autograd.Function):
    """out = act([x Ws ‖ mean_j(x_j) Wn] + bias) (TfgIDLayer.py:100-117): the self half is one MFMA kernel, the
    neighbour half is the one-kernel aggregate -> transform writing into the same buffer; the backward pass
    runs the aggregate -> transform kernel on the transposed mean operator."""
    @staticmethod
    def forward(ctx, x, Ws, Wn, bias, g, relu, grad_mode):
        x = _f32c(x, "x")
        ku, kn = Ws.size(1), Wn.size(1)
        out = placement.empty_or_torch((x.size(0), ku + kn), x.device, reads=(x,))
        b = None if bias is None else bias.detach()
        _dense_into(out[:, :ku], x, Ws.detach(), None if b is None else b[:ku], relu)
        _, P = _raw_agg_dense(g, x, Wn.detach(), None if b is None else b[ku:], relu,
                              want_P=ctx.needs_input_grad[2] and grad_mode, reduce=_lib.MEAN,
                              out=out[:, ku:])
        ctx.g, ctx.relu, ctx.ku, ctx.has_bias = g, relu, ku, bias is not None
        ctx.save_for_backward(x, P, Ws, Wn, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, P, Ws, Wn, out = ctx.saved_tensors
        ku = ctx.ku
        gm = gout.contiguous()
        done = False
        if ctx.relu and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and P is not None:
            # the ReLU mask rides in the two weight-gradient passes (one per half of the concatenated output), which
            # also leave the masked halves in one buffer for the input-gradient launches: no threshold_backward pass
            need_gm = ctx.needs_input_grad[0]
            mg = placement.empty_or_torch(tuple(gm.shape), gm.device, reads=(gm, out), streaming=True) if need_gm else None
            rs = _raw_dense_wgrad_relu(x, gm[:, :ku], out[:, :ku], want_bias=ctx.has_bias, want_gm=need_gm,
                                       gm_out=None if mg is None else mg[:, :ku])
            rn = None if rs is None else _raw_dense_wgrad_relu(P, gm[:, ku:], out[:, ku:], want_bias=ctx.has_bias,
                                                               want_gm=need_gm,
                                                               gm_out=None if mg is None else mg[:, ku:])
            if rs is not None and rn is not None:
                (dWs, dbs, _), (dWn, dbn, _) = rs, rn
                gm, done = mg, True
        if not done:
            if ctx.relu:
                gm = torch.ops.aten.threshold_backward(gm, out, 0.0)
            dWs, dbs = _wgrad_and_bias(x, gm[:, :ku], ctx.needs_input_grad[1], ctx.has_bias)
            dWn, dbn = _wgrad_and_bias(P, gm[:, ku:], ctx.needs_input_grad[2], ctx.has_bias)
        gs, gn = (gm[:, :ku], gm[:, ku:]) if gm is not None else (None, None)
        db = torch.cat([dbs, dbn]) if ctx.has_bias else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = times_wt(gs, Ws)
            gt = ctx.g.transpose_mean()
            Wnt = Wn.detach().t().contiguous()
            if agg_dense_supported(gt, gn, Wnt) and dx.stride(1) == 1 and dx.stride(0) % 2 == 0:
                # dx = gs Ws^T + (A^T gn) Wn^T: the second term's launch adds the first as it stores
                _raw_agg_dense(gt, gn, Wnt, out=dx, residual=dx)
            else:
                T, _ = _raw_spmm(gt, gn.contiguous(), _lib.SUM)
                dx.add_(T @ Wnt)
        return dx, dWs, dWn, db, None, None, None


def sage_concat(g, x, Ws, Wn, bias=None, relu=False):
    """act([x Ws ‖ mean-aggregate(x) Wn] + bias); one-kernel aggregate -> transform for the neighbour half when
    the shapes allow, else the aggregation kernel + concat_dense"""
    if (agg_dense_supported(g, x, Wn) and x.dtype == torch.float32 and Ws.size(1) % 2 == 0
            and g.num_cols == g.num_nodes):
        return _SageConcatFused.apply(x, Ws, Wn, bias, g, bool(relu), torch.is_grad_enabled())
    return concat_dense(x, spmm(g, x, "mean"), Ws, Wn, bias, relu=relu)


class _IndexAddRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, id_index, u):
        h = _f32c(h, "h").clone()
        u = _f32c(u, "u")
        L = lib()
        ids = id_index.to(torch.int64).contiguous()
        with torch.cuda.device(h.device):
            check(L.mp_rows_scatter_add_f32(ptr(h), h.stride(0), ptr(ids), ids.numel(), h.size(1),
                                            ptr(u), u.stride(0), _stream()), "mp_rows_scatter_add_f32")
        ctx.save_for_backward(ids)
        return h

    @staticmethod
    def backward(ctx, dh):
        (ids,) = ctx.saved_tensors
        return dh, None, gather_rows(dh.contiguous(), ids)


def index_add_rows(h, id_index, u):
    """out = h ; out[id[k]] += u[k]   (tensor_scatter_nd_add, TfgIDLayer.py:107,165,330,515;
    index_add_, idconv.py:67,155,251,310,375)"""
    return _IndexAddRows.apply(h, id_index, u)


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ids):
        x = _f32c(x, "x")
        L = lib()
        ids = ids.to(torch.int64).contiguous()
        out = torch.empty((ids.numel(), x.size(1)), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(L.mp_rows_gather_f32(ptr(x), x.stride(0), ptr(ids), ids.numel(), x.size(1), ptr(out),
                                       out.stride(0), _stream()), "mp_rows_gather_f32")
        ctx.save_for_backward(ids)
        ctx.n = x.size(0)
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        dx = torch.zeros((ctx.n, dout.size(1)), dtype=torch.float32, device=dout.device)
        L = lib()
        dout = dout.contiguous()
        with torch.cuda.device(dout.device):
            check(L.mp_rows_scatter_add_f32(ptr(dx), dx.stride(0), ptr(ids), ids.numel(), dx.size(1),
                                            ptr(dout), dout.stride(0), _stream()))
        return dx, None


def gather_rows(x, ids):
    """x[ids]  (tf.gather / index_select of the identity rows)"""
    return _GatherRows.apply(x, ids)


# ---- attention pieces --------------------------------------------------------

def _raw_sddmm_dot(g, A, B, heads, scale):
    L = lib()
    s = torch.empty(max(g.nnz, 1) * heads, dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        st = L.mp_sddmm_dot_stream_f32(ptr(g.row_ids()), ptr(g.col), g.nnz, ptr(A), A.stride(0), ptr(B),
                                       B.stride(0), A.size(1), heads, float(scale), ptr(s), _stream())
        if st == 2:   # head layout the entry-balanced kernel does not cover
            st = L.mp_sddmm_dot_f32(ptr(g.rowptr), ptr(g.col), g.num_nodes, g.nnz, ptr(A), A.stride(0),
                                    ptr(B), B.stride(0), A.size(1), heads, float(scale), ptr(s), _stream())
        check(st, "mp_sddmm_dot")
    return s[:g.nnz * heads].view(g.nnz, heads)


def _raw_spmm_heads(g, a, V, heads):
    if heads == 1:   # one weight per entry: this is the hot aggregation kernel with val = a
        y, _ = _raw_spmm(g.with_values(a.reshape(-1).contiguous()), V, _lib.SUM)
        return y
    if heads in (2, 4, 8) and V.size(1) % heads == 0:
        # all heads in ONE launch of the hot kernel: full-row loads of V, every lane applies the weight of the head its
        # columns belong to (mp_spmm_csr_heads_f32); round 1 ran one launch per head on column slices
        L = lib()
        N, d = g.num_nodes, V.size(1)
        a = a.contiguous()
        y = placement.empty_or_torch((N, d), V.device, reads=(V,))
        plan, counts = g.plan()
        with torch.cuda.device(V.device):
            nb = C.c_size_t(0)
            check(L.mp_spmm_ws_bytes(counts, d, _lib.SUM, 0, C.byref(nb)))
            ws = torch.empty(nb.value, dtype=torch.uint8, device=V.device) if nb.value else None
            st = L.mp_spmm_csr_heads_f32(ptr(g.rowptr), ptr(g.col), ptr(a), N, ptr(plan), counts, heads, ptr(V),
                                         V.stride(0), ptr(y), y.stride(0), d, ptr(ws), nb.value, _stream())
        if st == 0:
            return y
        if st != 2:
            check(st, "mp_spmm_csr_heads_f32")
    if V.size(1) % heads == 0:
        # head layouts outside the one-launch kernel: one launch of the hot kernel per head on column slices
        dh = V.size(1) // heads
        y = torch.empty((g.num_nodes, V.size(1)), dtype=torch.float32, device=V.device)
        for h in range(heads):
            _raw_spmm(g.with_values(a[:, h].contiguous()), V[:, h * dh:(h + 1) * dh], _lib.SUM,
                      out=y[:, h * dh:(h + 1) * dh])
        return y
    L = lib()
    y = torch.empty((g.num_nodes, V.size(1)), dtype=torch.float32, device=V.device)
    with torch.cuda.device(V.device):
        check(L.mp_spmm_heads_f32(ptr(g.rowptr), ptr(g.col), ptr(a), g.num_nodes, heads, ptr(V),
                                  V.stride(0), ptr(y), y.stride(0), V.size(1), _stream()),
              "mp_spmm_heads_f32")
    return y


class _SddmmDot(torch.autograd.Function):
    """s[e,h] = scale * <Q[row_e, slice h], K[col_e, slice h]>"""
    @staticmethod
    def forward(ctx, Q, K, g, heads, scale):
        Q, K = _f32c(Q, "Q"), _f32c(K, "K")
        ctx.g, ctx.heads, ctx.scale = g, heads, scale
        ctx.save_for_backward(Q, K)
        return _raw_sddmm_dot(g, Q, K, heads, scale)

    @staticmethod
    def backward(ctx, ds):
        Q, K = ctx.saved_tensors
        g, heads = ctx.g, ctx.heads
        ds = (ds * ctx.scale).contiguous()
        # dQ[i, slice h] = sum_e ds[e,h] K[col_e, slice h]   (an aggregation over in-edges)
        dQ = _raw_spmm_heads(g, ds, K, heads)
        # dK[j, slice h] = sum_{e: col_e = j} ds[e,h] Q[row_e, slice h]   (over out-edges)
        gt = g._transpose_sorted()     # (per-entry values are permuted through gt.pos: also when A^T = A)
        dK = _raw_spmm_heads(gt, ds[gt.pos.long()].contiguous(), Q, heads)
        return dQ, dK, None, None, None


def sddmm_dot(g, Q, K, heads=1, scale=1.0):
    return _SddmmDot.apply(Q, K, g, int(heads), float(scale))


class _SddmmAdd(torch.autograd.Function):
    """s[e] = leaky_relu(ai[row_e] + aj[col_e])"""
    @staticmethod
    def forward(ctx, ai, aj, g, slope):
        L = lib()
        ai, aj = ai.contiguous(), aj.contiguous()
        s = torch.empty(max(g.nnz, 1), dtype=torch.float32, device=ai.device)
        with torch.cuda.device(ai.device):
            check(L.mp_sddmm_add_f32(ptr(g.rowptr), ptr(g.col), g.num_nodes, g.nnz, ptr(ai), ptr(aj),
                                     float(slope), ptr(s), _stream()), "mp_sddmm_add_f32")
        s = s[:g.nnz]
        ctx.g, ctx.slope = g, slope
        ctx.save_for_backward(s)
        return s.view(g.nnz, 1)

    @staticmethod
    def backward(ctx, ds):
        (s,) = ctx.saved_tensors
        g = ctx.g
        gs = ds.reshape(-1) * torch.where(s > 0, torch.ones_like(s), torch.full_like(s, ctx.slope))
        dai = torch.zeros(g.num_nodes, dtype=torch.float32, device=s.device)
        daj = torch.zeros(g.num_nodes, dtype=torch.float32, device=s.device)
        dai.index_add_(0, g.row_ids().long(), gs)
        daj.index_add_(0, g.col.long(), gs)
        return dai, daj, None, None


def sddmm_add(g, ai, aj, slope=0.2):
    return _SddmmAdd.apply(ai, aj, g, float(slope))


class _GatAlpha(torch.autograd.Function):
    """alpha[e, h] = softmax over destination row of leaky_relu(a_dst[row_e, h] + a_src[col_e, h]): one launch for all
    heads, scores never stored (mp_gat_alpha_f32)"""
    @staticmethod
    def forward(ctx, a_dst, a_src, g, slope):
        L = lib()
        ad = a_dst.contiguous().float()
        asr = a_src.contiguous().float()
        H = ad.size(1)
        alpha = torch.empty((max(g.nnz, 1), H), dtype=torch.float32, device=ad.device)
        with torch.cuda.device(ad.device):
            check(L.mp_gat_alpha_f32(ptr(g.rowptr), ptr(g.col), g.num_nodes, g.nnz, H, ptr(ad), ptr(asr), float(slope),
                                     ptr(alpha), _stream()), "mp_gat_alpha_f32")
        alpha = alpha[:g.nnz]
        ctx.g, ctx.slope = g, slope
        ctx.save_for_backward(alpha, ad, asr)
        return alpha

    @staticmethod
    def backward(ctx, dalpha):
        alpha, ad, asr = ctx.saved_tensors
        g, L = ctx.g, lib()
        H = alpha.size(1)
        dalpha = dalpha.contiguous()
        ds = torch.empty_like(alpha)
        with torch.cuda.device(alpha.device):
            check(L.mp_csr_row_softmax_bwd_f32(ptr(g.rowptr), g.num_nodes, H, ptr(alpha), ptr(dalpha), ptr(ds),
                                               _stream()), "mp_csr_row_softmax_bwd_f32")
        rows, cols = g.row_ids().long(), g.col.long()
        pre = ad[rows] + asr[cols]
        ds = ds * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, ctx.slope))
        d_dst = torch.zeros_like(ad).index_add_(0, rows, ds)
        d_src = torch.zeros_like(asr).index_add_(0, cols, ds)
        return d_dst, d_src, None, None


def gat_alpha(g, a_dst, a_src, slope=0.2):
    """additive attention coefficients [nnz, H] for per-node terms a_dst, a_src [N, H] (idconv.py:319-327)"""
    return _GatAlpha.apply(a_dst, a_src, g, float(slope))


class _EdgeSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, g):
        L = lib()
        s = s.contiguous()
        heads = s.size(1)
        p = torch.empty_like(s)
        with torch.cuda.device(s.device):
            check(L.mp_csr_row_softmax_f32(ptr(g.rowptr), g.num_nodes, heads, ptr(s), ptr(p), _stream()),
                  "mp_csr_row_softmax_f32")
        ctx.g = g
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        L = lib()
        g = ctx.g
        dp = dp.contiguous()
        ds = torch.empty_like(p)
        with torch.cuda.device(p.device):
            check(L.mp_csr_row_softmax_bwd_f32(ptr(g.rowptr), g.num_nodes, p.size(1), ptr(p), ptr(dp),
                                               ptr(ds), _stream()), "mp_csr_row_softmax_bwd_f32")
        return ds, None


def edge_softmax(g, s):
    """softmax of the per-entry scores over each destination row, per head; s [nnz, H]"""
    return _EdgeSoftmax.apply(s, g)


class _SpmmEdgeValues(torch.autograd.Function):
    """y[i, slice h] = sum_e a[e,h] V[col_e, slice h], differentiable in a and V"""
    @staticmethod
    def forward(ctx, a, V, g, heads):
        a, V = a.contiguous(), _f32c(V, "V")
        ctx.g, ctx.heads = g, heads
        ctx.save_for_backward(a, V)
        return _raw_spmm_heads(g, a, V, heads)

    @staticmethod
    def backward(ctx, dy):
        a, V = ctx.saved_tensors
        g, heads = ctx.g, ctx.heads
        dy = dy.contiguous()
        da = _raw_sddmm_dot(g, dy, V, heads, 1.0)
        gt = g._transpose_sorted()     # (per-entry values are permuted through gt.pos: also when A^T = A)
        dV = _raw_spmm_heads(gt, a[gt.pos.long()].contiguous(), dy, heads)
        return da, dV, None, None


def spmm_edge_values(g, a, V, heads=1):
    return _SpmmEdgeValues.apply(a, V, g, int(heads))


# =========================================================================================
# Registered operators: torch.ops.mp.*
#
# The path's operators as PyTorch custom ops (torch.library): schemas, fake (meta) kernels for tracing /
# torch.compile / opcheck, autograd formulas built from other registered ops.  A graph crosses the boundary as
# an int handle (CSRGraph.handle): custom ops take tensors and scalars only, and the CSR, its plan and its cached
# transpose belong to the batch, not to a call.  `*_raw` ops are single launches of a C-ABI entry point (no
# autograd); the un-suffixed ops are what the layers call.
#
#   mp::spmm            y = act(reduce_j w_ij x_j + s x_i + b)         SparseAdj.matmul, sparse_adj.py:91-97
#   mp::idgnn_agg       (P, Q) = (A x, A S x)                          gcn_id two-branch form, TfgIDLayer.py:510-517
#   mp::agg_dense       act((A x + s x) W + b), one launch             aggregate -> kernel product, TfgIDLayer.py:510-523
#   mp::agg_dense_id    act(A (x W + S x W_id) + b)                    gcn_id / GCNIDConvLayer, idconv.py:150-177
#   mp::dense_fused     act(P W [+ Q W_id] + b)                        the transform after the aggregation
#   mp::bn_act          BatchNorm1d (training statistics) [+ ReLU]     graphgym/models/layer.py:26-35
# =========================================================================================
from typing import Optional, Tuple   # noqa: E402

from torch.library import custom_op, register_autograd   # noqa: E402

from .graph import from_handle   # noqa: E402

Tensor = torch.Tensor


def _none_if_empty(t):
    return None if t is None or t.numel() == 0 else t


def _empty_like_none(ref, dtype=torch.float32):
    return torch.empty((0,), dtype=dtype, device=ref.device)


# ---- raw launches -------------------------------------------------------------------------
@custom_op("mp::spmm_raw", mutates_args=(), device_types="cuda")
def _op_spmm_raw(x: Tensor, graph: int, variant: int, reduce: int, S: Optional[Tensor], self_scale: float,
                 bias: Optional[Tensor], relu: bool, want_argmax: bool) -> Tuple[Tensor, Tensor]:
    g = from_handle(graph).variant(variant)
    x = _f32c(x, "x")
    if x.size(0) != g.num_cols:
        raise ValueError(f"x has {x.size(0)} rows, the operator has {g.num_cols} columns")
    Sc = None if S is None else _f32c(S, "S")
    y, argmax = _raw_spmm(g, x, reduce, S=Sc, self_scale=self_scale, bias=None if bias is None else bias.contiguous(),
                          relu=relu, want_argmax=want_argmax)
    return y, (argmax if argmax is not None else _empty_like_none(x, torch.int32))


@_op_spmm_raw.register_fake
def _(x, graph, variant, reduce, S, self_scale, bias, relu, want_argmax):
    g = from_handle(graph)
    n = g.num_nodes if variant == 0 else g.num_cols
    return (x.new_empty((n, x.size(1)), dtype=torch.float32),
            x.new_empty((n, x.size(1)) if want_argmax else (0,), dtype=torch.int32))


@custom_op("mp::spmm_rows_raw", mutates_args=(), device_types="cuda")
def _op_spmm_rows_raw(x: Tensor, graph: int, variant: int, rows: Tensor) -> Tensor:
    """sum aggregation over the listed rows only of a graph variant: an [len(rows), n] operator"""
    sub = from_handle(graph).variant(variant).select_rows(rows)
    y, _ = _raw_spmm(sub, _f32c(x, "x"), _lib.SUM)
    return y


@_op_spmm_rows_raw.register_fake
def _(x, graph, variant, rows):
    return x.new_empty((rows.numel(), x.size(1)), dtype=torch.float32)


@custom_op("mp::spmm_max_bwd_raw", mutates_args=(), device_types="cuda")
def _op_spmm_max_bwd_raw(dy: Tensor, argmax: Tensor, graph: int) -> Tensor:
    g = from_handle(graph)
    dy = dy.contiguous()
    N, d = dy.shape
    dx = torch.zeros((g.num_cols, d), dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        check(lib().mp_spmm_max_bwd_f32(ptr(g.col), ptr(g.val), ptr(argmax), ptr(dy), dy.stride(0), N, d, ptr(dx),
                                        dx.stride(0), _stream()), "mp_spmm_max_bwd_f32")
    return dx


@_op_spmm_max_bwd_raw.register_fake
def _(dy, argmax, graph):
    return dy.new_empty((from_handle(graph).num_cols, dy.size(1)))


@custom_op("mp::idgnn_agg_raw", mutates_args=(), device_types="cuda")
def _op_idgnn_agg_raw(x: Tensor, graph: int, id_index: Tensor) -> Tuple[Tensor, Tensor]:
    g = from_handle(graph)
    x = _f32c(x, "x")
    L = lib()
    N, d = g.num_nodes, x.size(1)
    P = placement.empty_or_torch((N, d), x.device, reads=(x,))
    Q = placement.empty_or_torch((N, d), x.device, reads=(x,))
    if (d in AGG_TILES_WIDTHS and N >= AGG_TILES_MIN_ROWS and os.environ.get("MP_AGG_TILES", "1") != "0"
            and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and g.nnz > 0 and id_index.numel() > 0
            and g.max_row_entries() <= FUSED_MAX_ROW):
        # the tile structure (round 4): P in the pass of mp_agg_rows_tiles_f32, which also writes the zero rows of Q;
        # the rows of Q next to an identity node come from their few identity entries, gathered from x[id] by a small
        # kernel the call launches behind the tile kernel
        global AGG_TILES_CALLS
        AGG_TILES_CALLS += 1
        br = g.id_branch(id_index)
        Z = x.index_select(0, id_index.to(torch.int64))
        with torch.cuda.device(x.device):
            check(L.mp_idgnn_agg_tiles_f32(ptr(g.rowptr), ptr(g.col), ptr(g.val), N, ptr(x), x.stride(0), d, ptr(br.defer),
                                           ptr(br.rows), ptr(br.crp), ptr(br.slot), ptr(br.val), br.n_rows, ptr(Z),
                                           Z.stride(0), ptr(P), P.stride(0), ptr(Q), Q.stride(0), _stream()),
                  "mp_idgnn_agg_tiles_f32")
        return P, Q
    col_marked = g.mark_ids(id_index)
    plan, counts = g.plan()
    with torch.cuda.device(x.device):
        nb = C.c_size_t(0)
        check(L.mp_spmm_ws_bytes(counts, d, _lib.SUM, 1, C.byref(nb)))
        ws = torch.empty(nb.value, dtype=torch.uint8, device=x.device) if nb.value else None
        check(L.mp_idgnn_agg_f32(ptr(g.rowptr), ptr(col_marked), ptr(g.val), N, ptr(plan), counts,
                                 ptr(x), x.stride(0), ptr(P), P.stride(0), ptr(Q), Q.stride(0), d,
                                 ptr(ws), nb.value, _stream()), "mp_idgnn_agg_f32")
    return P, Q


@_op_idgnn_agg_raw.register_fake
def _(x, graph, id_index):
    n = from_handle(graph).num_nodes
    return x.new_empty((n, x.size(1))), x.new_empty((n, x.size(1)))


def _agg_dense_kernel_ok(g, x, W, S=None):
    """the one-kernel aggregate -> transform covers these operands (shapes, alignment, row lengths)"""
    return (agg_dense_supported(g, x, W) and x.dtype == torch.float32
            and (S is None or (S.stride(0) % 4 == 0 and S.data_ptr() % 16 == 0)))


@custom_op("mp::agg_dense_raw", mutates_args=(), device_types="cuda")
def _op_agg_dense_raw(x: Tensor, W: Tensor, bias: Optional[Tensor], graph: int, variant: int, reduce: int,
                      S: Optional[Tensor], self_scale: float, relu: bool, want_P: bool) -> Tuple[Tensor, Tensor]:
    """act((reduce_j w_ij x_j + s S_i) W + b): ONE launch where mp_agg_dense_f32 covers the operands, otherwise the
    aggregation kernel followed by the fused transform; returns (out, aggregated rows or an empty tensor)"""
    g = from_handle(graph).variant(variant)
    x = _f32c(x, "x")
    Sc = None if S is None else _f32c(S, "S")
    Wd = W.detach()
    if _agg_dense_kernel_ok(g, x, Wd, Sc):
        out, P = _raw_agg_dense(g, x, Wd, None if bias is None else bias.detach(), relu, S=Sc,
                                self_scale=self_scale, want_P=want_P, reduce=reduce)
    else:
        P, _ = _raw_spmm(g, x, reduce, S=Sc, self_scale=self_scale)
        out = _dense_any(P, Wd, None, None, bias, relu)
    return out, (P if (want_P and P is not None) else _empty_like_none(x))


@_op_agg_dense_raw.register_fake
def _(x, W, bias, graph, variant, reduce, S, self_scale, relu, want_P):
    g = from_handle(graph)
    n = g.num_nodes if variant == 0 else g.num_cols
    return x.new_empty((n, W.size(1))), x.new_empty((n, x.size(1)) if want_P else (0,))


@custom_op("mp::agg_dense_id_raw", mutates_args=(), device_types="cuda")
def _op_agg_dense_id_raw(x: Tensor, W: Tensor, W_id: Tensor, bias: Optional[Tensor], graph: int, id_index: Tensor,
                         self_scale: float, relu: bool, want_P: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """act((A x + s x) W + b + A_id Z), Z = x[id] W_id: the one-kernel layer with the activation deferred on the rows
    next to an identity node, then mp_id_fixup_f32 on those rows; returns (out, aggregated rows or empty, x[id])"""
    g = from_handle(graph)
    x = _f32c(x, "x")
    br = g.id_branch(id_index)
    ids = id_index.to(torch.int64)
    x_id = x.index_select(0, ids)
    Z = torch.mm(x_id, W_id.detach())
    Wd = W.detach()
    S = x if self_scale != 0.0 else None
    b = None if bias is None else bias.detach()
    if _agg_dense_kernel_ok(g, x, Wd, S):
        out, P = _raw_agg_dense(g, x, Wd, b, relu, S=S, self_scale=self_scale, want_P=want_P, defer_act=br.defer)
        act = _lib.ACT_RELU if relu else _lib.ACT_NONE
    else:   # shapes outside the one-kernel layer: aggregation kernel, transform without activation, then the fix-up
        P, _ = _raw_spmm(g, x, _lib.SUM, S=S, self_scale=self_scale)
        out = _dense_any(P, Wd, None, None, b, False)
        act = _lib.ACT_NONE
    with torch.cuda.device(x.device):
        check(lib().mp_id_fixup_f32(ptr(br.rows), ptr(br.crp), ptr(br.slot), ptr(br.val), br.n_rows, ptr(Z),
                                    Z.stride(0), ptr(out), out.stride(0), out.size(1), act, _stream()),
              "mp_id_fixup_f32")
    if relu and act == _lib.ACT_NONE:
        out = torch.relu_(out)
    return out, (P if (want_P and P is not None) else _empty_like_none(x)), x_id


@_op_agg_dense_id_raw.register_fake
def _(x, W, W_id, bias, graph, id_index, self_scale, relu, want_P):
    n = from_handle(graph).num_nodes
    return (x.new_empty((n, W.size(1))), x.new_empty((n, x.size(1)) if want_P else (0,)),
            x.new_empty((id_index.numel(), x.size(1))))


@custom_op("mp::id_branch_t_raw", mutates_args=(), device_types="cuda")
def _op_id_branch_t_raw(gm: Tensor, graph: int, id_index: Tensor) -> Tensor:
    """T = A_id^T g  [n_id, d]: the gradient reaching Z = x[id] W_id"""
    br = from_handle(graph).id_branch(id_index)
    T, _ = _raw_spmm(br.t, _f32c(gm, "g"), _lib.SUM)
    return T


@_op_id_branch_t_raw.register_fake
def _(gm, graph, id_index):
    return gm.new_empty((id_index.numel(), gm.size(1)))


def _dense_any(P, W, Q, W_id, bias, relu):
    """act(P W [+ Q W_id] + b): the engine's MFMA kernel where it pays / applies, library GEMMs otherwise"""
    Pc = _f32c(P, "P")
    Qc = None if Q is None else _f32c(Q, "Q")
    out = None
    if Qc is None and bias is None and not relu and not dense_x3_supported(Pc, Pc.size(1), W.size(1)):
        # a plain product outside the streaming kernel's shapes has nothing to fuse: the library GEMM; the general
        # MFMA kernel earns its keep when bias / activation / a second product ride along
        return torch.mm(Pc, W)
    out = _raw_dense_fused(Pc, W, Qc, W_id, bias, relu)
    if out is None:     # shape outside the fused kernel: library GEMMs
        out = Pc @ W
        if Qc is not None:
            out = out + Qc @ W_id
        if bias is not None:
            out = out + bias
        if relu:
            out = torch.relu(out)
    return out


@custom_op("mp::dense_fused_raw", mutates_args=(), device_types="cuda")
def _op_dense_fused_raw(P: Tensor, W: Tensor, Q: Optional[Tensor], W_id: Optional[Tensor], bias: Optional[Tensor],
                        relu: bool) -> Tensor:
    return _dense_any(P, W.detach(), Q, None if W_id is None else W_id.detach(),
                      None if bias is None else bias.detach(), relu)


@_op_dense_fused_raw.register_fake
def _(P, W, Q, W_id, bias, relu):
    return P.new_empty((P.size(0), W.size(1)))


@custom_op("mp::dense_wgrad_raw", mutates_args=(), device_types="cuda")
def _op_dense_wgrad_raw(X: Tensor, G: Tensor, want_w: bool, want_b: bool) -> Tuple[Tensor, Tensor]:
    """(X^T G, column sums of G) in one pass over G (either may be skipped: an empty tensor comes back)"""
    dW, db = _wgrad_and_bias(X, G.contiguous() if G.stride(-1) != 1 else G, want_w, want_b)
    return (dW if dW is not None else _empty_like_none(G)), (db if db is not None else _empty_like_none(G))


@_op_dense_wgrad_raw.register_fake
def _(X, G, want_w, want_b):
    return (G.new_empty((X.size(1), G.size(1)) if want_w else (0,)), G.new_empty((G.size(1),) if want_b else (0,)))


@custom_op("mp::dense_wgrad_relu_raw", mutates_args=(), device_types="cuda")
def _op_dense_wgrad_relu_raw(X: Tensor, G: Tensor, Y: Tensor, want_b: bool,
                             want_gm: bool = True) -> Tuple[Tensor, Tensor, Tensor]:
    """(X^T gm, column sums of gm, gm) with gm = G * [Y > 0]: the ReLU backward folded into the weight-gradient pass;
    want_gm False: gm is not written (nobody reads it) and comes back empty"""
    Gc = G if (G.stride(-1) == 1 and G.dim() == 2) else G.contiguous()
    Yc = Y if Y.stride(-1) == 1 else Y.contiguous()
    Xc = X if X.stride(-1) == 1 else X.contiguous()
    r = _raw_dense_wgrad_relu(Xc, Gc, Yc, want_bias=want_b, want_gm=want_gm)
    if r is None:     # shape outside the kernel: separate passes
        gm = torch.ops.aten.threshold_backward(Gc, Yc, 0.0)
        dW, db = _wgrad_and_bias(Xc, gm, True, want_b)
        return dW, (db if db is not None else _empty_like_none(G)), (gm if want_gm else _empty_like_none(G))
    dW, db, gm = r
    return dW, (db if db is not None else _empty_like_none(G)), (gm if gm is not None else _empty_like_none(G))


@_op_dense_wgrad_relu_raw.register_fake
def _(X, G, Y, want_b, want_gm=True):
    return (G.new_empty((X.size(1), G.size(1))), G.new_empty((G.size(1),) if want_b else (0,)),
            G.new_empty(G.shape if want_gm else (0,)))


def _masked_grads(P, gout, out, relu, need_w, need_b, need_gm=True):
    """(gm, dW, db) for a transform with an optional ReLU epilogue: with ReLU and a weight gradient wanted the mask rides
    in the weight-gradient pass (one kernel); otherwise threshold_backward / plain weight gradient.  need_gm False
    (no input gradient will be taken): gm may come back None and is not written."""
    if relu and need_w and P is not None and P.numel() > 0:
        dW, db, gm = torch.ops.mp.dense_wgrad_relu_raw(P, gout, out, need_b, need_gm)
        return (gm if need_gm else None), dW, _none_if_empty(db)
    gm = gout.contiguous()
    if relu:
        gm = torch.ops.aten.threshold_backward(gm, out, 0.0)
    if need_w or need_b:
        dW, db = torch.ops.mp.dense_wgrad_raw(P, gm, need_w, need_b)
        return gm, _none_if_empty(dW), _none_if_empty(db)
    return gm, None, None


# ---- differentiable operators ---------------------------------------------------------------
@custom_op("mp::spmm", mutates_args=(), device_types="cuda")
def _op_spmm(x: Tensor, graph: int, reduce: int, self_scale: float, bias: Optional[Tensor],
             relu: bool) -> Tuple[Tensor, Tensor]:
    return torch.ops.mp.spmm_raw(x, graph, 0, reduce, x if self_scale != 0.0 else None, self_scale, bias, relu,
                                 reduce == _lib.MAX)


@_op_spmm.register_fake
def _(x, graph, reduce, self_scale, bias, relu):
    n = from_handle(graph).num_nodes
    return x.new_empty((n, x.size(1))), x.new_empty((n, x.size(1)) if reduce == _lib.MAX else (0,), dtype=torch.int32)


def _spmm_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)   # unused outputs (the saved rows, argmax) get no zero-filled gradient tensors
    x, graph, reduce, self_scale, bias, relu = inputs
    y, argmax = output
    ctx.graph, ctx.reduce, ctx.self_scale, ctx.relu = graph, reduce, self_scale, relu
    ctx.g_alive = from_handle(graph)
    ctx.has_bias = bias is not None
    ctx.save_for_backward(y if relu else None, argmax)


def _spmm_backward(ctx, dy, _dargmax):
    y, argmax = ctx.saved_tensors
    if dy is None:
        return None, None, None, None, None, None
    dy = dy.contiguous()
    if ctx.relu:
        dy = torch.ops.aten.threshold_backward(dy, y, 0.0)    # one vectorised pass
    dbias = dy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[4]) else None
    dx = None
    if ctx.needs_input_grad[0]:
        S = dy if ctx.self_scale != 0.0 else None
        if ctx.reduce == _lib.MAX:
            dx = torch.ops.mp.spmm_max_bwd_raw(dy, argmax, ctx.graph)
            if ctx.self_scale != 0.0:
                dx = dx + ctx.self_scale * dy
        else:   # the same kernel on the transposed operator (mean: entries w / count(row))
            dx = torch.ops.mp.spmm_raw(dy, ctx.graph, 1 if ctx.reduce == _lib.SUM else 2, _lib.SUM, S,
                                       ctx.self_scale, None, False, False)[0]
    return dx, None, None, None, dbias, None


register_autograd("mp::spmm", _spmm_backward, setup_context=_spmm_setup)


def spmm(g, x, reduce="sum", self_scale=0.0, bias=None, relu=False):
    """y[i] = act( reduce_{j in N(i)} w_ij x[j] + self_scale * x[i] + bias )   (torch.ops.mp.spmm)

    reduce: 'sum'/'add' | 'mean' | 'max'.  The gradient flows to x and bias; entry values
    of g are constants here (attention weights go through spmm_edge_values)."""
    _require_hip(x, "x")
    r = _lib.REDUCE[reduce]
    if r == _lib.MAX and not (torch.is_grad_enabled() and (x.requires_grad or (bias is not None and bias.requires_grad))):
        # nothing will be differentiated: no argmax written (and the tile kernel may take the launch)
        return torch.ops.mp.spmm_raw(x, g.handle, 0, r, x if self_scale != 0.0 else None, float(self_scale), bias,
                                     bool(relu), False)[0]
    return torch.ops.mp.spmm(x, g.handle, r, float(self_scale), bias, bool(relu))[0]


@custom_op("mp::idgnn_agg", mutates_args=(), device_types="cuda")
def _op_idgnn_agg(x: Tensor, graph: int, id_index: Tensor) -> Tuple[Tensor, Tensor]:
    return torch.ops.mp.idgnn_agg_raw(x, graph, id_index)


@_op_idgnn_agg.register_fake
def _(x, graph, id_index):
    n = from_handle(graph).num_nodes
    return x.new_empty((n, x.size(1))), x.new_empty((n, x.size(1)))


def _idgnn_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)   # unused outputs (the saved rows, argmax) get no zero-filled gradient tensors
    x, graph, id_index = inputs
    ctx.graph, ctx.g_alive = graph, from_handle(graph)
    ctx.save_for_backward(id_index)


def _idgnn_backward(ctx, dP, dQ):
    (id_index,) = ctx.saved_tensors
    if dP is None and dQ is None:
        return None, None, None
    if dP is None:                            # only Q was used downstream
        dP = torch.zeros_like(dQ)
    dx = torch.ops.mp.spmm_raw(dP.contiguous(), ctx.graph, 1, _lib.SUM, None, 0.0, None, False, False)[0]
    if dQ is None:
        return dx, None, None
    # Q = A S x  =>  dx[id] += (A^T dQ)[id]: only the identity nodes' rows of A^T are needed, an
    # aggregation over their out-edges alone (an [n_id, N] operator), not a second full pass
    t = torch.ops.mp.spmm_rows_raw(dQ.contiguous(), ctx.graph, 1, id_index)
    return dx.index_add(0, id_index.to(torch.int64), t), None, None


register_autograd("mp::idgnn_agg", _idgnn_backward, setup_context=_idgnn_setup)


def idgnn_aggregate(g, id_index, x, col_marked=None):
    """(P, Q) with P = A x and Q = A S x, S selecting the identity nodes' rows: one pass over
    the edges.  P @ W + Q @ W_id equals A (x W + S x W_id) of gcn_id (TfgIDLayer.py:510-517)."""
    _require_hip(x, "x")
    return torch.ops.mp.idgnn_agg(x, g.handle, id_index)


@custom_op("mp::dense_fused", mutates_args=(), device_types="cuda")
def _op_dense_fused(P: Tensor, W: Tensor, Q: Optional[Tensor], W_id: Optional[Tensor], bias: Optional[Tensor],
                    relu: bool) -> Tensor:
    return torch.ops.mp.dense_fused_raw(P, W, Q, W_id, bias, relu)


@_op_dense_fused.register_fake
def _(P, W, Q, W_id, bias, relu):
    return P.new_empty((P.size(0), W.size(1)))


def _dense_setup(ctx, inputs, output):
    P, W, Q, W_id, bias, relu = inputs
    ctx.relu, ctx.has_q, ctx.has_bias = relu, Q is not None, bias is not None
    ctx.save_for_backward(P, W, Q, W_id, output if relu else None)


def _dense_backward(ctx, g):
    P, W, Q, W_id, out = ctx.saved_tensors
    need = ctx.needs_input_grad
    # the weight and bias gradients come out of one pass of the engine's split-K kernel, which also applies the ReLU
    # mask to g on the way; g @ W^T is the streaming transform with W^T (library GEMM outside its shapes)
    need_gm = need[0] or (ctx.has_q and (need[2] or need[3]))
    g, dW, db = _masked_grads(P, g, out, ctx.relu, need[1], ctx.has_bias and need[4], need_gm)
    dP = times_wt(g, W) if need[0] else None
    dQ = times_wt(g, W_id) if (ctx.has_q and need[2]) else None
    dWid = torch.ops.mp.dense_wgrad_raw(Q, g, True, False)[0] if (ctx.has_q and need[3]) else None
    return dP, dW, dQ, dWid, db, None


register_autograd("mp::dense_fused", _dense_backward, setup_context=_dense_setup)


def dense_fused(P, W, Q=None, W_id=None, bias=None, relu=False):
    """act(P @ W [+ Q @ W_id] + bias) in one kernel (torch.ops.mp.dense_fused; mp_dense_fused_f32)"""
    _require_hip(P, "P")
    return torch.ops.mp.dense_fused(P, W, Q, W_id, bias, bool(relu))


@custom_op("mp::agg_dense", mutates_args=(), device_types="cuda")
def _op_agg_dense(x: Tensor, W: Tensor, bias: Optional[Tensor], graph: int, reduce: int, self_scale: float,
                  relu: bool, want_P: bool) -> Tuple[Tensor, Tensor]:
    return torch.ops.mp.agg_dense_raw(x, W, bias, graph, 0, reduce, x if self_scale != 0.0 else None, self_scale, relu,
                                      want_P)


@_op_agg_dense.register_fake
def _(x, W, bias, graph, reduce, self_scale, relu, want_P):
    n = from_handle(graph).num_nodes
    return x.new_empty((n, W.size(1))), x.new_empty((n, x.size(1)) if want_P else (0,))


def _agg_dense_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)   # unused outputs (the saved rows, argmax) get no zero-filled gradient tensors
    x, W, bias, graph, reduce, self_scale, relu, want_P = inputs
    out, P = output
    ctx.graph, ctx.g_alive = graph, from_handle(graph)
    ctx.reduce, ctx.self_scale, ctx.relu, ctx.has_bias, ctx.want_P = reduce, self_scale, relu, bias is not None, want_P
    ctx.save_for_backward(P, W, out if relu else None, None if want_P else x)


def _agg_dense_backward(ctx, gout, _gP):
    P, W, out, x = ctx.saved_tensors
    need = ctx.needs_input_grad
    if gout is None:
        return None, None, None, None, None, None, None, None
    if not ctx.want_P and (need[1]):   # the aggregated rows were not kept (called outside grad mode bookkeeping)
        P = torch.ops.mp.spmm_raw(x, ctx.graph, 0, ctx.reduce, x if ctx.self_scale != 0.0 else None, ctx.self_scale,
                                  None, False, False)[0]
    gm, dW, db = _masked_grads(P, gout, out, ctx.relu, need[1], ctx.has_bias and need[2], need[0])
    dx = None
    if need[0]:
        # dx = (A^T g + s g) W^T: the same one-kernel layer on the transposed operator (mean: entries w / count)
        variant = 1 if ctx.reduce == _lib.SUM else 2
        dx = torch.ops.mp.agg_dense_raw(gm, W.t().contiguous(), None, ctx.graph, variant, _lib.SUM,
                                        gm if ctx.self_scale != 0.0 else None, ctx.self_scale, False, False)[0]
    return dx, dW, db, None, None, None, None, None


register_autograd("mp::agg_dense", _agg_dense_backward, setup_context=_agg_dense_setup)


def agg_dense(g, x, W, bias=None, relu=False, self_scale=0.0, reduce="sum"):
    """act((reduce_{j in N(i)} w_ij x[j] + self_scale * x[i]) W + bias)  (torch.ops.mp.agg_dense): aggregation and
    the feature transform that follows it in ONE kernel (mp_agg_dense_f32) when the shapes allow
    (agg_dense_supported), otherwise the aggregation kernel followed by the fused transform"""
    _require_hip(x, "x")
    if g.num_cols != g.num_nodes and self_scale != 0.0:
        raise ValueError("the self term needs a square operator")
    want_P = torch.is_grad_enabled() and W.requires_grad     # inference: no aggregated rows written
    return torch.ops.mp.agg_dense(x, W, bias, g.handle, _lib.REDUCE[reduce], float(self_scale), bool(relu), want_P)[0]


@custom_op("mp::agg_dense_id", mutates_args=(), device_types="cuda")
def _op_agg_dense_id(x: Tensor, W: Tensor, W_id: Tensor, bias: Optional[Tensor], graph: int, id_index: Tensor,
                     self_scale: float, relu: bool, want_P: bool) -> Tuple[Tensor, Tensor, Tensor]:
    return torch.ops.mp.agg_dense_id_raw(x, W, W_id, bias, graph, id_index, self_scale, relu, want_P)


@_op_agg_dense_id.register_fake
def _(x, W, W_id, bias, graph, id_index, self_scale, relu, want_P):
    n = from_handle(graph).num_nodes
    return (x.new_empty((n, W.size(1))), x.new_empty((n, x.size(1)) if want_P else (0,)),
            x.new_empty((id_index.numel(), x.size(1))))


def _agg_dense_id_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)   # unused outputs (the saved rows, argmax) get no zero-filled gradient tensors
    x, W, W_id, bias, graph, id_index, self_scale, relu, want_P = inputs
    out, P, x_id = output
    ctx.graph, ctx.g_alive = graph, from_handle(graph)
    ctx.self_scale, ctx.relu, ctx.has_bias, ctx.want_P = self_scale, relu, bias is not None, want_P
    ctx.save_for_backward(P, W, W_id, x_id, id_index, out if relu else None, None if want_P else x)


def _agg_dense_id_backward(ctx, gout, _gP, _gxid):
    P, W, W_id, x_id, id_index, out, x = ctx.saved_tensors
    need = ctx.needs_input_grad
    if gout is None:
        return None, None, None, None, None, None, None, None, None
    if not ctx.want_P and need[1]:
        P = torch.ops.mp.spmm_raw(x, ctx.graph, 0, _lib.SUM, x if ctx.self_scale != 0.0 else None, ctx.self_scale,
                                  None, False, False)[0]
    gm, dW, db = _masked_grads(P, gout, out, ctx.relu, need[1], ctx.has_bias and need[3], need[0] or need[2])
    # identity branch: T = A_id^T g [n_id, d_out];  dW_id = x_id^T T ;  dx[id] += T W_id^T
    T = torch.ops.mp.id_branch_t_raw(gm, ctx.graph, id_index) if (need[0] or need[2]) else None
    dWid = torch.mm(x_id.t(), T) if need[2] else None
    dx = None
    if need[0]:
        dx = torch.ops.mp.agg_dense_raw(gm, W.t().contiguous(), None, ctx.graph, 1, _lib.SUM,
                                        gm if ctx.self_scale != 0.0 else None, ctx.self_scale, False, False)[0]
        dx = dx.index_add(0, id_index.to(torch.int64), torch.mm(T, W_id.t()))
    return dx, dW, dWid, db, None, None, None, None, None


register_autograd("mp::agg_dense_id", _agg_dense_id_backward, setup_context=_agg_dense_id_setup)


def agg_dense_id(g, x, W, W_id, id_index, bias=None, relu=False, self_scale=0.0):
    """act(A (x W + S x W_id) + bias) with S selecting the identity nodes' rows (torch.ops.mp.agg_dense_id): one
    aggregate -> transform launch plus the identity branch's small product and row fix-up; None when the shapes are
    outside the one-kernel layer (the caller then runs idgnn_aggregate + dense_fused)"""
    _require_hip(x, "x")
    if not (agg_dense_supported(g, x, W) and x.dtype == torch.float32 and g.num_cols == g.num_nodes):
        return None
    want_P = torch.is_grad_enabled() and W.requires_grad
    return torch.ops.mp.agg_dense_id(x, W, W_id, bias, g.handle, id_index, float(self_scale), bool(relu), want_P)[0]

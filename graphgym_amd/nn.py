"""Linear (+ ReLU) and BatchNorm1d (+ ReLU) over the node axis on the engine's kernels (csrc/gemm.hip, csrc/bn.hip).

Drop-in for ``torch.nn.BatchNorm1d`` as GraphGym uses it after every conv (graphgym/models/layer.py:26-35)
and as the keras BatchNormalization in the TF path's GIN MLPs (main_zd.py:181-186): same parameter and buffer
names (weight, bias, running_mean, running_var, num_batches_tracked), same momentum / eps semantics.  In
training mode the statistics, the normalisation, the optional ReLU and the whole backward run as four
HBM-bound passes; torch's own kernel needs 0.5 s per backward call on a [10^7, 256] activation.
"""
import ctypes as C
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.library import custom_op, register_autograd

from . import placement
from ._lib import check, lib, ptr
from .graph import _require_hip, _stream

Tensor = torch.Tensor


def _bn_ws(N, d, device):
    nb = C.c_size_t(0)
    check(lib().mp_bn_ws_bytes(N, d, C.byref(nb)))
    return torch.empty(nb.value, dtype=torch.uint8, device=device), nb.value


@custom_op("mp::bn_fwd_raw", mutates_args=(), device_types="cuda")
def _op_bn_fwd_raw(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], eps: float,
                   relu: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """(y, mean, invstd, unbiased var) of BatchNorm1d over the node axis with batch statistics [+ ReLU]"""
    _require_hip(x, "x")
    x = x if (x.dtype == torch.float32 and x.stride(-1) == 1) else x.float().contiguous()
    N, d = x.shape
    y = placement.empty_or_torch((N, d), x.device, reads=(x,), streaming=True)
    mean = torch.empty(d, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    var_u = torch.empty_like(mean)
    w = None if weight is None else weight.detach().contiguous()
    b = None if bias is None else bias.detach().contiguous()
    with torch.cuda.device(x.device):
        ws, nb = _bn_ws(N, d, x.device)
        check(lib().mp_bn_train_fwd_f32(ptr(x), x.stride(0), N, d, ptr(w), ptr(b), float(eps), 1 if relu else 0,
                                        ptr(y), y.stride(0), ptr(mean), ptr(invstd), ptr(var_u), ptr(ws), nb,
                                        _stream()), "mp_bn_train_fwd_f32")
    return y, mean, invstd, var_u


@_op_bn_fwd_raw.register_fake
def _(x, weight, bias, eps, relu):
    d = x.size(1)
    return (x.new_empty(x.shape, dtype=torch.float32), x.new_empty((d,), dtype=torch.float32),
            x.new_empty((d,), dtype=torch.float32), x.new_empty((d,), dtype=torch.float32))


@custom_op("mp::bn_bwd_raw", mutates_args=(), device_types="cuda")
def _op_bn_bwd_raw(dy: Tensor, y: Optional[Tensor], x: Tensor, weight: Optional[Tensor], mean: Tensor,
                   invstd: Tensor, bias: Optional[Tensor] = None,
                   relu_from_x: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """(dx, dgamma, dbeta); y given = the forward's ReLU output (its mask is applied to dy on the fly); relu_from_x:
    the mask is recomputed from x, weight, bias, mean, invstd instead (mp_bn_train_bwd_relu_f32: y is not read)"""
    x = x if (x.dtype == torch.float32 and x.stride(-1) == 1) else x.float().contiguous()
    N, d = x.shape
    dy = dy.contiguous()
    dx = placement.empty_or_torch((N, d), x.device, reads=(dy, x), streaming=True)
    dgamma = torch.empty(d, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    w = None if weight is None else weight.detach().contiguous()
    with torch.cuda.device(x.device):
        ws, nb = _bn_ws(N, d, x.device)
        if relu_from_x:
            b = None if bias is None else bias.detach().contiguous()
            check(lib().mp_bn_train_bwd_relu_f32(ptr(dy), dy.stride(0), ptr(x), x.stride(0), N, d, ptr(w), ptr(b),
                                                 ptr(mean), ptr(invstd), ptr(dx), dx.stride(0), ptr(dgamma),
                                                 ptr(dbeta), ptr(ws), nb, _stream()), "mp_bn_train_bwd_relu_f32")
        else:
            check(lib().mp_bn_train_bwd_f32(ptr(dy), dy.stride(0), ptr(y), y.stride(0) if y is not None else 0,
                                            ptr(x), x.stride(0), N, d, ptr(w), ptr(mean), ptr(invstd), ptr(dx),
                                            dx.stride(0), ptr(dgamma), ptr(dbeta), ptr(ws), nb, _stream()),
                  "mp_bn_train_bwd_f32")
    return dx, dgamma, dbeta


@_op_bn_bwd_raw.register_fake
def _(dy, y, x, weight, mean, invstd, bias=None, relu_from_x=False):
    d = x.size(1)
    return (x.new_empty(x.shape, dtype=torch.float32), x.new_empty((d,), dtype=torch.float32),
            x.new_empty((d,), dtype=torch.float32))


@custom_op("mp::bn_act", mutates_args=(), device_types="cuda")
def _op_bn_act(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], eps: float,
               relu: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """graphgym/models/layer.py:26-35 (BatchNorm1d, then the activation) in training mode: (y, mean, invstd, var)"""
    return torch.ops.mp.bn_fwd_raw(x, weight, bias, eps, relu)


@_op_bn_act.register_fake
def _(x, weight, bias, eps, relu):
    d = x.size(1)
    return (x.new_empty(x.shape, dtype=torch.float32), x.new_empty((d,), dtype=torch.float32),
            x.new_empty((d,), dtype=torch.float32), x.new_empty((d,), dtype=torch.float32))


def _bn_setup(ctx, inputs, output):
    x, weight, bias, eps, relu = inputs
    y, mean, invstd, _ = output
    ctx.relu = relu
    ctx.has_affine = (weight is not None, bias is not None)
    # the ReLU mask of the backward pass is recomputed from x (bit for bit the forward's): y is not kept for it
    ctx.save_for_backward(x, weight, mean, invstd, bias if relu else None)


def _bn_backward(ctx, dy, _dmean, _dinvstd, _dvar):
    x, w, mean, invstd, b = ctx.saved_tensors
    dx, dgamma, dbeta = torch.ops.mp.bn_bwd_raw(dy, None, x, w, mean, invstd, b, ctx.relu)
    return dx, (dgamma if ctx.has_affine[0] else None), (dbeta if ctx.has_affine[1] else None), None, None


register_autograd("mp::bn_act", _bn_backward, setup_context=_bn_setup)


# ---- softmax cross-entropy over the labelled rows (graphgym/loss.py:53-68, 20-37) -------------------------------
def _check_ce_shapes(z, y, idx):
    """host-side shape contract (no device read): one label per selected row.  Label VALUES are checked on the device:
    a label outside [0, C) or an index outside the logit matrix is never dereferenced and turns the loss into NaN."""
    if z.dim() != 2:
        raise ValueError(f"logits must be [n_rows, C], got {tuple(z.shape)}")
    if idx is not None and idx.numel() != y.numel():
        raise ValueError(f"index selects {idx.numel()} rows but there are {y.numel()} labels")
    if idx is None and y.numel() > z.size(0):
        raise ValueError(f"{y.numel()} labels for {z.size(0)} rows of logits")


@custom_op("mp::softmax_ce_rows_raw", mutates_args=(), device_types="cuda")
def _op_ce_rows_raw(logits: Tensor, labels: Tensor, index: Optional[Tensor]) -> Tensor:
    _require_hip(logits, "logits")
    z = logits if (logits.dtype == torch.float32 and logits.stride(-1) == 1) else logits.float().contiguous()
    y = labels.to(torch.int64).contiguous()
    idx = None if index is None else index.to(torch.int64).contiguous()
    n_sel = y.numel()
    _check_ce_shapes(z, y, idx)
    out = torch.empty(n_sel, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        check(lib().mp_softmax_ce_rows_f32(ptr(z), z.stride(0), z.size(0), ptr(y), ptr(idx), n_sel, z.size(1), ptr(out),
                                           _stream()), "mp_softmax_ce_rows_f32")
    return out


@_op_ce_rows_raw.register_fake
def _(logits, labels, index):
    return logits.new_empty((labels.numel(),), dtype=torch.float32)


@custom_op("mp::softmax_ce_bwd_raw", mutates_args=(), device_types="cuda")
def _op_ce_bwd_raw(logits: Tensor, labels: Tensor, index: Optional[Tensor], gscale: Tensor, inv_n: float) -> Tensor:
    z = logits if (logits.dtype == torch.float32 and logits.stride(-1) == 1) else logits.float().contiguous()
    y = labels.to(torch.int64).contiguous()
    idx = None if index is None else index.to(torch.int64).contiguous()
    g = gscale.to(torch.float32).reshape(1).contiguous()
    _check_ce_shapes(z, y, idx)
    # with an index the rows accumulate onto zeros (a row listed twice gets both terms); without one every row of the
    # first n_sel is written once and the rest must still be zero
    d = torch.zeros_like(z) if (idx is not None or y.numel() < z.size(0)) else torch.empty_like(z)
    with torch.cuda.device(z.device):
        check(lib().mp_softmax_ce_bwd_f32(ptr(z), z.stride(0), z.size(0), ptr(y), ptr(idx), y.numel(), z.size(1), ptr(g),
                                          float(inv_n), ptr(d), d.stride(0), _stream()), "mp_softmax_ce_bwd_f32")
    return d


@_op_ce_bwd_raw.register_fake
def _(logits, labels, index, gscale, inv_n):
    return logits.new_empty(logits.shape, dtype=torch.float32)


@custom_op("mp::softmax_ce", mutates_args=(), device_types="cuda")
def _op_softmax_ce(logits: Tensor, labels: Tensor, index: Optional[Tensor]) -> Tensor:
    """mean over the labelled rows of softmax cross-entropy(logits[index], labels): one pass forward, one backward"""
    return torch.ops.mp.softmax_ce_rows_raw(logits, labels, index).mean()


@_op_softmax_ce.register_fake
def _(logits, labels, index):
    return logits.new_empty((), dtype=torch.float32)


def _ce_setup(ctx, inputs, output):
    logits, labels, index = inputs
    ctx.save_for_backward(logits, labels, index)


def _ce_backward(ctx, g):
    logits, labels, index = ctx.saved_tensors
    n = max(labels.numel(), 1)
    return torch.ops.mp.softmax_ce_bwd_raw(logits, labels, index, g, 1.0 / n), None, None


register_autograd("mp::softmax_ce", _ce_backward, setup_context=_ce_setup)


def softmax_cross_entropy(logits, labels, index=None):
    """F.cross_entropy(logits[index], labels, reduction='mean') on the engine (no gather copy, one pass each way);
    small problems stay with torch"""
    if (not logits.is_cuda or logits.dim() != 2 or labels.numel() * logits.size(1) < (1 << 20)
            or logits.dtype != torch.float32):
        sel = logits if index is None else logits[index]
        return F.cross_entropy(sel, labels, reduction="mean")
    return torch.ops.mp.softmax_ce(logits, labels, index)


class BatchNorm1d(nn.BatchNorm1d):
    """``nn.BatchNorm1d`` whose training-mode forward / backward run on the engine; ``relu=True`` fuses the
    activation that follows it in GraphGym's layer wrapper.  Eval mode uses the running statistics (torch)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, relu=False):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine,
                         track_running_stats=track_running_stats)
        self.relu = relu

    def forward(self, x):
        if x.dim() != 2:
            raise ValueError("expected [num_nodes, num_features]")
        use_batch_stats = self.training or not self.track_running_stats
        if not (use_batch_stats and x.is_cuda and x.size(0) > 1):
            y = super().forward(x)
            return torch.relu(y) if self.relu else y
        y, mean, _, var_u = torch.ops.mp.bn_act(x, self.weight, self.bias, float(self.eps), bool(self.relu))
        if self.training and self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(var_u, alpha=m)
        return y

    def extra_repr(self):
        return super().extra_repr() + f", relu={self.relu}"


class _NarrowHead(torch.autograd.Function):
    """y = x W^T + b for a narrow output (a classifier head, <= 16 classes) over many rows: the products with a
    [N, <= 16] side are the library's (they stream at the HBM rate), the weight gradient x^T g is the engine's
    narrow-output kernel (mp_dense_wgrad_f32: 10 GB at the HBM rate; the library's split-K GEMM takes 7.2 ms for it at
    10^7 x 256 x 10) with the bias gradient out of the same pass"""
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.mm(g, weight) if ctx.needs_input_grad[0] else None
        dW, db = ops._wgrad_and_bias(x, g, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        return dx, (None if dW is None else dW.t()), db


class Linear(nn.Linear):
    """torch.nn.Linear (same parameters, same state dict) whose forward is the engine's transform kernel with bias
    and an optional ReLU fused into the store — the keras Dense(d, relu) / Dense(d) of the TF path's MLPs
    (main_zd.py:181-186,214-225) and GraphGym's Linear -> ReLU -> Linear (idconv.py:432-435).  Backward: the weight
    and bias gradients come from the engine's split-K kernel, g W' from the streaming transform with W^T (the library GEMM
    outside its shapes)."""

    def __init__(self, in_features, out_features, bias=True, relu=False):
        super().__init__(in_features, out_features, bias=bias)
        self.relu = bool(relu)

    def forward(self, x):
        from . import ops
        # the engine's kernels pay off on wide outputs over many rows (scripts/linear_bench.py: forward + backward
        # 39.8 vs 45.0 ms at 10^7 x 256 -> 256, 18.6 vs 27.8 ms at 1 -> 256; the library wins at 256 -> 10 and on
        # small batches), so narrow heads and small inputs stay on the library
        if (x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and self.out_features <= 16 and not self.relu
                and self.in_features % 4 == 0 and x.size(0) >= (1 << 17) and x.stride(1) == 1 and x.stride(0) % 4 == 0
                and x.data_ptr() % 16 == 0 and torch.is_grad_enabled()):
            return _NarrowHead.apply(x, self.weight, self.bias)
        if (x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32 or self.out_features < 32
                or x.size(0) * self.out_features < (1 << 23)):
            y = F.linear(x, self.weight, self.bias)
            return torch.relu(y) if self.relu else y
        return ops.dense_fused(x, self.weight.t(), bias=self.bias, relu=self.relu)

    def extra_repr(self):
        return super().extra_repr() + (", relu=True" if self.relu else "")

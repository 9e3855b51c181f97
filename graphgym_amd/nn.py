"""Linear (+ ReLU) and BatchNorm1d (+ ReLU) over the node axis on the engine's kernels (csrc/gemm.hip, csrc/bn.hip).

Drop-in for ``torch.nn.BatchNorm1d`` as GraphGym uses it after every conv (graphgym/models/layer.py:26-35)
and as the keras BatchNormalization in the TF path's GIN MLPs (main_zd.py:181-186): same parameter and buffer
names (weight, bias, running_mean, running_var, num_batches_tracked), same momentum / eps semantics.  In
training mode the statistics, the normalisation, the optional ReLU and the whole backward run as four
HBM-bound passes; torch's own kernel needs 0.5 s per backward call on a [10^7, 256] activation.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import placement
from ._lib import check, lib, ptr
from .graph import _require_hip, _stream


class _BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu):
        _require_hip(x, "x")
        x = x if (x.dtype == torch.float32 and x.stride(-1) == 1) else x.float().contiguous()
        L = lib()
        N, d = x.shape
        y = placement.empty_or_torch((N, d), x.device, reads=(x,))
        mean = torch.empty(d, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        var_u = torch.empty_like(mean)
        w = None if weight is None else weight.detach().contiguous()
        b = None if bias is None else bias.detach().contiguous()
        with torch.cuda.device(x.device):
            nb = C.c_size_t(0)
            check(L.mp_bn_ws_bytes(N, d, C.byref(nb)))
            ws = torch.empty(nb.value, dtype=torch.uint8, device=x.device)
            check(L.mp_bn_train_fwd_f32(ptr(x), x.stride(0), N, d, ptr(w), ptr(b), float(eps), 1 if relu else 0,
                                        ptr(y), y.stride(0), ptr(mean), ptr(invstd), ptr(var_u), ptr(ws), nb.value,
                                        _stream()), "mp_bn_train_fwd_f32")
        ctx.relu = relu
        ctx.has_affine = (weight is not None, bias is not None)
        ctx.save_for_backward(x, w, mean, invstd, y if relu else None)
        ctx.mark_non_differentiable(mean, var_u)
        return y, mean, var_u

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        x, w, mean, invstd, y = ctx.saved_tensors
        L = lib()
        N, d = x.shape
        dy = dy.contiguous()
        dx = placement.empty_or_torch((N, d), x.device, reads=(dy, x))
        dgamma = torch.empty(d, dtype=torch.float32, device=x.device)
        dbeta = torch.empty_like(dgamma)
        with torch.cuda.device(x.device):
            nb = C.c_size_t(0)
            check(L.mp_bn_ws_bytes(N, d, C.byref(nb)))
            ws = torch.empty(nb.value, dtype=torch.uint8, device=x.device)
            check(L.mp_bn_train_bwd_f32(ptr(dy), dy.stride(0), ptr(y), y.stride(0) if y is not None else 0, ptr(x),
                                        x.stride(0), N, d, ptr(w), ptr(mean), ptr(invstd), ptr(dx), dx.stride(0),
                                        ptr(dgamma), ptr(dbeta), ptr(ws), nb.value, _stream()), "mp_bn_train_bwd_f32")
        return dx, (dgamma if ctx.has_affine[0] else None), (dbeta if ctx.has_affine[1] else None), None, None


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
        y, mean, var_u = _BatchNormAct.apply(x, self.weight, self.bias, self.eps, self.relu)
        if self.training and self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(var_u, alpha=m)
        return y

    def extra_repr(self):
        return super().extra_repr() + f", relu={self.relu}"


class Linear(nn.Linear):
    """torch.nn.Linear (same parameters, same state dict) whose forward is the engine's transform kernel with bias
    and an optional ReLU fused into the store — the keras Dense(d, relu) / Dense(d) of the TF path's MLPs
    (main_zd.py:181-186,214-225) and GraphGym's Linear -> ReLU -> Linear (idconv.py:432-435).  Backward: the weight
    and bias gradients come from the engine's split-K kernel, g W' (a plain product) from the library GEMM."""

    def __init__(self, in_features, out_features, bias=True, relu=False):
        super().__init__(in_features, out_features, bias=bias)
        self.relu = bool(relu)

    def forward(self, x):
        from . import ops
        # the engine's kernels pay off on wide outputs over many rows (scripts/linear_bench.py: forward + backward
        # 39.8 vs 45.0 ms at 10^7 x 256 -> 256, 18.6 vs 27.8 ms at 1 -> 256; the library wins at 256 -> 10 and on
        # small batches), so narrow heads and small inputs stay on the library
        if (x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32 or self.out_features < 32
                or x.size(0) * self.out_features < (1 << 23)):
            y = F.linear(x, self.weight, self.bias)
            return torch.relu(y) if self.relu else y
        return ops.dense_fused(x, self.weight.t(), bias=self.bias, relu=self.relu)

    def extra_repr(self):
        return super().extra_repr() + (", relu=True" if self.relu else "")

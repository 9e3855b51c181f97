"""The path's operators as PyTorch custom ops (torch.ops.mp.*): registered with schemas and fake kernels (CPU
suite: no device needed), torch.library.opcheck on the GPU (schema, fake tensors, autograd registration, AOT
dispatch of forward + backward)."""
import pytest
import torch

NAMES = ["spmm", "idgnn_agg", "agg_dense", "agg_dense_id", "dense_fused", "bn_act",
         "spmm_raw", "spmm_rows_raw", "spmm_max_bwd_raw", "idgnn_agg_raw", "agg_dense_raw", "agg_dense_id_raw",
         "id_branch_t_raw", "dense_fused_raw", "dense_wgrad_raw", "dense_wgrad_relu_raw", "bn_fwd_raw", "bn_bwd_raw",
         "softmax_ce", "softmax_ce_rows_raw", "softmax_ce_bwd_raw"]


def test_ops_are_registered_with_schemas():
    import graphgym_amd  # noqa: F401  (importing the package registers the ops)
    from graphgym_amd import nn as _nn, ops as _ops  # noqa: F401
    for n in NAMES:
        op = getattr(torch.ops.mp, n).default
        assert op._schema.name == "mp::" + n
    s = str(torch.ops.mp.spmm.default._schema)
    assert "Tensor x" in s and "graph" in s and "Tensor? bias" in s
    assert not torch.ops.mp.spmm.default._schema.is_mutable


def test_fake_kernels_give_shapes_without_a_device():
    """tracing (torch.compile, export, opcheck) sees shapes and dtypes through register_fake, no kernel runs"""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from graphgym_amd import graph, nn as _nn, ops as _ops  # noqa: F401
    g = graph.CSRGraph(torch.zeros(11, dtype=torch.int32), torch.zeros(0, dtype=torch.int32), None, None, 10, 0)
    with FakeTensorMode():
        x = torch.empty(10, 64, device="cuda")
        W = torch.empty(64, 32, device="cuda")
        Wi = torch.empty(64, 32, device="cuda")
        ids = torch.empty(3, dtype=torch.int64, device="cuda")
        y, am = torch.ops.mp.spmm(x, g.handle, 2, 0.0, None, False)
        assert y.shape == (10, 64) and am.shape == (10, 64) and am.dtype == torch.int32
        P, Q = torch.ops.mp.idgnn_agg(x, g.handle, ids)
        assert P.shape == Q.shape == (10, 64)
        out, Pk = torch.ops.mp.agg_dense(x, W, None, g.handle, 0, 1.0, True, False)
        assert out.shape == (10, 32) and Pk.numel() == 0
        out, Pk, xid = torch.ops.mp.agg_dense_id(x, W, Wi, None, g.handle, ids, 0.0, True, True)
        assert out.shape == (10, 32) and Pk.shape == (10, 64) and xid.shape == (3, 64)
        assert torch.ops.mp.dense_fused(x, W, x, Wi, None, True).shape == (10, 32)
        yb, mean, invstd, var = torch.ops.mp.bn_act(x, None, None, 1e-5, True)
        assert yb.shape == (10, 64) and mean.shape == invstd.shape == var.shape == (64,)
    with pytest.raises(Exception):
        from graphgym_amd.graph import from_handle
        from_handle(10 ** 9)                                  # a dead handle is an error, not a silent default


@pytest.mark.gpu
def test_opcheck(dev):
    import graphgym_amd as ga
    from graphgym_amd import nn as _nn, ops  # noqa: F401
    gen = torch.Generator().manual_seed(0)
    n, F, d = 300, 64, 32
    ei = torch.randint(0, n, (2, 4000), generator=gen)
    ei = torch.cat([ei, ei.flip(0)], 1)
    w = torch.rand(ei.size(1), generator=gen) + 0.1
    g = ga.CSRGraph.from_edge_index(ei.to(dev), n, w.to(dev))
    h = g.handle
    ids = torch.arange(0, n, 11, device=dev)

    def t(*shape, grad=True):
        return torch.randn(*shape, generator=gen).to(dev).requires_grad_(grad)
    cases = [
        (torch.ops.mp.spmm.default, (t(n, F), h, 0, 0.5, t(F), True)),
        (torch.ops.mp.spmm.default, (t(n, F), h, 1, 0.0, None, False)),
        (torch.ops.mp.spmm.default, (t(n, F), h, 2, 0.0, None, False)),
        (torch.ops.mp.idgnn_agg.default, (t(n, F), h, ids)),
        (torch.ops.mp.dense_fused.default, (t(n, F), t(F, d), t(n, F), t(F, d), t(d), True)),
        (torch.ops.mp.dense_fused.default, (t(n, F), t(F, d), None, None, None, False)),
        (torch.ops.mp.agg_dense.default, (t(n, F), t(F, d), t(d), h, 0, 1.0, True, True)),
        (torch.ops.mp.agg_dense.default, (t(n, F), t(F, d), None, h, 1, 0.0, False, True)),
        (torch.ops.mp.agg_dense_id.default, (t(n, F), t(F, d), t(F, d), t(d), h, ids, 0.0, True, True)),
        (torch.ops.mp.bn_act.default, (t(n, F), t(F), t(F), 1e-5, True)),
        (torch.ops.mp.spmm_raw.default, (t(n, F, grad=False), h, 1, 0, None, 0.0, None, False, False)),
        (torch.ops.mp.dense_wgrad_raw.default, (t(n, F, grad=False), t(n, d, grad=False), True, True)),
        (torch.ops.mp.dense_wgrad_relu_raw.default, (t(n, F, grad=False), t(n, d, grad=False),
                                                     torch.relu(t(n, d, grad=False)), True)),
        (torch.ops.mp.softmax_ce.default, (t(n, 7), torch.randint(0, 7, (n // 2,), generator=gen).to(dev),
                                           torch.randperm(n, generator=gen)[: n // 2].to(dev))),
    ]
    for op, args in cases:
        res = torch.library.opcheck(op, args, raise_exception=True)
        assert all(v == "SUCCESS" for v in res.values()), (op, res)


@pytest.mark.gpu
def test_layers_run_on_the_registered_ops(dev):
    """a layer's forward + backward dispatches torch.ops.mp.* (seen by a dispatch-mode tracer), not opaque Python"""
    import graphgym_amd as ga  # noqa: F401
    from graphgym_amd import layers as L
    from torch.utils._python_dispatch import TorchDispatchMode

    seen = set()

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if name.startswith("mp."):
                seen.add(name.split(".")[1])
            return func(*args, **(kwargs or {}))

    gen = torch.Generator().manual_seed(1)
    n, F = 400, 64
    ei = torch.randint(0, n, (2, 3000), generator=gen)
    ei = torch.cat([ei, ei.flip(0)], 1).to(dev)
    x = torch.randn(n, F, generator=gen).to(dev).requires_grad_(True)
    ids = torch.arange(0, n, 9, device=dev)
    layer = L.IDGCN(F, activation=torch.relu, in_features=F).to(dev)
    with Spy():
        layer([x, ei, ids]).sum().backward()
    assert "agg_dense_id" in seen or "agg_dense_id_raw" in seen, seen
    assert ({"dense_wgrad_raw", "dense_wgrad_relu_raw"} & seen) and "agg_dense_raw" in seen, seen   # the backward formula is registered ops too

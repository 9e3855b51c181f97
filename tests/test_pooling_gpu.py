"""global_{add,mean,max}_pool on the engine vs torch_scatter's semantics restated in the oracle
(graphgym/models/pooling.py:12-33), forward and backward, with and without the ego centre selection."""
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ego", [False, True])
@pytest.mark.parametrize("reduce", ["add", "mean", "max"])
def test_global_pool(dev, ego, reduce):
    from graphgym_amd import pooling
    from graphgym_amd.config import cfg
    g = torch.Generator().manual_seed(3)
    N, G, d = 700, 9, 48
    batch = torch.sort(torch.randint(0, G, (N,), generator=g)).values
    batch[batch == 4] = 5                                            # graph 4 is empty
    x = torch.randn(N, d, generator=g)
    ids = torch.randperm(N, generator=g)[:60]
    dy = torch.randn(G, d, generator=g)
    old = cfg.dataset.transform
    try:
        cfg.dataset.transform = 'ego' if ego else 'none'
        xg = x.to(dev).requires_grad_(True)
        out = pooling.pooling_dict[reduce](xg, batch.to(dev), ids.to(dev), size=G)
        out.backward(dy.to(dev))
    finally:
        cfg.dataset.transform = old
    xr = x.clone().requires_grad_(True)
    xs, bs = (xr[ids], batch[ids]) if ego else (xr, batch)
    if reduce == "max":   # route ties like the kernel: first entry in (graph, node) order
        order = torch.argsort(bs * N + (ids if ego else torch.arange(N)), stable=True)
        xs, bs = xs[order], bs[order]
    ref = R.scatter(xs, bs, G, reduce)
    ref.backward(dy)
    err = float((out.detach().cpu() - ref.detach()).abs().max())
    assert err <= 1e-5 * max(1.0, float(ref.detach().abs().max()))
    assert float((xg.grad.cpu() - xr.grad).abs().max()) <= 1e-5 * max(1.0, float(xr.grad.abs().max()))
    assert float(out.detach().cpu()[4].abs().max()) == 0.0           # empty graph -> 0

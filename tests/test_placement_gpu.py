"""The engine's placement of large outputs (graphgym_amd/placement.py, csrc/arena.hip): buffers are ordinary
tensors with tensor lifetime, the arena reuses what dies, and the pair (read matrix, placed output) times within
3 % of the best pair the arena offers."""
import ctypes as C
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GiB = 1 << 30


def _copy_ms(src, dst, reps=4):
    from graphgym_amd import placement
    return placement._probe(src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size(), reps)


def test_arena_tensors_behave_like_tensors(dev):
    from graphgym_amd import placement
    ar = placement.arena(dev)
    assert ar is not None, "no arena on a fresh MI355X box?"
    before = ar.stats()["in_use"]
    t = ar.empty((1 << 20, 64))                     # 256 MiB
    assert t.is_cuda and t.dtype == torch.float32 and t.shape == (1 << 20, 64) and t.is_contiguous()
    assert ar.owns(t) and ar.stats()["in_use"] >= before + t.numel() * 4
    t.fill_(2.0)
    v = t[5:9, :3]                                   # a view keeps the buffer alive
    del t
    gc.collect()
    assert ar.stats()["in_use"] >= before + (1 << 28)
    assert float(v.sum()) == 24.0
    u = (v * 2).sum()                                # ordinary torch ops on it
    assert float(u) == 48.0
    del v, u
    gc.collect()
    assert ar.stats()["in_use"] == before            # the deleter gave the range back
    # autograd through an arena tensor
    a = ar.empty((1024, 256))
    a.normal_()
    a.requires_grad_(True)
    (a * a).sum().backward()
    assert torch.allclose(a.grad, 2 * a.detach())
    # the conflict map is a symmetric matrix with a clear spread (there is something to place by)
    M = ar.conflict
    assert M.shape == (ar.n_gran, ar.n_gran) and np.allclose(M, M.T) and M.min() >= 0.0


def test_placed_output_is_within_3pct_of_the_best_pair(dev):
    """The thing placement is for, measured directly: the aggregation Y = A X (X = 4 GiB: 2^22 nodes x 256 fp32, BA graph)
    is timed with Y at EVERY free granule-aligned position of the arena (fastest of three launches each).
      * Y placed with verify="all" (what bench.py does for its resident output) is within 3 % of the best position;
      * Y placed by prediction alone (what every operator output gets: no timing at allocation) is within 3 % of the
        best as well;
    and the worst position must actually be slower (otherwise the box shows no placement effect and there is nothing to
    test: skipped).  Measured on the build's boxes: best 8.21 ms, median +0.45 %, worst +13 %, predicted +0.0-0.3 %,
    verified +0.2-0.3 % — the aggregation is slow exactly when Y shares X's blocks, and both placements avoid them."""
    import graphgym_amd as ga
    from graphgym_amd import graphgen, ops, placement
    ar = placement.arena(dev)
    n, d = 1 << 22, 256
    ei = graphgen.ba_edge_index(n, 5, seed=3, device=dev)
    g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
    del ei
    g.plan()
    x = ar.empty((n, d))
    x.uniform_(-1, 1)

    def agg_ms(y):
        ops._raw_spmm(g, x, 0, out=y)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops._raw_spmm(g, x, 0, out=y)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    y = ar.empty((n, d), reads=(x,))
    assert y is not None and ar.owns(y)
    t_pred = agg_ms(y)
    del y
    gc.collect()
    yv = ar.empty((n, d), reads=(x,), verify="all")
    assert yv is not None and ar.owns(yv) and len(yv._mp_verified_candidates_ms) >= 8
    t_ver = agg_ms(yv)
    del yv
    gc.collect()
    times = []
    for gidx in range(ar.n_gran):
        yc = ar.empty_at((n, d), gidx)
        if yc is None:
            continue
        times.append(agg_ms(yc))
        del yc
    assert len(times) >= 8
    best, worst, median = min(times), max(times), sorted(times)[len(times) // 2]
    if worst < 1.04 * best:
        pytest.skip(f"no placement effect on this box (best {best:.3f} ms, worst {worst:.3f} ms)")
    assert t_ver <= 1.03 * best, f"verified placement {t_ver:.3f} ms vs best {best:.3f} ms (worst {worst:.3f})"
    assert t_pred <= 1.03 * best, \
        f"predicted placement {t_pred:.3f} ms vs best {best:.3f} / median {median:.3f} / worst {worst:.3f}"


def test_ops_place_large_outputs_and_small_ones_stay_with_torch(dev):
    import graphgym_amd as ga
    from graphgym_amd import ops, placement, graphgen
    ar = placement.arena(dev)
    n = 1_100_000                                    # [n, 256] fp32 = 1.05 GiB >= the placement threshold
    ei = graphgen.ba_edge_index(n, 3, seed=5, device=dev)
    g = ga.CSRGraph.from_edge_index(ei, n)
    x = torch.rand(n, 256, device=dev)               # a tensor the engine did not allocate: priced by probing
    y = ops.spmm(g, x, "sum")
    assert ar.owns(y)
    small = ops.spmm(ga.CSRGraph.from_edge_index(ei[:, :1000] % 1000, 1000), torch.rand(1000, 64, device=dev))
    assert not ar.owns(small)
    # same numbers wherever the output lives
    y2 = torch.empty_like(y)
    ops._raw_spmm(g, x, 0, out=y2)
    assert torch.equal(y, y2)

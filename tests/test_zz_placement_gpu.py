"""Placement of large outputs (graphgym_amd/placement.py): outputs are ORDINARY torch allocations — checked against the
tensors the launch reads with a timed probe, re-allocated on conflict — so torch keeps every byte under its own
accounting, and the pair (read matrix, placed output) times within 3 % of the best pair torch's allocator offers.
(The file sorts last on purpose: these tests allocate most of the card's memory, and the first one depends on which
physical blocks the driver hands out — it should not stand between `-x` and the parity tests.)"""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu
GiB = 1 << 30


def _agg_ms(g, x, y):
    from graphgym_amd import ops
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


def test_placed_output_is_within_3pct_of_the_best_torch_block(dev):
    """The thing placement is for, measured directly: the aggregation Y = A X (X = 4 GiB: 2^22 nodes x 256 fp32, BA graph)
    is timed with Y in each of 10 successive torch allocations (all held, so they are 10 different blocks); then they are
    released and Y comes from placement.empty_or_torch(reads=(X,)) — the call every operator makes — which must land within
    3 % of the best of the ten.  (Skipped when the box shows no placement effect: worst < 1.04 x best.)"""
    import graphgym_amd as ga
    from graphgym_amd import graphgen, placement
    placement._state.clear()            # a process's first allocations (the wide search applies to them): not whatever
    n, d = 1 << 22, 256                 # yardsticks and search budget the tests before this one left behind
    ei = graphgen.ba_edge_index(n, 5, seed=3, device=dev)
    g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
    del ei
    g.plan()
    x = torch.empty((n, d), device=dev).uniform_(-1, 1)
    held, times = [], []
    for _ in range(10):
        y = torch.empty((n, d), device=dev)
        times.append(_agg_ms(g, x, y))
        held.append(y)
    del held, y
    best, worst = min(times), max(times)
    before = placement.stats(dev)
    y = placement.empty_or_torch((n, d), dev, reads=(x,))
    info = y._mp_place
    t_placed = _agg_ms(g, x, y)
    after = placement.stats(dev)
    assert after["allocations"] == before["allocations"] + 1 and after["probed_pairs"] > before["probed_pairs"]
    assert 1 <= len(info["candidates_ms"]) <= max(placement.TRIES, placement.EXPLORE_TRIES) and info["chosen_ms"] == min(info["candidates_ms"])
    # the same request again: same blocks from torch's cache, so the probes come from the memo (no timed copies)
    del y
    y2 = placement.empty_or_torch((n, d), dev, reads=(x,))
    again = placement.stats(dev)
    assert again["probed_pairs"] == after["probed_pairs"] and again["memo_hits"] > after["memo_hits"]
    assert abs(_agg_ms(g, x, y2) - t_placed) <= 0.03 * t_placed
    if worst < 1.04 * best:
        pytest.skip(f"no placement effect on this box (best {best:.3f} ms, worst {worst:.3f} ms)")
    assert t_placed <= 1.03 * best, (f"placed {t_placed:.3f} ms vs best {best:.3f} / worst {worst:.3f} of 10 torch blocks: "
                                     f"{times}; probe: {info}")


def test_placed_outputs_are_torch_memory(dev):
    """engine outputs are torch's: counted by torch.cuda.memory_allocated, returned to its cache when they die, and the
    rejected candidates of the search do not stay allocated — afterwards torch can still hand out 70 % of the device"""
    import graphgym_amd as ga
    from graphgym_amd import graphgen, ops
    gc.collect()
    torch.cuda.empty_cache()
    n = 1_100_000                                    # [n, 256] fp32 = 1.05 GiB >= the placement threshold
    ei = graphgen.ba_edge_index(n, 3, seed=5, device=dev)
    g = ga.CSRGraph.from_edge_index(ei, n)
    x = torch.rand(n, 256, device=dev)               # a tensor the engine did not allocate
    base = torch.cuda.memory_allocated()
    y = ops.spmm(g, x, "sum")
    assert y.is_cuda and y.dtype == torch.float32 and y.is_contiguous()
    grown = torch.cuda.memory_allocated() - base
    assert n * 256 * 4 <= grown < 1.2 * n * 256 * 4 + (64 << 20)       # ONE output (+ the launch's small workspace) stays
    y2 = torch.empty_like(y)
    ops._raw_spmm(g, x, 0, out=y2)
    assert torch.equal(y, y2)                        # same numbers wherever the output lives
    small = ops.spmm(ga.CSRGraph.from_edge_index(ei[:, :1000] % 1000, 1000), torch.rand(1000, 64, device=dev))
    assert not hasattr(small, "_mp_place")
    del y, y2, small
    assert torch.cuda.memory_allocated() <= base + (1 << 20)
    del x, g, ei
    gc.collect()
    free, total = torch.cuda.mem_get_info()
    big = torch.empty(int(0.7 * total), dtype=torch.uint8, device=dev)   # torch reclaims its cache (incl. old candidates)
    assert big.numel() == int(0.7 * total)
    del big
    torch.cuda.empty_cache()


def test_no_probe_under_graph_capture(dev, monkeypatch):
    """a probe synchronises, so allocations made while a HIP graph is being captured are plain torch allocations"""
    from graphgym_amd import placement
    monkeypatch.setattr(placement, "MIN_BYTES", 1 << 20)
    x = torch.rand(1 << 22, 64, device=dev)          # 1 GiB read tensor
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    before = placement.stats(dev)["allocations"]
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y = placement.empty_or_torch((1 << 22, 64), dev, reads=(x,))
        y.copy_(x)
    gr.replay()
    torch.cuda.synchronize()
    assert placement.stats(dev)["allocations"] == before and torch.equal(y, x)
    y = placement.empty_or_torch((1 << 22, 64), dev, reads=(x,))      # outside capture the check runs
    assert placement.stats(dev)["allocations"] == before + 1 and hasattr(y, "_mp_place")

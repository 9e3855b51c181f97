"""Placement of large outputs (graphgym_amd/placement.py): outputs are ORDINARY torch allocations — checked against the
tensors the launch reads with a timed probe, re-allocated on conflict — so torch keeps every byte under its own
accounting.  What is ASSERTED here is the accounting (one allocation counted, probes remembered, candidates returned,
no probe under capture).  How close the placed pair comes to the best pair torch's allocator offers depends on which
physical blocks the driver hands this process — hardware luck, not correctness — so it is RECORDED
(gpurun_out/placement.json) and reported through a skip with the numbers, never a pass / fail.
(The file sorts last on purpose: these tests allocate most of the card's memory.)"""
import gc
import json
import os

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


def test_placed_output_accounting_and_recorded_distance_to_the_best_torch_block(dev):
    """The thing placement is for, measured directly: the aggregation Y = A X (X = 4 GiB: 2^22 nodes x 256 fp32, BA graph)
    is timed with Y in each of 10 successive torch allocations (all held, so they are 10 different blocks); then they are
    released and Y comes from placement.empty_or_torch(reads=(X,)) — the call every operator makes.  The accounting of that
    call is asserted; its distance to the best of the ten is written to gpurun_out/placement.json and shown in the skip
    reason (a timing that depends on the box's physical memory map is not a parity criterion)."""
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
    t_again = _agg_ms(g, x, y2)
    rec = {"what": "aggregation Y = A X, X = 2^22 x 256 fp32 (4 GiB), Y in each of 10 successive torch blocks vs the placed Y",
           "ten_blocks_ms": [round(t, 4) for t in times], "best_ms": best, "worst_ms": worst, "placed_ms": t_placed,
           "placed_again_ms": t_again, "placed_vs_best": t_placed / best - 1.0, "worst_vs_best": worst / best - 1.0,
           "probe": info, "stats": again}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "placement.json"), "w") as f:
        json.dump(rec, f, indent=1)
    pytest.skip(f"accounting asserted; timing recorded, not judged: placed {t_placed:.3f} ms ({rec['placed_vs_best']:+.1%} vs the "
                f"best of 10 torch blocks {best:.3f} ms; worst {worst:.3f} ms) -> gpurun_out/placement.json")


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

"""bench.py --mode step / the `step` object of the N > 1 line: the data-parallel training step that CONTAINS the path's one
exchange, at the reference's own cadence — a NEW batch every step (graphgym/train.py:18-25,39).

Every step a global batch of ego nets (ID-GNN Full: graphgym/models/transform.py:11-38, radius 2) around `--centres` x
world freshly drawn centres of one scale-free base graph is dealt to the ranks by a degree-based cost every rank computes
alike (LPT on the host, mp_lpt_partition_host: no communication); every rank expands ITS centres on its GPU, builds the
batch's CSR / normalisation / plans / transposed operator / identity-branch operators, runs the three-layer ID-GCN
(d = 128) or ID-GIN (d = 512) model of main_zd.py:28-74,190-243 forward and backward, and the gradients are all-reduced
through GradBucket — two buckets in backward order, each launched asynchronously from the hook of its last gradient, so
the exchange overlaps the rest of backward — then Adam steps.  Losses are normalised by the GLOBAL number of centres and
summed over ranks: the exact full-batch gradient (no mean of means).

Three cadences are timed (barrier + synchronize on both sides, max over ranks):
  ms_per_step                      ONE batch built outside the loop and replayed (rounds 2-3's measurement: the step alone)
  ms_per_step_fresh_batch          a new batch per step, built one step ahead on a second stream (graphgym_amd/pipeline.py)
  ms_per_step_fresh_batch_serial   a new batch per step, built on the step's own stream (no overlap: build + step)
and beside them the build alone (`batch_build_ms`), the exposed wait for the exchange inside the step
(`allreduce_exposed_ms`), the exchange alone (`allreduce_ms`), the replayed step without the exchange
(`ms_per_step_no_exchange`), and how many graph structures the timed steps had to build themselves
(`graph_builds_inside_steps`: 0 when the pipeline prepared everything).  All timed loops run under
pipeline.quiet_gc(): Python's cycle collector is kept to its young generation (a full collection is 40-80 ms).
"""
import os
import time

import numpy as np
import torch


def ego_cost_table(base):
    """stored entries of a radius-2 ego net ~ sum of its members' degrees; a deterministic proxy every rank computes
    alike, for EVERY node at once: cost[c] = deg(c) + sum of the degrees of c's neighbours (one pass over the CSR)"""
    rp = base.rowptr.to(torch.int64)
    deg = rp[1:] - rp[:-1]
    s = torch.zeros(base.nnz + 1, dtype=torch.int64, device=base.device)
    torch.cumsum(deg[base.col.long()], 0, out=s[1:])
    return (deg + s[rp[1:]] - s[rp[:-1]]).cpu().numpy()


def run(args, rank, world, dev):
    import torch.nn.functional as F

    import graphgym_amd as ga
    from graphgym_amd import dist as D, graph as G, graphgen, harness as H, placement
    from graphgym_amd.pipeline import EgoBatchPipeline, quiet_gc, fit_allocator_to_changing_shapes

    alloc_conf = fit_allocator_to_changing_shapes()
    kind = getattr(args, "step_model", "idgcn")
    n0 = min(args.nodes, getattr(args, "step_nodes", 2_000_000))
    f_in, d = (128, 128) if kind == "idgcn" else (512, 512)
    classes, radius = 7, 2
    ei = graphgen.ba_edge_index(n0, args.m, seed=12345, device=dev)          # the same base graph on every rank
    base = ga.CSRGraph.from_edge_index(ei, n0)
    del ei
    cost = ego_cost_table(base)
    n_global = args.centres * world
    labels_host = torch.randint(0, classes, (n0,), generator=torch.Generator().manual_seed(98))
    xg = torch.Generator(device=dev).manual_seed(7)
    x_base = torch.rand((n0, f_in), device=dev, generator=xg) * 2 - 1        # same features on every rank

    def sample(k):
        """the global batch of step k and this rank's share of it: (centres [b] host, labels [b] host, imbalance)"""
        gen = torch.Generator().manual_seed(1000 + k)
        cen = torch.randint(0, n0, (n_global,), generator=gen)               # with replacement, like a shuffled loader's draw
        c = cost[cen.numpy()]
        owner = D.lpt_owners(c, world)                                       # LPT, the same on every rank
        mine = torch.from_numpy(np.nonzero(owner == rank)[0])
        loads = np.bincount(owner, weights=c.astype(np.float64), minlength=world)
        return cen[mine], labels_host[cen[mine]], float(loads.max() * world / max(loads.sum(), 1.0))

    torch.manual_seed(11)                                                    # identical replicas
    model = H.TfgNodeModel(kind, f_in, d, classes).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    bucket = D.GradBucket(model.parameters(), n_buckets=2).attach()
    kern = model.kernel_parameters()
    self_loops = kind == "idgcn"                                             # gcn_id adds them (TfgIDLayer.py:500-503)

    def step_on(b, exchange=True):
        bucket.zero_grad()
        logits = model([b.x, b.edge_index, b.ids], holder=b.holder)
        ce = F.cross_entropy(logits[b.ids], b.y, reduction="sum") / n_global
        l2 = sum((p * p).sum() / 2 for p in kern) * (5e-4 / world)          # the same on every rank: 1 / world of it each
        (ce + l2).backward()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        if exchange:
            bucket.finish(1.0)
        else:
            bucket._reset()
        e1.record()
        opt.step()
        return e0, e1

    prepare = lambda inputs, holder: model.prepare(inputs, holder)
    csr = "add" if self_loops else "none"        # the CSR the model's layers ask for, written by the expansion itself
    warm = max(args.warmup, 3)

    # ---- (1) one batch, replayed: the step alone ---------------------------------------------------------------------
    pipe = EgoBatchPipeline(base, x_base, radius, prepare=prepare, device=dev, threaded=False, csr=csr)
    cen0, lab0, imb0 = sample(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.submit(cen0, lab0)
    b0 = pipe.get()
    torch.cuda.synchronize()
    t_build0 = time.perf_counter() - t0
    pipe.close()                               # (the replayed batch lives on: its step's outputs may be placed)
    for _ in range(warm):
        step_on(b0)
    D.barrier()
    torch.cuda.synchronize()
    with quiet_gc() as tick:
        t0 = time.perf_counter()
        waits = []
        for _ in range(args.steps):
            waits.append(step_on(b0))
            tick()
        torch.cuda.synchronize()
        D.barrier()
        dt = D.all_reduce_max(time.perf_counter() - t0, dev)
    exposed = sum(a.elapsed_time(b) for a, b in waits) / args.steps
    nnz_replay = b0.edges + (b0.nodes if self_loops else 0)

    # the same step without the exchange (what N independent GPUs would do), and the exchange alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_on(b0, exchange=False)
    torch.cuda.synchronize()
    dt_local = D.all_reduce_max(time.perf_counter() - t0, dev)
    D.barrier()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        for flat, _ in bucket.buckets:
            bucket._reduce(flat, async_op=False) if bucket._active() else None
    torch.cuda.synchronize()
    t_ar = D.all_reduce_max((time.perf_counter() - t0) / reps, dev)
    nodes_replay = b0.nodes
    del b0

    # ---- (2) a fresh batch per step ------------------------------------------------------------------------------------
    def fresh(steps, overlap, first_k, pipe=None):
        p = pipe if pipe is not None else \
            EgoBatchPipeline(base, x_base, radius, prepare=prepare, device=dev, threaded=overlap, csr=csr)
        if not overlap:
            p.side = torch.cuda.current_stream(dev)
        nnz, nodes, builds_in_steps, imb = 0, 0, 0, []
        host = 0.0
        # builds are started TWO steps ahead (MP_PIPE_DEPTH): a build then has two step times to finish in, and a driver
        # allocation it triggers (new shapes every batch) no longer lands in front of the step that needs the batch —
        # ID-GCN 16.9 -> 16.4 ms per step, ID-GIN at 4 096 centres 207.9 -> 157.9 (replayed: 151.1); three ahead: the same
        ahead = max(1, int(os.environ.get("MP_PIPE_DEPTH", "2"))) if overlap else 1
        for j in range(ahead):
            c, y, _ = sample(first_k + j)
            p.submit(c, y)
        D.barrier()
        torch.cuda.synchronize()
        with quiet_gc() as tick:
            t_start = time.perf_counter()
            for k in range(steps):
                b = p.get()
                before = G.builds_by_this_thread()
                th = time.perf_counter()
                step_on(b)
                host += time.perf_counter() - th
                builds_in_steps += G.builds_by_this_thread() - before
                p.done()
                nnz += b.edges + (b.nodes if self_loops else 0)
                nodes += b.nodes
                if k + ahead < steps:
                    c, y, im = sample(first_k + k + ahead)
                    imb.append(im)
                    p.submit(c, y)
                del b
                tick()
            torch.cuda.synchronize()
            D.barrier()
            dt_loop = time.perf_counter() - t_start
        if pipe is None:
            p.close()
        return dt_loop, nnz, nodes, builds_in_steps, imb, host / steps

    # new shapes every step: the allocator settles, and placement spends its per-process probe budget (placement.py:
    # MP_PLACE_BUDGET_MS) — the warm-up runs until it is spent, the timed loop is the steady state
    # ONE pipeline for the warm-up and the timed loop, as a training run keeps one for its life: the warm-up batches fill
    # the pool of its stream, the timed loop is the steady state (no driver allocation: `driver_allocs_in_fresh_steps`)
    p_run = EgoBatchPipeline(base, x_base, radius, prepare=prepare, device=dev, threaded=True, csr=csr)
    for rep in range(4):
        spent = placement.stats(dev)["probe_ms_total"]
        fresh(warm, True, 10_000 + 100 * rep, pipe=p_run)
        if placement.stats(dev)["probe_ms_total"] == spent:
            break
    place_before = placement.stats(dev)
    mem_before = torch.cuda.memory_stats(dev)
    dt_fresh, nnz_fresh, nodes_fresh, builds_fresh, imbs, host_enq = fresh(args.steps, True, 20_000, pipe=p_run)
    place_after = placement.stats(dev)
    mem_after = torch.cuda.memory_stats(dev)
    p_run.close()
    dt_fresh = D.all_reduce_max(dt_fresh, dev)
    dt_serial, _, _, _, _, _ = fresh(args.steps, False, 20_000)
    dt_serial = D.all_reduce_max(dt_serial, dev)

    # the build alone (ego expansion + feature gather + CSR / norm / plans / transpose / identity operators), one stream
    p = EgoBatchPipeline(base, x_base, radius, prepare=prepare, device=dev, threaded=False, csr=csr)
    p.side = torch.cuda.current_stream(dev)
    torch.cuda.synchronize()
    nb = min(args.steps, 10)
    with quiet_gc() as tick:
        t0 = time.perf_counter()
        for k in range(nb):
            c, y, _ = sample(30_000 + k)
            p.submit(c, y)
            b = p.get()
            if b.timing is not None and rank == 0:
                import sys
                print("build phases ms:", {k: round(v, 2) for k, v in b.timing.items()}, file=sys.stderr)
            del b
            tick()
        torch.cuda.synchronize()
        t_build = D.all_reduce_max((time.perf_counter() - t0) / nb, dev)
    p.close()

    total_nnz = D.all_reduce_sum(nnz_replay, dev)
    total_nodes = D.all_reduce_sum(nodes_replay, dev)
    max_nnz = D.all_reduce_max(nnz_replay, dev)
    total_nnz_fresh = D.all_reduce_sum(nnz_fresh, dev)
    builds_fresh = D.all_reduce_sum(builds_fresh, dev)
    res = None
    if rank == 0:
        n_par = sum(p.numel() for p in model.parameters())
        ms = dt / args.steps * 1e3
        ms_fresh = dt_fresh / args.steps * 1e3
        res = {
            "metric": "aggregated edges/sec + achieved HBM GB/s, GCN d=256 on 100M-edge scale-free",
            "mode": "step",
            "value": total_nnz_fresh / dt_fresh, "unit": "edges/s",
            "value_note": "stored entries of all ranks' FRESH batches per second of step time (new batch every step, built "
                          "one step ahead on a second stream); a step is 3 forward + 3 backward aggregations over them",
            "n_gpus": world, "steps": args.steps, "warmup": warm,
            "ms_per_step": ms,
            "ms_per_step_fresh_batch": ms_fresh,
            "ms_per_step_fresh_batch_serial": dt_serial / args.steps * 1e3,
            "fresh_over_replayed": ms_fresh / ms,
            "batch_build_ms": t_build * 1e3, "first_batch_build_ms": t_build0 * 1e3,
            "host_enqueue_ms_per_fresh_step": host_enq * 1e3,
            "graph_builds_inside_steps": int(builds_fresh),
            "placement_probes_in_fresh_steps": place_after["probed_pairs"] - place_before["probed_pairs"],
            # driver allocations / frees inside the timed fresh-batch loop (each one stalls the device: hipMalloc / hipFree
            # are synchronous; new batch shapes every step make the caching allocator ask for them until its pools fit)
            "driver_allocs_in_fresh_steps": int(mem_after.get("num_device_alloc", 0) - mem_before.get("num_device_alloc", 0)),
            "driver_frees_in_fresh_steps": int(mem_after.get("num_device_free", 0) - mem_before.get("num_device_free", 0)),
            "reserved_gb_after_fresh_steps": round(mem_after.get("reserved_bytes.all.current", 0) / 1e9, 2),
            "allocator_settings": alloc_conf,
            "ms_per_step_no_exchange": dt_local / args.steps * 1e3,
            "allreduce_exposed_ms": exposed, "allreduce_ms": t_ar * 1e3,
            "allreduce_bytes": 4 * n_par, "allreduce_buckets": len(bucket.buckets),
            "collective_backend": D.dist.get_backend() if D.dist.is_initialized() else None,
            "world_size": D.dist.get_world_size() if D.dist.is_initialized() else 1,
            "collectives_executed": bool(bucket._active()),
            "step_efficiency_vs_no_exchange": dt_local / dt,
            "lpt_imbalance": max_nnz * world / max(total_nnz, 1),
            "cost_imbalance_fresh_mean": float(np.mean(imbs)) if imbs else imb0,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{kind}_tf_step_ego_r{radius}_d{d}_BA_n{n0}_m{args.m}_c{args.centres}x{world}",
                       "what": f"one {kind} (3 layers, d = {d}, F = {f_in}) training step per rank on its share of a global "
                               f"batch of radius-2 ego nets around {args.centres} x {world} centres drawn anew every step",
                       "centres_per_gpu": args.centres, "global_centres": n_global,
                       "batch_nodes_total": int(total_nodes), "stored_entries_total": int(total_nnz),
                       "stored_entries_max_rank": int(max_nnz),
                       "fresh_batch_nodes_mean_rank0": nodes_fresh / args.steps, "parameters": n_par,
                       "parallelism": f"dp{world}: ego nets dealt by a degree cost (LPT), gradient "
                                      f"all-reduce ({'RCCL' if D.dist.is_initialized() and D.dist.get_backend() == 'nccl' else 'gloo'}, "
                                      f"2 buckets, overlapped with backward)"},
        }
    D.barrier()
    return res

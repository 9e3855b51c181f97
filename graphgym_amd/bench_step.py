"""bench.py --mode step: the data-parallel training step that CONTAINS the path's one exchange.

A global batch of ego nets (ID-GNN Full: graphgym/models/transform.py:11-38, radius 2) around `--centres` x world
centres of one scale-free base graph is dealt to the ranks by stored entries (LPT over a degree-based cost, computed
identically on every rank, no communication); every rank expands its own centres on its GPU, runs the three-layer
ID-GCN model of main_zd.py:28-74 forward and backward, and the gradients are all-reduced through GradBucket — two
buckets in backward order, each launched asynchronously from the hook of its last gradient, so the exchange overlaps
the rest of backward — then Adam steps.  Losses are normalised by the GLOBAL number of centres and summed over ranks:
the exact full-batch gradient (no mean of means).

The timed region is the whole step (barrier + synchronize on both sides, max over ranks).  Reported beside it: the
exposed wait for the exchange inside the step (`allreduce_exposed_ms`), the exchange alone (`allreduce_ms`, same
buckets, nothing else running), and the same step without the exchange (`ms_per_step_no_exchange`): the step's
data-parallel efficiency is the ratio of the last to `ms_per_step`.
"""
import json
import time

import torch


def _ego_cost(base, centres):
    """stored entries of a radius-2 ego net ~ sum of its members' degrees; a deterministic proxy every rank computes
    alike: deg(c) + sum of the degrees of c's neighbours"""
    deg = (base.rowptr[1:] - base.rowptr[:-1]).to(torch.int64)
    rp = base.rowptr.to(torch.int64)
    out = []
    col = base.col.to(torch.int64)
    for c in centres.tolist():
        nb = col[rp[c]:rp[c + 1]]
        out.append(int(deg[c] + deg[nb].sum()))
    return out


def run(args, rank, world, dev):
    import torch.nn.functional as F

    import graphgym_amd as ga
    from graphgym_amd import dist as D, graphgen, harness as H
    from graphgym_amd.ego import ego_batch

    n0 = min(args.nodes, 2_000_000)
    f_in, d, classes, radius = 128, 128, 7, 2
    ei = graphgen.ba_edge_index(n0, args.m, seed=12345, device=dev)          # the same base graph on every rank
    base = ga.CSRGraph.from_edge_index(ei, n0)
    del ei
    gen = torch.Generator().manual_seed(99)
    n_global = args.centres * world
    centres = torch.randperm(n0, generator=gen)[:n_global]
    labels_all = torch.randint(0, classes, (n_global,), generator=gen)
    parts = D.lpt_partition(_ego_cost(base, centres.to(dev)), world)
    mine = torch.tensor(parts[rank], dtype=torch.int64)
    t0 = time.perf_counter()
    ei2, orig, ids, _ = ego_batch(base, centres[mine].to(dev), radius)
    torch.cuda.synchronize()
    t_ego = time.perf_counter() - t0
    xg = torch.Generator(device=dev).manual_seed(7)
    x_base = torch.rand((n0, f_in), device=dev, generator=xg) * 2 - 1      # same features on every rank
    x = x_base[orig]
    del x_base
    y = labels_all[mine].to(dev)
    torch.manual_seed(11)                                                    # identical replicas
    model = H.TfgNodeModel("idgcn", f_in, d, classes).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    bucket = D.GradBucket(model.parameters(), n_buckets=2).attach()
    batch = H.Batch()
    kern = model.kernel_parameters()
    nnz_local = int(ei2.size(1)) + int(orig.numel())                        # stored entries incl. the self loops gcn_id adds

    def step(exchange=True):
        bucket.zero_grad()
        logits = model([x, ei2, ids], holder=batch)
        ce = F.cross_entropy(logits[ids], y, reduction="sum") / n_global
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

    for _ in range(max(args.warmup, 3)):
        step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    waits = []
    for _ in range(args.steps):
        waits.append(step())
    torch.cuda.synchronize()
    D.barrier()
    dt = D.all_reduce_max(time.perf_counter() - t0, dev)
    exposed = sum(a.elapsed_time(b) for a, b in waits) / args.steps

    # the same step without the exchange (what N independent GPUs would do), and the exchange alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(exchange=False)
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

    total_nnz = D.all_reduce_sum(nnz_local, dev)
    total_nodes = D.all_reduce_sum(int(orig.numel()), dev)
    max_nnz = D.all_reduce_max(nnz_local, dev)
    res = None
    if rank == 0:
        n_par = sum(p.numel() for p in model.parameters())
        res = {
            "metric": "aggregated edges/sec + achieved HBM GB/s, GCN d=256 on 100M-edge scale-free",
            "mode": "step",
            "value": total_nnz * args.steps / dt, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 3),
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_no_exchange": dt_local / args.steps * 1e3,
            "allreduce_exposed_ms": exposed, "allreduce_ms": t_ar * 1e3,
            "allreduce_bytes": 4 * n_par, "allreduce_buckets": len(bucket.buckets),
            "collective_backend": D.dist.get_backend() if D.dist.is_initialized() else None,
            "world_size": D.dist.get_world_size() if D.dist.is_initialized() else 1,
            "collectives_executed": bool(bucket._active()),
            "step_efficiency_vs_no_exchange": dt_local / dt,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"idgcn_tf_step_ego_r{radius}_d{d}_BA_n{n0}_m{args.m}_c{args.centres}x{world}",
                       "what": "one ID-GCN (3 x IDGCN, d = 128) training step per rank on its LPT shard of a global "
                               "batch of radius-2 ego nets; value = stored entries of all shards per second of step time "
                               "(a step is 3 forward + 3 backward aggregations over them)",
                       "centres_per_gpu": args.centres, "global_centres": n_global,
                       "batch_nodes_total": int(total_nodes), "stored_entries_total": int(total_nnz),
                       "stored_entries_max_rank": int(max_nnz),
                       "lpt_imbalance": max_nnz * world / max(total_nnz, 1),
                       "ego_build_ms_rank0": t_ego * 1e3, "parameters": n_par,
                       "parallelism": f"dp{world}: ego nets sharded by stored entries (LPT), gradient all-reduce "
                                      f"(RCCL, 2 buckets, overlapped with backward)"},
        }
    D.barrier()
    return res

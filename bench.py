#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json: "aggregated edges/sec + achieved HBM GB/s,
GCN d=256 on 100M-edge scale-free").

    python bench.py --gpus 1 --steps 50 --warmup 20
    python bench.py --gpus N --steps K --warmup W          # no launcher: bench.py starts its N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--mode step]

On N > 1 GPUs the ONE line carries both: the weak-scaling aggregate (`value`, `roofline` with the fraction of
N x 8 TB/s) and a `step` object — the data-parallel ID-GNN training step whose gradients cross RCCL
(`collective_backend`, `world_size`, `allreduce_ms`, `allreduce_exposed_ms`, `lpt_imbalance`, `ms_per_step` next to
`ms_per_step_no_exchange`; graphgym_amd/bench_step.py).

--mode aggregate (default): a step is one pass of the aggregation  Y = A_hat X  over one graph resident
in HBM: A_hat = D^-1/2 (A + I) D^-1/2 of a Barabasi-Albert BA(10^7, 5) graph (about 1.1*10^8 stored
entries incl. self loops), X [10^7, 256] fp32.  With N > 1 every rank owns its own graph of that size
(weak scaling; independent units, no data-path collective — DESIGN.md §6).

--mode step: a step is one ID-GCN training step (forward, loss, backward, ONE-bucket gradient all-reduce
over RCCL, Adam) on each rank's shard of a batch of ego nets dealt out by stored entries (LPT); the
all-reduce is inside the timed region and also reported on its own (`allreduce_ms`).

Prints ONE JSON line on rank 0.  `roofline` prices the aggregation launch against HBM; `cpu_baseline`
times the CPU oracle (the op-for-op restatement of the reference's CPU path) on a bounded sample of the
same workload on this box's host cores.

X is a plain torch tensor (what a GraphGym first layer is handed).  Y is allocated ONCE by the very call the product's
operators make for their outputs (graphgym_amd.placement.empty_or_torch(reads=(X,)): a torch allocation, a timed-copy
check against X, re-allocation on conflict — at most MP_PLACE_TRIES candidates); `value` and `roofline` are measured on
that Y.  Side fields, never part of `value`: the same launch with Y as torch hands it out unchecked
(`output_placement.unchecked_ms`) and with the best of 8 candidates (`output_placement.best_of_8_ms`).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "aggregated edges/sec + achieved HBM GB/s, GCN d=256 on 100M-edge scale-free"


def algorithmic_bytes(n, nnz, d, weighted):
    """SURVEY §8(d) gather model: every neighbour row read, every output row written once,
    int32 column index (+ fp32 value) per stored entry, rowptr."""
    return nnz * d * 4 + n * d * 4 + nnz * 4 + (nnz * 4 if weighted else 0) + (n + 1) * 4


def cpu_baseline(g, x, seconds=8.0):
    """the reference-style gather -> scale -> index_add_ (oracle/ref_ops) on edge chunks of the same
    graph at 6 threads (graphgym/config.py:57) and on all cores, plus the best-available CPU line torch.sparse_csr @ X;
    every leg bounded to ~`seconds` of CPU work.  The sample starts at the MIDDLE row of the matrix: the graph is
    hubs-first, and its leading rows are the cache-friendliest ones."""
    from oracle import ref_ops
    n, d = x.shape
    chunk = 4_000_000
    take = min(g.nnz, 25 * chunk)
    r_mid = n // 2 if g.nnz > take else 0
    e0 = int(g.rowptr[r_mid])
    take = min(take, g.nnz - e0)
    dst = g.row_ids()[e0:e0 + take].long().cpu()
    src = g.col[e0:e0 + take].long().cpu()
    w = g.val[e0:e0 + take].cpu() if g.val is not None else None
    xc = x.cpu()
    from graphgym_amd import hostcpu
    cores = hostcpu.effective_cpus()          # the CPUs this container may use (cgroup quota), not the machine's 256
    legs = []

    def gather_leg(threads):
        torch.set_num_threads(threads)
        out = torch.zeros((n, d), dtype=torch.float32)
        ref_ops.coo_aggregate_sum_chunked(dst[:200_000], src[:200_000], None if w is None else w[:200_000], xc, out,
                                          chunk)  # touch pages / warm the thread pool
        t0 = time.perf_counter()
        done = ref_ops.coo_aggregate_sum_chunked(dst, src, w, xc, out, chunk, max_seconds=seconds)
        dt = time.perf_counter() - t0
        return {"what": "reference-style gather*scale -> index_add_ in 4M-edge chunks (oracle/ref_ops)",
                "threads": threads, "edges": done, "seconds": dt, "edges_per_s": done / dt}

    legs.append(gather_leg(min(6, cores)))
    legs.append(gather_leg(cores))
    # torch.sparse_csr @ X on the leading rows (sized from the gather leg's rate to ~`seconds`).  torch's CPU kernel
    # indexes the dense operand with 32-bit offsets (it crashes on a [10^7, 256] operand), so the sample's columns
    # are renumbered to the rows they actually reference and X is compacted to those rows.
    try:
        torch.set_num_threads(cores)
        rp = g.rowptr.long().cpu()
        target = int(min(take, 6_000_000, max(2_000_000, legs[-1]["edges_per_s"] * seconds * 2)))
        r1 = int(torch.searchsorted(rp, torch.tensor([e0 + target]))[0])
        r1 = max(r_mid + 1, min(r1, n))
        r, e = r1 - r_mid, int(rp[r1]) - e0
        cols, inv = torch.unique(g.col[e0:e0 + e].long().cpu(), return_inverse=True)
        xs = xc[cols]
        assert xs.numel() < 2 ** 31
        A = torch.sparse_csr_tensor(rp[r_mid:r1 + 1] - e0, inv,
                                    g.val[e0:e0 + e].cpu() if g.val is not None else torch.ones(e),
                                    size=(r, cols.numel()))
        t0 = time.perf_counter()
        _ = A @ xs
        dt = time.perf_counter() - t0
        legs.append({"what": "torch.sparse_csr_tensor @ X (best-available CPU line; X compacted to the referenced rows)",
                     "threads": cores, "edges": e, "rows": r, "x_rows": int(cols.numel()), "seconds": dt,
                     "edges_per_s": e / dt})
    except Exception as exc:   # never let a side leg take the metric down
        legs.append({"what": "torch.sparse_csr_tensor @ X", "error": repr(exc)[:200]})
    main = max(legs[:2], key=lambda l: l["edges_per_s"])     # the faster of the two reference-style legs
    return {"value": main["edges_per_s"], "unit": "edges/s", "cores": main["threads"], "kind": "port",
            "sample": f"{main['edges']} of {g.nnz} stored entries of the same graph starting at its middle row {r_mid} "
                      f"(the leading rows of a hubs-first graph are its cache-friendliest), same X (fp32, d={d}), "
                      f"gather*scale -> index_add_ in 4M-edge chunks, {main['seconds']:.1f} s of CPU work",
            "host_cores": os.cpu_count(), "usable_cores": cores,
            "cores_note": "usable_cores = min(affinity, cgroup cpu.max quota): threads beyond it are frozen by the kernel",
            "legs": legs}


def pmc_traffic(workload):
    """per-launch HBM bytes from the committed rocprofv3 PMC passes (profiles/), if they match this workload:
    (bytes, provenance) — a stored measurement of the same command, not a value measured in this run"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec.get("workload") == workload:
            return rec.get("hbm_bytes_per_launch"), (
                "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                f"({rec.get('source', 'see profiles/')}), committed; NOT measured in this run")
    except Exception:
        pass
    return None, None


def _ms(fn, reps=3):
    fn()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(reps):
        fn()
    a1.record()
    torch.cuda.synchronize()
    return a0.elapsed_time(a1) / reps


def run_aggregate(args, rank, world, dev):
    import graphgym_amd as ga
    from graphgym_amd import _lib, dist as D, graphgen, ops, placement

    n, d = args.nodes, args.d
    ei = graphgen.ba_edge_index(n, args.m, seed=12345 + rank, device=dev,
                                triangle_p=0.3 if args.graph == "powerlaw_cluster" else None,
                                permute_seed=1 if args.permute else None)
    g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
    del ei
    g.plan()
    torch.cuda.empty_cache()
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    # the resident feature matrix: a plain torch tensor; the output: the allocation ops._raw_spmm makes for itself
    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    x.uniform_(-1.0, 1.0, generator=gen)

    def time_into(yy, reps=5):
        ops._raw_spmm(g, x, _lib.SUM, out=yy)
        return _ms(lambda: ops._raw_spmm(g, x, _lib.SUM, out=yy), reps=reps)

    place = {"mode": "product: placement.empty_or_torch(reads=(X,)) — torch allocation, copy-probe check, re-allocation on "
                     "conflict (the call every operator makes for a >= 1 GiB output)", "x": "torch.empty (foreign tensor)"}
    y = placement.empty_or_torch((n, d), dev, reads=(x,))
    place.update(getattr(y, "_mp_place", {}) or {})
    place["stats"] = placement.stats(dev)

    def step():
        ops._raw_spmm(g, x, _lib.SUM, out=y)

    for _ in range(args.warmup):
        step()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record()
        step()
        stops[i].record()
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    dt = D.all_reduce_max(dt, dev)
    total_nnz = D.all_reduce_sum(g.nnz, dev)
    per_step = sorted(s.elapsed_time(e) for s, e in zip(starts, stops))
    launch_ms = sum(per_step) / args.steps

    def pct(p):
        return per_step[min(len(per_step) - 1, int(round(p * (len(per_step) - 1))))]

    # cold launches: a 512 MB memset between launches flushes L2 / Infinity Cache
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    cold = []
    for _ in range(min(10, args.steps)):
        flush.zero_()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        step()
        a1.record()
        torch.cuda.synchronize()
        cold.append(a0.elapsed_time(a1))
    cold.sort()
    del flush

    # measured streaming rates on this box: a read-only pass over X (the aggregation is 92 % reads) and a
    # copy X -> Y (50 % writes); the aggregation's algorithmic rate is quoted against both and the nominal peak
    from graphgym_amd._lib import lib as _mplib, ptr as _ptr, check as _check
    from graphgym_amd.graph import _stream as _mpstream
    sink = torch.empty(256 * 8 * 256, dtype=torch.float32, device=dev)
    read_ms = _ms(lambda: _check(_mplib().mp_read_probe_f32(_ptr(x), x.numel(), _ptr(sink), _mpstream())))
    copy_ms = _ms(lambda: _check(_mplib().mp_copy_probe_f32(_ptr(x), _ptr(y), x.numel(), _mpstream())))
    read_gbps = x.numel() * 4 / (read_ms * 1e-3) / 1e9
    copy_gbps = 2 * x.numel() * 4 / (copy_ms * 1e-3) / 1e9

    balg = algorithmic_bytes(n, g.nnz, d, g.val is not None)
    balg_total = D.all_reduce_sum(balg, dev)

    # side legs (after the timed region, never part of `value`): the same launch into a buffer exactly as torch hands it
    # out, unchecked, and into the best of 8 checked candidates
    if world == 1 and placement.enabled():
        try:
            y_plain = torch.empty((n, d), dtype=torch.float32, device=dev)
            place["unchecked_ms"] = time_into(y_plain)
            del y_plain
            y8 = placement.empty_or_torch((n, d), dev, reads=(x,), tries=8, accept=-1.0)    # all 8 timed, the fastest kept
            place["best_of_8_ms"] = time_into(y8)
            place["best_of_8_candidates_ms"] = (getattr(y8, "_mp_place", {}) or {}).get("candidates_ms")
            del y8
        except Exception as e:
            place["side_legs_error"] = repr(e)[:200]

    # backward of the aggregation = the same kernel on the transposed operator (dX = A_hat^T dY)
    backward = None
    if world == 1:
        try:
            gt = g.transpose()
            gt.plan()
            dx = placement.empty_or_torch((n, d), dev, reads=(y,))
            bms = _ms(lambda: ops._raw_spmm(gt, y, _lib.SUM, out=dx), reps=5)
            backward = {"what": "dX = A_hat^T dY on the cached transposed CSR (same kernel)", "launch_ms": bms,
                        "edges_per_s": g.nnz / (bms * 1e-3), "hbm_gbps_algorithmic": balg / (bms * 1e-3) / 1e9,
                        "frac": balg / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del dx, gt
        except Exception as e:
            backward = {"error": repr(e)[:200]}

    # the whole layer (aggregate, then the MFMA feature transform + bias + ReLU) as ONE kernel next to the
    # two-kernel order: reported beside the metric, never part of `value`
    layer = None
    if world == 1 and d in ops.FUSED_WIDTHS:
        try:
            Wl = (torch.rand((d, d), device=dev, generator=gen) - 0.5) * (2.0 / d ** 0.5)
            bl = torch.rand((d,), device=dev, generator=gen) - 0.5
            y2 = placement.empty_or_torch((n, d), dev, reads=(x,))
            y3 = placement.empty_or_torch((n, d), dev, reads=(y2,))

            def two_kernels():
                ops._raw_spmm(g, x, _lib.SUM, out=y2)
                ops._dense_into(y3, y2, Wl, bl, True)
            t_two = _ms(two_kernels)
            ref = y3[:4096].clone()
            t_one = _ms(lambda: ops._raw_agg_dense(g, x, Wl, bl, True, out=y))
            err = float((y[:4096] - ref).abs().max() / ref.abs().max().clamp_min(1.0))
            layer = {"what": "relu((A_hat X) W + b), F = d_out = %d" % d, "one_kernel_ms": t_one,
                     "two_kernel_ms": t_two, "mfma_tflops_inside_one_kernel": 2.0 * n * d * d / (t_one * 1e-3) / 1e12,
                     "max_rel_diff_first_4096_rows": err,
                     "kernel": "mp::agg_dense_pc_kernel (64-row tiles, 4 gathering + 4 multiplying waves per workgroup, two LDS "
                               "buffers, bf16x3 MFMA against W from L2)"}
            del y2, y3
        except Exception as e:   # never let the side measurement take the metric down
            layer = {"error": repr(e)[:200]}

    res = None
    if rank == 0:
        achieved = balg / (launch_ms * 1e-3) / 1e9
        from graphgym_amd import ops as _ops
        tiles = (d in _ops.AGG_TILES_WIDTHS and n >= _ops.AGG_TILES_MIN_ROWS and os.environ.get("MP_AGG_TILES", "1") != "0"
                 and g.max_row_entries() <= _ops.FUSED_MAX_ROW)
        kernel_label = ("mp::agg_dense_pc_kernel<..., AGG_ONLY> through mp_agg_rows_tiles_f32 (64-row tiles gathered into "
                        "two LDS buffers by 4 waves of a workgroup, stored by 4 others; what ops.spmm dispatches for "
                        "sum / mean at d = 128 / 256 / 512)" if tiles else
                        "mp::agg_rows_kernel<4,SUM,weighted> (+ hub pieces/finalize, same launch group)")
        gname = "BA" if args.graph == "ba" else "HK0.3"
        workload = f"gcn_norm_sum_d{d}_{gname}_n{n}_m{args.m}" + ("_perm" if args.permute else "")
        traffic, traffic_source = pmc_traffic(workload)
        res = {
            "metric": METRIC,
            "value": total_nnz * args.steps / dt,
            "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "nodes_per_gpu": n, "stored_entries_per_gpu": g.nnz,
                       "feature_dim": d, "reduce": "sum", "edge_weights": "D^-1/2 (A+I) D^-1/2",
                       "graph": (f"Barabasi-Albert BA({n},{args.m})" if args.graph == "ba" else
                                 f"Holme-Kim powerlaw_cluster({n},{args.m},0.3)") +
                                " seed 12345+rank, symmetrised, deduplicated, self loops added" +
                                (", nodes randomly relabelled" if args.permute else ""),
                       "index_dtype": "int32",
                       "output_placement": place,
                       "parallelism": f"{world} independent graph(s), one per GPU, no data-path collective"},
            "hbm_gbps_algorithmic": achieved,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "stream_read_gbps_measured": read_gbps, "stream_copy_gbps_measured": copy_gbps,
                         "frac_of_measured_read_stream": achieved / read_gbps,
                         "algorithmic_bytes_per_launch": balg, "launch_ms": launch_ms,
                         "launch_ms_min_median_max": [per_step[0], per_step[len(per_step) // 2], per_step[-1]],
                         "launch_ms_p10_p90": [pct(0.1), pct(0.9)],
                         "launch_ms_cold_median": cold[len(cold) // 2] if cold else None,
                         "cold_note": "512 MB memset between launches (L2 / Infinity Cache flushed)",
                         "kernel": kernel_label},
        }
        if backward is not None:
            res["backward"] = backward
        if layer is not None:
            res["layer"] = layer
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(g, x)
        if world > 1:
            # the whole job against the whole job's roof: N independent HBM systems
            rf = res["roofline"]
            rf["achieved_all_gpus"] = balg_total / (res["ms_per_step"] * 1e-3) / 1e9
            rf["peak_all_gpus"] = HBM_PEAK_GBS * world
            rf["frac_all_gpus"] = rf["achieved_all_gpus"] / rf["peak_all_gpus"]
            rf["note"] = ("achieved / frac: rank 0's launches by HIP events; *_all_gpus: every rank's algorithmic bytes over "
                          "the barrier-bracketed max-over-ranks step time, against world x 8 TB/s")
    D.barrier()
    return res if rank == 0 else None


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it (the driver's command): start N fresh rank processes of this
    same script — one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment — relay rank 0's stdout
    (the one JSON line) and return the first non-zero exit code.  This parent never touches the GPU and never replaces
    itself (no exec of a process that has initialised HIP); a rank that dies takes the others down (exact PIDs) instead of
    leaving them in a collective."""
    import signal
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MP_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from graphgym_amd import hostcpu
        env.setdefault("OMP_NUM_THREADS", str(max(1, hostcpu.effective_cpus() // n)))
        # rank 0's stdout is the protocol; the other ranks' stdout joins stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr, cwd=os.getcwd()))
    rc = 0
    alive = list(procs)
    try:
        while alive:
            time.sleep(0.2)
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in alive:           # the others would wait for it in a barrier forever
                        q.send_signal(signal.SIGTERM)
            if rc != 0 and alive:
                t_end = time.time() + 15
                while alive and time.time() < t_end:
                    time.sleep(0.2)
                    alive = [q for q in alive if q.poll() is None]
                for q in alive:
                    q.kill()
                alive = []
    except KeyboardInterrupt:
        for q in alive:
            q.kill()
        raise
    return rc if rc >= 0 else 128 - rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=["auto", "aggregate", "step", "both"], default="auto",
                    help="auto = aggregate on one GPU; on N > 1 GPUs the aggregate line plus a `step` object (the "
                         "data-parallel training step that contains the RCCL gradient all-reduce)")
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--m", type=int, default=5)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--graph", choices=["ba", "powerlaw_cluster"], default="ba",
                    help="ba = Barabasi-Albert (default); powerlaw_cluster = Holme-Kim with triangle probability 0.3")
    ap.add_argument("--permute", action="store_true", help="relabel nodes by a random permutation (seed 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--centres", type=int, default=4096, help="step: ego-net centres per GPU in the global batch")
    ap.add_argument("--step-model", choices=["idgcn", "idgin"], default="idgcn")
    ap.add_argument("--step-nodes", type=int, default=2_000_000, help="step: nodes of the base graph the ego nets are cut from")
    ap.add_argument("--step-steps", type=int, default=20, help="timed steps of the `step` object in --mode both")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # no launcher set the ranks up: be the launcher (before anything in this process touches the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    from graphgym_amd import dist as D

    mode = args.mode
    if mode == "auto":
        mode = "aggregate" if args.gpus == 1 else "both"
    if mode == "step" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # a one-rank step still EXECUTES its exchange (a one-rank RCCL group runs the same communicator set-up, stream
        # hand-off and async work objects as an 8-rank one): nothing RCCL-side is first run when the node appears
        os.environ.setdefault("MP_DIST_FORCE", "1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
    rank, local, world = D.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", torch.cuda.current_device())
    res = None
    if mode in ("aggregate", "both"):
        res = run_aggregate(args, rank, world, dev)
    if mode in ("step", "both"):
        from graphgym_amd import bench_step
        if mode == "both":
            import copy
            sargs = copy.copy(args)
            sargs.steps, sargs.warmup = args.step_steps, 3
            torch.cuda.empty_cache()
            step = bench_step.run(sargs, rank, world, dev)
            if rank == 0:
                res["step"] = step
        else:
            res = bench_step.run(args, rank, world, dev)
    if rank == 0:
        print(json.dumps(res), flush=True)
    D.barrier()
    if D.dist.is_available() and D.dist.is_initialized():
        D.dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Host-side logic that needs no GPU: registry semantics, config, layer construction,
sharding, and the N>1 gradient exchange over gloo (world_size 2)."""
import os
import socket
import textwrap

import pytest
import torch
import torch.multiprocessing as mp


def test_registry_semantics_match_reference():
    from graphgym_amd import registry as R
    d = {}
    R.register("a", int, d)
    with pytest.raises(KeyError, match="already pre-defined"):
        R.register("a", float, d)                                    # register.py:6-10
    import graphgym_amd.graphgym_plugin as plugin
    for key in ["gcnconv", "sageconv", "gatconv", "ginconv", "generalconv", "idconv", "gcnidconv",
                "sageidconv", "gatidconv", "ginidconv", "Tfg-gcnconv", "Tfg-sageconv", "Tfg-gatconv",
                "Tfg-ginconv", "Tfg-idgcn", "Tfg-idsage", "Tfg-idgat", "Tfg-idgin"]:
        assert R.layer_dict[key] is plugin.ALL_KEYS[key]
    with pytest.raises(KeyError):
        R.register_layer("gcnidconv", object)
    assert plugin.install(override=True) == list(plugin.ALL_KEYS)    # re-import / re-install is idempotent


def test_config_yaml_keys(tmp_path):
    from graphgym_amd import config
    y = tmp_path / "c.yaml"
    y.write_text(textwrap.dedent("""
        out_dir: results
        dataset: {format: nx, name: scalefree, transform: ego}
        gnn: {layers_mp: 3, dim_inner: 128, layer_type: Tfg-idgcn, agg: mean, l2norm: True}
        optim: {base_lr: 0.01}
    """))
    c = config.load_cfg(str(y), target=config._defaults())
    assert c.gnn.layer_type == "Tfg-idgcn" and c.gnn.dim_inner == 128 and c.gnn.agg == "mean"
    assert c.dataset.transform == "ego" and c.gnn.normalize_adj is False and c.bn.eps == 1e-5


def test_layer_parameters_are_named_and_shaped_like_the_reference():
    from graphgym_amd import layers as L
    m = L.GCNIDConv(8, 16, bias=True)
    assert sorted(n for n, _ in m.named_parameters()) == ["model.bias", "model.weight", "model.weight_id"]
    assert m.model.weight.shape == (8, 16) and float(m.model.bias.abs().sum()) == 0.0   # zeros init
    assert L.GCNIDConv(8, 16).model.bias is None                                         # bias=False default
    m = L.SAGEIDConv(8, 16)
    assert m.model.weight.shape == (16, 16) and m.model.concat                           # 2*in with concat
    m = L.GATIDConv(8, 16, bias=True)
    assert m.model.att.shape == (1, 1, 32)
    m = L.GINIDConv(8, 16)
    assert isinstance(m.model.nn[0], torch.nn.Linear) and float(m.model.eps) == 0.0
    assert "eps" in dict(m.model.named_buffers())
    m = L.TfgIDSAGE(8, 16, bias=True)
    assert m.model.self_kernel.shape == (8, 8) and m.model.id_kernel.shape == (8, 8) and m.model.bias.shape == (16,)
    m = L.IDGCN(16)                                                                      # keras-style lazy build
    assert m.kernel is None
    m.build(5)
    assert m.kernel.shape == (5, 16) and m.kernel_id.shape == (5, 16)
    bound = (6.0 / (5 + 16)) ** 0.5
    assert float(m.kernel.abs().max()) <= bound                                          # glorot_uniform


def test_lpt_partition_balances_nnz():
    from graphgym_amd.dist import lpt_partition
    costs = [100, 1, 1, 1, 50, 49, 2, 98]
    parts = lpt_partition(costs, 2)
    assert sorted(i for p in parts for i in p) == list(range(8))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 2
    assert lpt_partition(costs, 2) == parts
    assert lpt_partition([5], 4) == [[0], [], [], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from graphgym_amd import dist as D
    r, _, w = D.init_from_env(device_type="cpu")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.ReLU(), torch.nn.Linear(3, 2))
    # each rank owns different units -> different gradients
    units = D.lpt_partition([10, 20, 30, 40], w)[r]
    x = torch.arange(4 * 4, dtype=torch.float32).view(4, 4)[units] / 10.0
    model(x).sum().backward()
    local = [p.grad.clone() for p in model.parameters()]
    bucket = D.GradBucket(model.parameters())
    bucket.all_reduce_mean()
    D.barrier()
    tmax = D.all_reduce_max(float(r + 1), torch.device("cpu"))
    tsum = D.all_reduce_sum(float(len(units)), torch.device("cpu"))
    torch.save({"local": local, "avg": [p.grad.clone() for p in model.parameters()], "tmax": tmax,
                "tsum": tsum, "units": units}, os.path.join(out, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_gradient_all_reduce_gloo_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "r0.pt", weights_only=True)
    b = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert sorted(a["units"] + b["units"]) == [0, 1, 2, 3]
    for la, lb, ga, gb in zip(a["local"], b["local"], a["avg"], b["avg"]):
        assert torch.allclose(ga, (la + lb) / 2, atol=1e-6)
        assert torch.equal(ga, gb)                                   # ranks end with identical gradients
    assert a["tmax"] == 2.0 and a["tsum"] == 4.0


def test_bench_refuses_to_run_without_a_gpu():
    """no CPU path: on a box without a HIP device bench.py exits with a message instead of a fallback"""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)


def test_order_choice_and_fused_shape_gate(monkeypatch):
    """'auto' gathers at the narrower width and aggregates first at equal widths where the one-kernel
    aggregate -> transform applies; the shape gate and its MP_FUSED=0 switch are host logic"""
    import types
    import torch
    from graphgym_amd import layers, ops
    assert layers._pick_order("auto", 64, 256) == "aggregate_first"
    assert layers._pick_order("auto", 256, 64) == "transform_first"
    assert layers._pick_order("auto", 256, 256) == "aggregate_first"      # fused kernel width
    assert layers._pick_order("auto", 100, 100) == "transform_first"      # not a fused width
    assert layers._pick_order("transform_first", 64, 256) == "transform_first"
    g = types.SimpleNamespace(nnz=10, max_row_entries=lambda: 5)
    x, W = torch.zeros(8, 256), torch.zeros(256, 64)
    monkeypatch.delenv("MP_FUSED", raising=False)
    assert ops.agg_dense_supported(g, x, W)
    assert not ops.agg_dense_supported(g, torch.zeros(8, 96), torch.zeros(96, 64))     # width
    assert not ops.agg_dense_supported(g, x, torch.zeros(256, 7))                       # odd d_out
    assert not ops.agg_dense_supported(types.SimpleNamespace(nnz=0, max_row_entries=lambda: 0), x, W)   # empty operator
    assert not ops.agg_dense_supported(types.SimpleNamespace(nnz=10, max_row_entries=lambda: 1 << 20), x, W)   # star-like hub
    monkeypatch.setenv("MP_FUSED", "0")
    assert not ops.agg_dense_supported(g, x, W)


def _overlap_worker(rank, world, port, out):
    """two ranks with different numbers of labelled examples: globally normalised losses + the overlapped two-bucket
    exchange give the full-batch gradient on every rank (a mean of per-rank mean losses would not)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from graphgym_amd import dist as D
    D.init_from_env("cpu")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(10, 6, generator=g), torch.randint(0, 3, (10,), generator=g)
    lo, hi = (0, 7) if rank == 0 else (7, 10)                        # 7 examples on rank 0, 3 on rank 1
    total = D.global_count(hi - lo, torch.device("cpu"))
    bucket = D.GradBucket(model.parameters(), n_buckets=2).attach()
    assert len(bucket.buckets) == 2 and all(p.grad is not None for p in model.parameters())
    for _ in range(2):                                               # second step: views survive zero_grad
        bucket.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(X[lo:hi]), Y[lo:hi], reduction="sum") / total
        loss.backward()
        bucket.finish(1.0)
    torch.save({"grads": [p.grad.clone() for p in model.parameters()], "total": total},
               os.path.join(out, f"o{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_overlapped_buckets_give_the_full_batch_gradient_gloo_world2(tmp_path):
    port = _free_port()
    mp.spawn(_overlap_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "o0.pt", weights_only=True)
    b = torch.load(tmp_path / "o1.pt", weights_only=True)
    assert a["total"] == 10.0
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(10, 6, generator=g), torch.randint(0, 3, (10,), generator=g)
    torch.nn.functional.cross_entropy(model(X), Y, reduction="mean").backward()
    for p, ga, gb in zip(model.parameters(), a["grads"], b["grads"]):
        assert torch.allclose(ga, p.grad, atol=1e-6) and torch.equal(ga, gb)


def test_bench_starts_its_own_ranks_and_relays_their_exit_code():
    """`python bench.py --gpus 2` without a launcher (the driver's command) must start two rank processes itself.  Here
    there is no GPU, so both ranks end with bench.py's "needs a HIP device" exit — which shows that two ranks were started,
    that the process group formed (they got past init), that the parent relayed a non-zero code and that nothing hung."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered on the GPU by tests/test_bench_gpu.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert r.stderr.count("needs a HIP device") == 2, r.stderr[-1500:]
    assert r.stdout.strip() == ""


@pytest.mark.parametrize("bounds", [[0, 3, 3, 10], [0, 5, 7], [0, 4, 8, 12], [0, 1, 9, 10, 10]])
def test_ragged_row_partition_pad_and_unpad_helpers(bounds):
    """the index arithmetic of the row-partitioned exchange (dist._gather_rows / _scatter_sum_rows): a one-rank RCCL
    group can never be ragged, so the padded layouts are checked here against torch.cat / slicing references"""
    from graphgym_amd import dist as D
    world, n, d = len(bounds) - 1, bounds[-1], 3
    max_rows = max(bounds[p + 1] - bounds[p] for p in range(world))
    full = torch.arange(n * d, dtype=torch.float32).view(n, d) + 1
    # all-gather: every rank pads its rows, the concatenation of the padded chunks is un-padded to the full matrix
    chunks = [D.pad_rows(full[bounds[p]:bounds[p + 1]], max_rows) for p in range(world)]
    for p, c in enumerate(chunks):
        assert c.shape == (max_rows, d) and bool((c[bounds[p + 1] - bounds[p]:] == 0).all())
    assert torch.equal(D.unpad_gathered(torch.cat(chunks, 0), bounds, max_rows), full)
    # reduce-scatter: chunk p of the padded layout holds rank p's rows; summing the layouts of all ranks and slicing
    # chunk p gives rank p the sum of everybody's partial rows
    partials = [full * (r + 1) for r in range(world)]
    summed = sum(D.pad_chunks(x, bounds, max_rows) for x in partials)
    want = sum(partials)
    for p in range(world):
        assert torch.equal(summed[p, :bounds[p + 1] - bounds[p]], want[bounds[p]:bounds[p + 1]])
        assert bool((summed[p, bounds[p + 1] - bounds[p]:] == 0).all())


def _ragged_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphgym_amd import dist as D
    import types
    bounds = [0, 2, 2, 9]                                      # ragged, with an empty range
    part = types.SimpleNamespace(world=world, rank=rank, bounds=bounds, max_rows=7, rows=(bounds[rank], bounds[rank + 1]))
    full = torch.arange(9 * 4, dtype=torch.float32).view(9, 4)
    got = D._gather_rows(part, full[bounds[rank]:bounds[rank + 1]].clone())
    ok = torch.equal(got, full)
    mine = D._scatter_sum_rows(part, full * (rank + 1))
    ok = ok and torch.equal(mine, full[bounds[rank]:bounds[rank + 1]] * 6)      # 1 + 2 + 3
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_ragged_gather_and_scatter_sum_over_gloo_world3():
    """the same branches (padding, equal chunks, un-padding) under a real process group with THREE ranks and ragged
    bounds — the tensor collectives gloo offers (all_gather_into_tensor; all_reduce of the padded layout)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ragged_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True), (1, True), (2, True)]


def test_engine_lpt_matches_the_python_lpt_and_balances_heavy_tails():
    import numpy as np
    from graphgym_amd import dist as D
    rng = np.random.RandomState(0)
    costs = (rng.pareto(1.5, 4096) * 100 + 10).astype(np.int64)            # heavy-tailed, like ego-net sizes
    costs[100:110] = costs[100]                                             # ties
    owner = D.lpt_owners(costs, 8)
    parts = D.lpt_partition(costs.tolist(), 8)
    for r, p in enumerate(parts):
        assert np.array_equal(np.nonzero(owner == r)[0], np.asarray(p))     # the same assignment, tie-breaks included
    loads = np.bincount(owner, weights=costs.astype(np.float64), minlength=8)
    assert loads.max() / loads.mean() < 1.01
    assert D.lpt_owners(np.zeros(0, dtype=np.int64), 4).size == 0


def test_inactive_grad_bucket_allows_gradient_accumulation():
    """a single process exchanges nothing: an attached bucket must not get in the way of two backward passes per step"""
    from graphgym_amd import dist as D
    lin = torch.nn.Linear(4, 3)
    bucket = D.GradBucket(lin.parameters(), n_buckets=2).attach()
    bucket.zero_grad()
    x = torch.randn(5, 4)
    lin(x).sum().backward()
    lin(x).sum().backward()                                                  # accumulates into the bucket views
    bucket.finish(1.0)
    ref = torch.nn.Linear(4, 3)
    ref.load_state_dict(lin.state_dict())
    (ref(x).sum() * 2).backward()
    assert torch.allclose(lin.weight.grad, ref.weight.grad) and torch.allclose(lin.bias.grad, ref.bias.grad)


def test_effective_cpus_respects_the_cgroup_quota_and_fits_torch():
    """hostcpu: the usable CPU count is bounded by affinity and cgroup quota; fit_torch_threads never raises the pool"""
    from graphgym_amd import hostcpu
    n = hostcpu.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            assert n <= max(1, int(int(q) / int(p)))
    except OSError:
        pass
    before = torch.get_num_threads()
    assert hostcpu.fit_torch_threads() <= max(before, n)


def test_quiet_gc_disables_automatic_collection_and_restores_it():
    import gc
    from graphgym_amd.pipeline import quiet_gc
    assert gc.isenabled()
    with quiet_gc() as tick:
        assert not gc.isenabled()
        for _ in range(20):
            tick()
    assert gc.isenabled()


def test_allocator_settings_for_changing_batch_shapes(monkeypatch):
    """pipeline.fit_allocator_to_changing_shapes: sets torch's roundup_power2_divisions at run time (what a loop with new
    batch shapes every step needs, bench_step's driver_allocs_in_fresh_steps) and can be switched off"""
    from graphgym_amd.pipeline import fit_allocator_to_changing_shapes
    assert fit_allocator_to_changing_shapes(8) in ("roundup_power2_divisions:8", None)
    monkeypatch.setenv("MP_KEEP_ALLOCATOR", "1")
    assert fit_allocator_to_changing_shapes(8) is None

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

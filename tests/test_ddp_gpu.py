"""The N > 1 path on the real kernels: two ranks (sharing the box's one GPU, gloo for the exchange —
RCCL refuses two ranks on one device) each own a shard of the batch's graphs, run the ID-GCN step on
the engine, all-reduce gradients through GradBucket, and must end with the gradient of the full batch."""
import os
import socket

import networkx as nx
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _graphs():
    return [nx.powerlaw_cluster_graph(24 + 4 * (i % 3), 2, 0.3, seed=i) for i in range(6)]


def _union(graphs, dev):
    parts, off = [], 0
    for G in graphs:
        e = np.array(list(G.edges()), dtype=np.int64) + off
        parts.append(np.concatenate([e, e[:, ::-1]]))
        off += G.number_of_nodes()
    return torch.from_numpy(np.concatenate(parts).T.copy()).to(dev), off


def _loss_and_grads(graphs, labels, total_centres, dev, bucket_fn=None):
    import graphgym_amd as ga
    from graphgym_amd import harness as H
    from graphgym_amd.ego import ego_batch
    torch.manual_seed(7)
    model = H.TfgNodeModel("idgcn", 3, 16, 4).to(dev)
    ei, n = _union(graphs, dev)
    base = ga.CSRGraph.from_edge_index(ei, n)
    ei2, orig, ids, _ = ego_batch(base, torch.arange(n, device=dev), 2)
    # per-node features that depend only on the node's own graph (identical in a shard and in the full batch)
    rows = [[v / G.number_of_nodes(), G.degree(v) / 10.0, nx.clustering(G, v)] for G in graphs for v in range(G.number_of_nodes())]
    feats = torch.tensor(rows, dtype=torch.float32, device=dev)
    x = feats[orig]
    logits = model([x, ei2, ids], holder=H.Batch())
    y = torch.as_tensor(labels, device=dev)
    loss = torch.nn.functional.cross_entropy(logits[ids], y, reduction="sum") / total_centres
    loss.backward()
    if bucket_fn is not None:
        bucket_fn(model)
    return [p.grad.detach().cpu().clone() for p in model.parameters()]


def _labels(graphs):
    return [int(v) % 4 for G in graphs for v in range(G.number_of_nodes())]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MP_SHARE_DEVICE="1", MP_DIST_BACKEND="gloo")
    from graphgym_amd import dist as D
    r, _, w = D.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    graphs = _graphs()
    costs = [2 * G.number_of_edges() for G in graphs]
    mine = D.lpt_partition(costs, w)[r]
    total = sum(G.number_of_nodes() for G in graphs)
    sub = [graphs[i] for i in mine]

    def exchange(model):
        b = D.GradBucket(model.parameters())
        b.all_reduce_mean()
        for p in model.parameters():       # mean over ranks of per-rank sums/total -> x world = full-batch gradient
            p.grad.mul_(w)
    grads = _loss_and_grads(sub, _labels(sub), total, dev, exchange)
    torch.save({"grads": grads, "mine": mine}, os.path.join(out, f"r{rank}.pt"))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_equals_full_batch(dev, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "r0.pt", weights_only=True)
    b = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert sorted(a["mine"] + b["mine"]) == list(range(6))
    graphs = _graphs()
    full = _loss_and_grads(graphs, _labels(graphs), sum(G.number_of_nodes() for G in graphs), dev)
    for ga_, gb_, gf in zip(a["grads"], b["grads"], full):
        assert torch.equal(ga_, gb_)
        assert float((ga_ - gf).abs().max()) <= 1e-5 * max(1.0, float(gf.abs().max()))


def _part_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MP_SHARE_DEVICE="1", MP_DIST_BACKEND="gloo")
    import graphgym_amd as ga
    from graphgym_amd import dist as D, graphgen
    r, _, w = D.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    n, d = 3000, 48
    ei = graphgen.ba_edge_index(n, 4, seed=5, device=dev)
    g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
    part = D.RowPartition(g)
    gen = torch.Generator().manual_seed(1)
    h = torch.randn(n, d, generator=gen).to(dev)
    W1 = torch.randn(d, d, generator=gen).to(dev).requires_grad_(True)
    r0, r1 = part.rows
    h_loc = h[r0:r1].clone().requires_grad_(True)
    # two partitioned layers: aggregate (halo exchange) then transform, rows stay partitioned
    z = torch.relu(D.halo_aggregate(part, h_loc, "sum") @ W1)
    z = D.halo_aggregate(part, z, "mean")
    dy = torch.randn(n, d, generator=gen).to(dev)
    (z * dy[r0:r1]).sum().backward()
    b = D.GradBucket([W1]); b.all_reduce_mean(); W1.grad.mul_(w)
    torch.save({"z": z.detach().cpu(), "dh": h_loc.grad.cpu(), "dW": W1.grad.cpu(), "rows": (r0, r1),
                "bounds": part.bounds, "nnz_local": part.local.nnz}, os.path.join(out, f"p{rank}.pt"))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_row_partitioned_graph_matches_single_process(dev, tmp_path):
    """one graph split by destination rows over 2 ranks (feature all-gather forward, reduce-scatter of
    dH backward) == the unpartitioned computation"""
    import graphgym_amd as ga
    from graphgym_amd import graphgen, ops
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_part_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "p0.pt", weights_only=True)
    b = torch.load(tmp_path / "p1.pt", weights_only=True)
    n, d = 3000, 48
    ei = graphgen.ba_edge_index(n, 4, seed=5, device=dev)
    g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
    gen = torch.Generator().manual_seed(1)
    h = torch.randn(n, d, generator=gen).to(dev).requires_grad_(True)
    W1 = torch.randn(d, d, generator=gen).to(dev).requires_grad_(True)
    z = ops.spmm(g, torch.relu(ops.spmm(g, h, "sum") @ W1), "mean")
    dy = torch.randn(n, d, generator=gen).to(dev)
    (z * dy).sum().backward()
    assert a["bounds"] == b["bounds"] and a["bounds"][0] == 0 and a["bounds"][-1] == n
    assert abs(a["nnz_local"] - b["nnz_local"]) <= 0.05 * g.nnz          # nnz-balanced ranges
    zz = torch.cat([a["z"], b["z"]]); dh = torch.cat([a["dh"], b["dh"]])
    tol = lambda ref: 1e-5 * max(1.0, float(ref.abs().max()))
    assert float((zz - z.detach().cpu()).abs().max()) <= tol(z.detach())
    assert float((dh - h.grad.cpu()).abs().max()) <= tol(h.grad)
    assert float((a["dW"] - W1.grad.cpu()).abs().max()) <= 10 * tol(W1.grad)
    assert torch.equal(a["dW"], b["dW"])

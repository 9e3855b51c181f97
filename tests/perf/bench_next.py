"""Measurements for the rows SURVEY §8(f) marks "next", each beside the CPU oracle on the same box:

  1. COO -> CSR build with self loops + GCN normalisation + plan + transpose (rank 2), C2 and C4 sizes
  2. ego-net expansion (rank 1): all centres of a batch of 64-node graphs (the reference's use),
     and a batch of centres on a 10^6-node graph
  3. one ID-GCN training step on an ego batch (config C3 shape, d = 128)

    python tests/perf/bench_next.py > profiles/rNN_next.jsonl

Lives under tests/ because it times the CPU oracle next to the engine (oracle/ is test infrastructure).
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import networkx as nx
import numpy as np
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, harness as H
from graphgym_amd.ego import ego_batch
from oracle import ref_layers as RL, ref_ops as R

dev = torch.device("cuda:0")
torch.set_num_threads(min(16, os.cpu_count() or 1))

def sync_time(fn, iters=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters, r

def emit(**kw):
    print(json.dumps(kw), flush=True)

def csr_build(n):
    ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
    E = ei.size(1)
    def build():
        g = ga.CSRGraph.from_edge_index(ei, n, add_self_loops=True).gcn_norm("row")
        g.plan(); g.transpose()
        return g
    t, g = sync_time(build)
    # CPU oracle: the per-layer-call work of gcn_norm_adj (TfgIDLayer.py:528-566) on a sample of the edges
    take = min(E, 20_000_000)
    eic = torch.stack([ei[1, :take], ei[0, :take]]).cpu()
    t0 = time.perf_counter()
    R.gcn_norm_adj(R.SparseAdj(eic, None, [n, n]))
    tc = (time.perf_counter() - t0) * E / take
    emit(what="csr_build+selfloops+gcn_norm+plan+transpose", nodes=n, input_edges=E, stored_entries=g.nnz,
         gpu_ms=t * 1e3, gpu_edges_per_s=E / t,
         cpu_oracle_ms_per_layer_call=tc * 1e3, cpu_note=f"gcn_norm_adj only (no sort), extrapolated from {take} edges, {torch.get_num_threads()} threads")
    del g, ei
    torch.cuda.empty_cache()

def ego_small(n_graphs=16, radius=3):
    # a GraphGym batch: 16 x 64-node scale-free graphs (config C1/C3 shape), every node a centre
    graphs = [nx.powerlaw_cluster_graph(64, 4, 0.3, seed=s) for s in range(n_graphs)]
    offs, parts = 0, []
    for G in graphs:
        e = np.array(list(G.edges()), dtype=np.int64) + offs
        parts.append(np.concatenate([e, e[:, ::-1]]))
        offs += 64
    base_ei = torch.from_numpy(np.concatenate(parts).T.copy()).to(dev)
    n = offs
    base = ga.CSRGraph.from_edge_index(base_ei, n)
    cen = torch.arange(n, device=dev)
    t, (ei, orig, ids, ego_of) = sync_time(lambda: ego_batch(base, cen, radius))
    t0 = time.perf_counter()
    for G in graphs:
        RL.ego_nets(G, radius)
    tc = time.perf_counter() - t0
    emit(what="ego_nets batch of 16 x 64-node graphs", radius=radius, centres=n, expanded_nodes=int(orig.numel()),
         expanded_edges=int(ei.size(1)), gpu_ms=t * 1e3, cpu_oracle_ms=tc * 1e3, speedup=tc / t,
         cpu_note="oracle = networkx restatement of transform.py:11-38, 1 thread (pure Python)")
    return ei, orig, ids, n

def ego_large(n=1_000_000, B=64, radius=2):
    ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
    base = ga.CSRGraph.from_edge_index(ei, n)
    gen = torch.Generator().manual_seed(1)
    cen = torch.randint(0, n, (B,), generator=gen).to(dev)
    t, (e2, orig, ids, ego_of) = sync_time(lambda: ego_batch(base, cen, radius))
    emit(what="ego batch on BA(1e6,5)", radius=radius, centres=B, expanded_nodes=int(orig.numel()),
         expanded_edges=int(e2.size(1)), gpu_ms=t * 1e3, expanded_edges_per_s=e2.size(1) / t)

def idgcn_step(d=128, f_in=1):
    ei, orig, ids, n = ego_small(16, 3)
    N = orig.numel()
    x = torch.ones(N, f_in, device=dev)                      # node_feature = ones, as the bundled datasets
    labels = torch.randint(0, 10, (n,), device=dev)
    batch = H.Batch(edge_index=ei, node_id_index=ids)
    model = H.TfgNodeModel("idgcn", f_in, d, 10).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    def fl():
        return H.tfg_loss(model([x, ei, ids], holder=batch), ids, labels, model.kernel_parameters())
    t, _ = sync_time(lambda: H.train_step(model, opt, fl), iters=20)
    # CPU oracle: same model restated (forward + backward), torch CPU
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    xc, eic, idc, lc = x.cpu(), ei.cpu(), ids.cpu(), labels.cpu()
    def cpu_step():
        h = xc
        for i in range(3):
            h = RL.gcn_id(h, eic, idc, None, P[f"convs.{i}.kernel"], P[f"convs.{i}.kernel_id"], P[f"convs.{i}.bias"], "relu")
        h = torch.relu(h @ P["mlp.1.weight"].t() + P["mlp.1.bias"]) @ P["mlp.3.weight"].t() + P["mlp.3.bias"]
        loss = torch.nn.functional.cross_entropy(h[idc], lc)
        loss.backward()
    cpu_step()
    t0 = time.perf_counter()
    for _ in range(3):
        cpu_step()
    tc = (time.perf_counter() - t0) / 3
    emit(what="idgcn_tf training step on an ego batch (16 x 64-node graphs, radius 3)", d=d, nodes=int(N),
         edges=int(ei.size(1)), gpu_ms_per_step=t * 1e3, cpu_oracle_ms_per_step=tc * 1e3, speedup=tc / t,
         cpu_threads=torch.get_num_threads())

def cora_like_step(d=128, centres=128, radius=2):
    """config C3 shape with a synthetic stand-in for Cora (the dataset is not available offline):
    N = 2708, 10556 directed edges, F = 1433, 7 classes; ID-GCN Full on a batch of 128 ego nets, radius 2."""
    n, f_in, classes = 2708, 1433, 7
    G = nx.gnm_random_graph(n, 5278, seed=3)
    e = np.array(list(G.edges()), dtype=np.int64)
    base_ei = torch.from_numpy(np.concatenate([e, e[:, ::-1]]).T.copy()).to(dev)
    base = ga.CSRGraph.from_edge_index(base_ei, n)
    gen = torch.Generator().manual_seed(0)
    feats = (torch.rand(n, f_in, generator=gen) < 0.0127).float().to(dev)       # sparse bag-of-words density of Cora
    labels_all = torch.randint(0, classes, (n,), generator=gen).to(dev)
    cen = torch.randperm(n, generator=gen)[:centres].to(dev)
    t_ego, (ei, orig, ids, _) = sync_time(lambda: ego_batch(base, cen, radius))
    x = feats[orig]
    labels = labels_all[cen]
    batch = H.Batch(edge_index=ei, node_id_index=ids)
    model = H.TfgNodeModel("idgcn", f_in, d, classes).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    def fl():
        return H.tfg_loss(model([x, ei, ids], holder=batch), ids, labels, model.kernel_parameters())
    t, _ = sync_time(lambda: H.train_step(model, opt, fl), iters=20)
    emit(what="idgcn_tf step, Cora-like synthetic (N=2708, E=10556, F=1433), 128 ego nets radius 2", d=d,
         batch_nodes=int(orig.numel()), batch_edges=int(ei.size(1)), ego_build_ms=t_ego * 1e3, gpu_ms_per_step=t * 1e3)


def identity_features(n_graphs=256, k=10):
    """`node_identity` (ID-GNN Fast) for a whole dataset of 64-node graphs (scalefree.pkl holds 256 of them):
    the reference runs compute_identity once per graph on the CPU (dense n x n powers)"""
    from graphgym_amd.identity import compute_identity
    eis, batch, off = [], [], 0
    graphs = [nx.powerlaw_cluster_graph(64, 4, 0.3, seed=s) for s in range(n_graphs)]
    per_graph = []
    for gi, G in enumerate(graphs):
        e = torch.tensor(list(G.edges()), dtype=torch.int64).t()
        ei = torch.cat([e, e.flip(0)], dim=1)
        per_graph.append(ei)
        eis.append(ei + off)
        batch.append(torch.full((64,), gi, dtype=torch.int64))
        off += 64
    ei, batch = torch.cat(eis, dim=1).to(dev), torch.cat(batch).to(dev)
    t_gpu, _ = sync_time(lambda: compute_identity(ei, off, k, batch=batch))
    t0 = time.perf_counter()
    ref = torch.cat([RL.compute_identity(e, 64, k) for e in per_graph])
    t_cpu = time.perf_counter() - t0
    err = float((compute_identity(ei, off, k, batch=batch).cpu() - ref).abs().max())
    emit(what=f"node_identity features, {n_graphs} graphs x 64 nodes, k={k}", nodes=off, edges=int(ei.size(1)),
         gpu_ms=t_gpu * 1e3, cpu_oracle_ms=t_cpu * 1e3, speedup=t_cpu / t_gpu, max_abs_diff=err,
         cpu_note="oracle = dense restatement of identity.py:25-35 per graph, torch CPU")
    # one larger graph (Cora-sized): width n, blocked
    G = nx.barabasi_albert_graph(2708, 2, seed=5)
    e = torch.tensor(list(G.edges()), dtype=torch.int64).t()
    ei1 = torch.cat([e, e.flip(0)], dim=1)
    t_gpu, _ = sync_time(lambda: compute_identity(ei1.to(dev), 2708, k))
    t0 = time.perf_counter()
    ref = RL.compute_identity(ei1, 2708, k)
    t_cpu = time.perf_counter() - t0
    err = float((compute_identity(ei1.to(dev), 2708, k).cpu() - ref).abs().max())
    emit(what=f"node_identity features, one 2708-node graph (Cora-sized), k={k}", gpu_ms=t_gpu * 1e3,
         cpu_oracle_ms=t_cpu * 1e3, speedup=t_cpu / t_gpu, max_abs_diff=err)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "identity":
        identity_features()
        sys.exit(0)
    identity_features()
    cora_like_step()
    csr_build(1_000_000)
    csr_build(10_000_000)
    ego_large()
    idgcn_step()

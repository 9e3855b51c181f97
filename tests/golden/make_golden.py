"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

PARITY UNPINNED: the reference ships no golden tensors and cannot be imported here
(DESIGN.md §3), so these vectors are outputs of oracle/ (the op-for-op restatement of
its CPU path), frozen so that (a) the oracle cannot drift silently and (b) the GPU box,
which has no /root/reference, checks against fixed data.  The input graphs are synthetic
graphs of the shape of the reference's bundled datasets (64-node
networkx.powerlaw_cluster_graph, datasets/syn_graph.py:42) — the bundled *.pkl files are
pickles and are deliberately not loaded.
"""
import os
import sys

import networkx as nx
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_layers as RL  # noqa: E402
from oracle import ref_ops as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def directed(G):
    """both directions of every undirected edge, PyG convention [source; destination]"""
    e = np.array(list(G.edges()), dtype=np.int64).reshape(-1, 2)
    ei = np.concatenate([e, e[:, ::-1]], axis=0).T
    return np.ascontiguousarray(ei)


HAND_GRAPHS = {
    # name: (num_nodes, edge_index [2,E] as [source; destination])
    "path4": (4, directed(nx.path_graph(4))),
    "star5": (5, directed(nx.star_graph(4))),
    "triangle_pendant": (4, directed(nx.Graph([(0, 1), (1, 2), (2, 0), (2, 3)]))),
    "two_components_isolated": (6, directed(nx.Graph([(0, 1), (1, 2), (3, 4)]))),        # node 5 isolated
    "multigraph_selfloop": (3, np.array([[0, 1, 1, 2, 2, 0, 1], [1, 0, 0, 2, 1, 2, 1]], dtype=np.int64)),
}


def gen_aggregation():
    """K3-K9: sum / mean / max aggregation, degrees, GCN normalisation in both flavours"""
    rec = {}
    g = torch.Generator().manual_seed(20240501)
    graphs = dict(HAND_GRAPHS)
    for s in range(4):
        G = nx.powerlaw_cluster_graph(64, 2 + s, 0.3, seed=100 + s)
        graphs[f"powerlaw64_{s}"] = (64, directed(G))
    for name, (n, ei) in graphs.items():
        ei_t = torch.from_numpy(ei)
        E = ei.shape[1]
        w = torch.rand(E, generator=g) + 0.5
        rec[f"{name}/n"] = np.int64(n)
        rec[f"{name}/edge_index"] = ei
        rec[f"{name}/w"] = w.numpy()
        for d in (1, 3, 64):
            x = torch.randn(n, d, generator=g)
            rec[f"{name}/x{d}"] = x.numpy()
            for red in ("sum", "mean", "max"):
                rec[f"{name}/agg_{red}_d{d}"] = R.coo_aggregate(ei_t[1], ei_t[0], None, x, n, red).numpy()
                rec[f"{name}/aggw_{red}_d{d}"] = R.coo_aggregate(ei_t[1], ei_t[0], w, x, n, red).numpy()
        # TF flavour normalisation: edge_index[0] = row (TfgIDLayer.py:528-566)
        sa = R.gcn_norm_adj(R.SparseAdj(torch.stack([ei_t[1], ei_t[0]]), w, [n, n]))
        rec[f"{name}/tf_norm_index"] = sa.edge_index.numpy()
        rec[f"{name}/tf_norm_weight"] = sa.edge_weight.numpy()
        # PyG flavour (idconv.py:132-148)
        pei, pw = R.pyg_gcn_norm(ei_t, n, w)
        rec[f"{name}/pyg_norm_index"] = pei.numpy()
        rec[f"{name}/pyg_norm_weight"] = pw.numpy()
    np.savez_compressed(os.path.join(OUT, "aggregation.npz"), **rec)
    return len(rec)


def ego_batch(n=12, m=2, radius=2, seed=7, with_map=False):
    G = nx.powerlaw_cluster_graph(n, m, 0.3, seed=seed)
    H, ids, orig, ego_of = RL.ego_nets(G, radius, return_map=True)
    if with_map:
        return H.number_of_nodes(), directed(H), ids.numpy(), directed(G), orig.numpy(), ego_of.numpy()
    return H.number_of_nodes(), directed(H), ids.numpy(), directed(G)


def gen_ego():
    """transform.py:11-38 on small synthetic graphs: expanded edge list + id index"""
    rec = {}
    for k, (n, m, radius) in enumerate([(12, 2, 2), (16, 3, 3), (10, 2, 5), (40, 2, 1)]):
        N, ei, ids, base, orig, ego_of = ego_batch(n, m, radius, seed=11 + k, with_map=True)
        rec[f"g{k}/orig_node"] = orig
        rec[f"g{k}/ego_of_node"] = ego_of
        rec[f"g{k}/base_n"] = np.int64(n)
        rec[f"g{k}/radius"] = np.int64(radius)
        rec[f"g{k}/base_edge_index"] = base
        rec[f"g{k}/ego_n"] = np.int64(N)
        rec[f"g{k}/ego_edge_index"] = ei
        rec[f"g{k}/node_id_index"] = ids
    np.savez_compressed(os.path.join(OUT, "ego.npz"), **rec)
    return len(rec)


def _glorot(gen, *shape):
    stdv = (6.0 / (shape[-2] + shape[-1])) ** 0.5
    return ((torch.rand(*shape, generator=gen) * 2 - 1) * stdv).requires_grad_(True)


def _mlp_params(gen, d_in, d_out):
    return [_glorot(gen, d_in, d_out), (torch.randn(d_out, generator=gen) * 0.1).requires_grad_(True),
            _glorot(gen, d_out, d_out), (torch.randn(d_out, generator=gen) * 0.1).requires_grad_(True)]


def _mlp_fn(p):
    # Linear -> ReLU -> Linear with weights stored [in, out]
    return lambda h: torch.relu(h @ p[0] + p[1]) @ p[2] + p[3]


def gen_layers():
    """one forward + backward record per layer key and flavour (fixed seed weights)"""
    rec = {}
    gen = torch.Generator().manual_seed(424242)
    N, ei_np, ids_np, _ = ego_batch(12, 2, 2, seed=7)
    ei = torch.from_numpy(ei_np)              # [source; destination]; symmetric, so also valid as [row; col]
    ids = torch.from_numpy(ids_np)
    F_in, D = 8, 16
    x0 = torch.randn(N, F_in, generator=gen)
    dy = torch.randn(N, D, generator=gen)
    rec["n"], rec["edge_index"], rec["node_id_index"] = np.int64(N), ei_np, ids_np
    rec["x"], rec["dy"] = x0.numpy(), dy.numpy()

    def record(key, fn, params):
        x = x0.clone().requires_grad_(True)
        out = fn(x)
        grads = torch.autograd.grad(out, [x] + list(params.values()), grad_outputs=dy[:, :out.size(1)],
                                    allow_unused=True)
        rec[f"{key}/out"] = out.detach().numpy()
        rec[f"{key}/grad_x"] = grads[0].numpy()
        for (pn, p), g in zip(params.items(), grads[1:]):
            rec[f"{key}/param/{pn}"] = p.detach().numpy()
            rec[f"{key}/grad/{pn}"] = (torch.zeros_like(p) if g is None else g).numpy()
        # the same record evaluated in float64 (same fp32 inputs and weights, exact arithmetic for this purpose):
        # the tests hold the engine to 1e-5 of THIS, or to twice the fp32 oracle's own distance from it where the
        # GEMMs' fp32 rounding exceeds 1e-5
        saved = [p.data for p in params.values()]
        torch.set_default_dtype(torch.float64)
        try:
            for p in params.values():
                p.data = p.data.double()
            x = x0.double().requires_grad_(True)
            out = fn(x)
            grads = torch.autograd.grad(out, [x] + list(params.values()), grad_outputs=dy[:, :out.size(1)].double(),
                                        allow_unused=True)
            rec[f"{key}/out64"] = out.detach().numpy()
            rec[f"{key}/grad_x64"] = grads[0].numpy()
            for (pn, p), g in zip(params.items(), grads[1:]):
                rec[f"{key}/grad64/{pn}"] = (torch.zeros_like(p) if g is None else g).numpy()
        finally:
            torch.set_default_dtype(torch.float32)
            for p, d in zip(params.values(), saved):
                p.data = d

    b = lambda: (torch.randn(D, generator=gen) * 0.1).requires_grad_(True)
    # ---- PyG family ----
    W, Wid, bias = _glorot(gen, F_in, D), _glorot(gen, F_in, D), b()
    record("gcnidconv", lambda x: RL.gcnid_conv(x, ei, ids, W, Wid, bias),
           {"weight": W, "weight_id": Wid, "bias": bias})
    for agg in ("add", "mean", "max"):
        W, Wid, bias = _glorot(gen, F_in, D), _glorot(gen, F_in, D), b()
        record(f"idconv_{agg}", lambda x, W=W, Wid=Wid, bias=bias, agg=agg:
               RL.generalid_conv(x, ei, ids, W, Wid, bias, agg=agg), {"weight": W, "weight_id": Wid, "bias": bias})
        W, Ws, bias = _glorot(gen, F_in, D), _glorot(gen, F_in, D), b()
        record(f"generalconv_{agg}", lambda x, W=W, Ws=Ws, bias=bias, agg=agg:
               RL.general_conv(x, ei, W, Ws, bias, agg=agg), {"weight": W, "weight_self": Ws, "bias": bias})
    W, Wid, bias = _glorot(gen, 2 * F_in, D), _glorot(gen, 2 * F_in, D), b()
    record("sageidconv", lambda x: RL.sageid_conv(x, ei, ids, W, Wid, bias, concat=True),
           {"weight": W, "weight_id": Wid, "bias": bias})
    W, Wid, att, bias = _glorot(gen, F_in, D), _glorot(gen, F_in, D), _glorot(gen, 1, 1, 2 * D), b()
    record("gatidconv", lambda x: RL.gatid_conv(x, ei, ids, W, Wid, att, bias),
           {"weight": W, "weight_id": Wid, "att": att, "bias": bias})
    p, pid = _mlp_params(gen, F_in, D), _mlp_params(gen, F_in, D)
    record("ginidconv", lambda x: RL.ginid_conv(x, ei, ids, _mlp_fn(p), _mlp_fn(pid)),
           {**{f"nn.{i}": t for i, t in enumerate(p)}, **{f"nn_id.{i}": t for i, t in enumerate(pid)}})
    W, bias = _glorot(gen, F_in, D), b()
    record("gcnconv", lambda x: RL.pyg_gcn_conv(x, ei, W, bias), {"weight": W, "bias": bias})
    Wl, bl, Wr = _glorot(gen, D, F_in), b(), _glorot(gen, D, F_in)
    record("sageconv", lambda x: RL.pyg_sage_conv(x, ei, Wl, bl, Wr),
           {"lin_l.weight": Wl, "lin_l.bias": bl, "lin_r.weight": Wr})
    W, ai, aj, bias = _glorot(gen, F_in, D), _glorot(gen, 1, D), _glorot(gen, 1, D), b()
    record("gatconv", lambda x: RL.pyg_gat_conv(x, ei, W, ai.view(-1), aj.view(-1), bias),
           {"weight": W, "att_dst": ai, "att_src": aj, "bias": bias})
    p = _mlp_params(gen, F_in, D)
    record("ginconv", lambda x: RL.pyg_gin_conv(x, ei, _mlp_fn(p)), {f"nn.{i}": t for i, t in enumerate(p)})
    # ---- TF family (edge_index[0] = row) ----
    W, Wid, bias = _glorot(gen, F_in, D), _glorot(gen, F_in, D), b()
    record("tf_idgcn", lambda x: RL.gcn_id(x, ei, ids, None, W, Wid, bias, "relu"),
           {"kernel": W, "kernel_id": Wid, "bias": bias})
    W, bias = _glorot(gen, F_in, D), b()
    record("tf_gcn", lambda x: RL.tfg_gcn(x, ei, None, W, bias, "relu"), {"kernel": W, "bias": bias})
    Ws, Wi, Wn, bias = _glorot(gen, F_in, D // 2), _glorot(gen, F_in, D // 2), _glorot(gen, F_in, D // 2), b()
    record("tf_idsage", lambda x: RL.idsage(x, ei, ids, None, Ws, Wi, Wn, bias, "relu"),
           {"self_kernel": Ws, "id_kernel": Wi, "neighbor_kernel": Wn, "bias": bias})
    Ws, Wn, bias = _glorot(gen, F_in, D // 2), _glorot(gen, F_in, D // 2), b()
    record("tf_sage", lambda x: RL.tfg_mean_graph_sage(x, ei, None, Ws, Wn, bias, "relu"),
           {"self_kernel": Ws, "neighbor_kernel": Wn, "bias": bias})
    p, pid = _mlp_params(gen, F_in, D), _mlp_params(gen, F_in, D)
    record("tf_idgin", lambda x: RL.idgin(x, ei, ids, _mlp_fn(p), _mlp_fn(pid)),
           {**{f"mlp.{i}": t for i, t in enumerate(p)}, **{f"mlp_id.{i}": t for i, t in enumerate(pid)}})
    p = _mlp_params(gen, F_in, D)
    record("tf_gin", lambda x: RL.tfg_gin(x, ei, _mlp_fn(p)), {f"mlp.{i}": t for i, t in enumerate(p)})
    for heads in (1, 4):
        Wq, bq, Wk, bk = _glorot(gen, F_in, D), b(), _glorot(gen, F_in, D), b()
        W, Wid, bias = _glorot(gen, F_in, D), _glorot(gen, F_in, D), b()
        record(f"tf_idgat_h{heads}", lambda x, heads=heads, Wq=Wq, bq=bq, Wk=Wk, bk=bk, W=W, Wid=Wid, bias=bias:
               RL.gat_id(x, ei, ids, Wq, bq, Wk, bk, W, Wid, bias, "relu", num_heads=heads),
               {"query_kernel": Wq, "query_bias": bq, "key_kernel": Wk, "key_bias": bk, "kernel": W,
                "kernel_id": Wid, "bias": bias})
    Wq, bq, Wk, bk, W, bias = _glorot(gen, F_in, D), b(), _glorot(gen, F_in, D), b(), _glorot(gen, F_in, D), b()
    record("tf_gat", lambda x: RL.tfg_gat(x, ei, Wq, bq, Wk, bk, W, bias, "relu"),
           {"query_kernel": Wq, "query_bias": bq, "key_kernel": Wk, "key_bias": bk, "kernel": W, "bias": bias})
    np.savez_compressed(os.path.join(OUT, "layers.npz"), **rec)
    return len(rec)


if __name__ == "__main__":
    torch.set_num_threads(1)
    print("aggregation.npz", gen_aggregation())
    print("ego.npz", gen_ego())
    print("layers.npz", gen_layers())
    for f in ("aggregation.npz", "ego.npz", "layers.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")

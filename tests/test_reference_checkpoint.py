"""The layer / model boundary pinned by the reference's OWN artefacts.

The reference holds two trained `gcnidconv` checkpoints (run/results/node/1/ckpt/999.ckpt: TU_BZR graph task, F = 53;
run/results/node-Copy1/1/ckpt/999.ckpt: TU_PROTEINS node task, F = 3; both 1 pre-MP linear + 3 x GCNIDConv, d = 128,
BatchNorm, l2norm, transform 'ego'; written by graphgym/checkpoint.py:43-53).  tests/golden/make_ref_ckpt.py turned them
into arrays with a weights-only loader (tests/golden/ref_ckpt_*.npz: tensors, key order, the config entries the model
assembly reads).  They pin the state-dict contract of SURVEY §8 row B:

  * not gpu: harness.GNN built from the checkpoint's own config has exactly the reference's keys, in its order, with its
    shapes; load_state_dict(strict=True) succeeds; the keys survive accelerate() and a save / load round trip;
  * gpu: the loaded model's eval forward on a synthetic ego batch (the datasets are downloads, not in the reference) equals
    the float64 oracle evaluated from the same trained tensors — through the plain path, through accelerate()'s folded
    path, and in train mode after accelerate (engine BatchNorm) against torch's own modules.

What this does NOT pin: numerics of the third-party kernels (the oracle stays a restatement).
"""
import ast
import os

import numpy as np
import pytest
import torch

from _tol import assert_close_rows

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPTS = {"ref_ckpt_node": dict(f_in=53, classes=1, task="graph"),
         "ref_ckpt_node_copy1": dict(f_in=3, classes=6, task="node")}


def _load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    keys = [str(k) for k in z["__keys__"]]
    state = {k: torch.from_numpy(np.array(z[k])) for k in keys}
    conf = {}
    for item in z["__cfg__"]:
        k, v = str(item).split("=", 1)
        conf[k] = ast.literal_eval(v)
    return keys, state, conf


class _cfg_from:
    """apply the checkpoint's config.yaml entries to the engine's cfg for the duration of a test"""

    def __init__(self, conf):
        self.conf = conf

    def __enter__(self):
        from graphgym_amd.config import cfg
        self.saved = []
        for dotted, v in self.conf.items():
            sec, key = dotted.split(".")
            node = getattr(cfg, sec)
            self.saved.append((node, key, getattr(node, key, None), hasattr(node, key)))
            setattr(node, key, v)
        return cfg

    def __exit__(self, *exc):
        for node, key, old, had in reversed(self.saved):
            if had:
                setattr(node, key, old)
            else:
                delattr(node, key)


@pytest.mark.parametrize("name", sorted(CKPTS))
def test_reference_state_dict_loads_strict_and_round_trips(name):
    import graphgym_amd.graphgym_plugin as plugin
    from graphgym_amd import harness as H
    keys, state, conf = _load(name)
    info = CKPTS[name]
    assert conf["gnn.layer_type"] == "gcnidconv" and conf["dataset.task"] == info["task"]
    with _cfg_from(conf):
        model = H.GNN(info["f_in"], info["classes"])
        own = model.state_dict()
        assert list(own.keys()) == keys                                       # same names, same order
        for k in keys:
            assert tuple(own[k].shape) == tuple(state[k].shape) and own[k].dtype == state[k].dtype, k
        res = model.load_state_dict(state, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        for k, v in model.state_dict().items():
            assert torch.equal(v, state[k]), k
        # the conv layers carry no bias under BatchNorm (layer.py:24-25), the head does (head.py:24-25)
        assert model.mp.layer0.layer.model.bias is None and model.post_mp.layer_post_mp.model[0].model.bias is not None
        # accelerate() swaps modules in place without touching the contract
        assert plugin.accelerate(model) == conf["gnn.layers_pre_mp"] + conf["gnn.layers_mp"]
        assert list(model.state_dict().keys()) == keys
        model2 = H.GNN(info["f_in"], info["classes"])
        model2.load_state_dict(model.state_dict(), strict=True)
        for k, v in model2.state_dict().items():
            assert torch.equal(v, state[k]), k


def _ego_batch(dev, f_in, seed, n_graphs=12):
    """a synthetic batch of the checkpoints' kind: small molecule-sized graphs, every node expanded into its radius-3
    ego net by the GPU batcher (transform.py:11-38; radius = layers_mp), disjoint union, DeepSNAP's batch fields"""
    import networkx as nx
    import graphgym_amd as ga
    from graphgym_amd.ego import ego_batch
    from graphgym_amd.harness import Batch
    rng = np.random.RandomState(seed)
    graphs = [nx.connected_watts_strogatz_graph(int(rng.randint(12, 40)), 4, 0.3, seed=int(rng.randint(1 << 30)))
              for _ in range(n_graphs)]
    U = nx.disjoint_union_all(graphs)
    n = U.number_of_nodes()
    e = np.array(list(U.edges()), dtype=np.int64)
    ei = torch.from_numpy(np.ascontiguousarray(np.concatenate([e, e[:, ::-1]], 0).T))
    graph_of = torch.from_numpy(np.repeat(np.arange(n_graphs), [g.number_of_nodes() for g in graphs]))
    base = ga.CSRGraph.from_edge_index(ei.to(dev), n, validate=True)
    eei, orig, ids, ego_of = ego_batch(base, torch.arange(n, device=dev), 3)
    gen = torch.Generator().manual_seed(seed)
    feat = torch.randn(n, f_in, generator=gen)
    b = Batch(node_feature=feat.to(dev)[orig], edge_index=eei, node_id_index=ids,
              batch=graph_of.to(dev)[orig], node_label_index=torch.arange(n, device=dev),
              node_label=torch.zeros(n, dtype=torch.int64, device=dev),
              graph_label=torch.zeros(n_graphs, dtype=torch.int64, device=dev))
    return b, n_graphs


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CKPTS))
def test_trained_reference_weights_forward_matches_oracle(dev, name):
    import graphgym_amd.graphgym_plugin as plugin
    from graphgym_amd import harness as H
    from oracle import ref_layers as RL
    keys, state, conf = _load(name)
    info = CKPTS[name]
    with _cfg_from(conf):
        model = H.GNN(info["f_in"], info["classes"])
        model.load_state_dict(state, strict=True)
        model = model.to(dev).eval()
        batch, n_graphs = _ego_batch(dev, info["f_in"], seed=11)
        x0 = batch.node_feature.clone()
        kw = dict(layers_pre_mp=conf["gnn.layers_pre_mp"], layers_mp=conf["gnn.layers_mp"], bn_eps=conf["bn.eps"],
                  l2norm=conf["gnn.l2norm"], task=info["task"], ego=True,
                  node_label_index=batch.node_label_index.cpu(), batch=batch.batch.cpu(), num_graphs=n_graphs)
        ref64 = RL.graphgym_gnn_eval(state, x0.cpu().double(), batch.edge_index.cpu(), batch.node_id_index.cpu(), **kw)
        ref32 = RL.graphgym_gnn_eval(state, x0.cpu(), batch.edge_index.cpu(), batch.node_id_index.cpu(), **kw)
        assert bool(torch.isfinite(ref64).all()) and float(ref64.abs().max()) > 0
        with torch.no_grad():
            pred, _ = model(batch)
        assert_close_rows(pred, ref64, 1e-5, ref32=ref32, what=f"{name}: eval forward, plain path", deep=True)   # whole model: 1 linear + 3 conv + BN + l2norm + head
        # accelerate(): BatchNorm(eval) + ReLU folded into the aggregation's row flush for the gcnidconv layers
        plugin.accelerate(model)
        batch.node_feature = x0.clone()
        with torch.no_grad():
            pred2, _ = model(batch)
        assert_close_rows(pred2, ref64, 1e-5, ref32=ref32, what=f"{name}: eval forward, folded path", deep=True)   # whole model: 1 linear + 3 conv + BN + l2norm + head
        # train mode after accelerate (engine BatchNorm on batch statistics) against torch's own modules in float64
        model.train()
        batch.node_feature = x0.clone()
        pred3, _ = model(batch)
        def train_ref(dtype):
            h = x0.cpu().to(dtype)
            st = {k: v.to(dtype) if v.is_floating_point() else v for k, v in state.items()}
            bn = lambda t, pre: torch.relu(torch.nn.functional.batch_norm(
                t, None, None, st[pre + ".weight"], st[pre + ".bias"], True, 0.1, conf["bn.eps"]))
            for i in range(conf["gnn.layers_pre_mp"]):
                h = bn(h @ st[f"pre_mp.Layer_{i}.layer.model.weight"].t(), f"pre_mp.Layer_{i}.post_layer.0")
            for i in range(conf["gnn.layers_mp"]):
                h = bn(RL.gcnid_conv(h, batch.edge_index.cpu(), batch.node_id_index.cpu(),
                                     st[f"mp.layer{i}.layer.model.weight"], st[f"mp.layer{i}.layer.model.weight_id"],
                                     bias=None), f"mp.layer{i}.post_layer.0")
            h = torch.nn.functional.normalize(h, p=2, dim=-1)
            Wp, bp = st["post_mp.layer_post_mp.model.0.model.weight"], st["post_mp.layer_post_mp.model.0.model.bias"]
            if info["task"] == "node":
                return (h @ Wp.t() + bp)[batch.node_label_index.cpu()]
            ids = batch.node_id_index.cpu()
            pooled = torch.zeros(n_graphs, h.size(1), dtype=h.dtype).index_add_(0, batch.batch.cpu()[ids], h[ids])
            return pooled @ Wp.t() + bp
        ref_tr, ref_tr32 = train_ref(torch.float64), train_ref(torch.float32)
        assert_close_rows(pred3, ref_tr, 1e-5, ref32=ref_tr32, what=f"{name}: train-mode forward", deep=True)   # whole model: 1 linear + 3 conv + BN + l2norm + head

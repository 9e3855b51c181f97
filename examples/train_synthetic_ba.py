#!/usr/bin/env python3
"""End-to-end sanity run on the engine: the reference's synthetic node-classification task
(predict the binned clustering coefficient, config/idgcn_tf/idgcn_node_ba.yaml) with the
TF-path GCN vs ID-GCN models of main_zd.py, on synthetic BA(64, 2) graphs of the shape of the
bundled datasets/ba.pkl (100 graphs x 64 nodes; the pickle itself is not loaded).

Reference sanity band (README.md:104-118, results/val/final/Tfg-{gcn,idgcn}_ba_avg_acc.txt):
GCN 0.695, ID-GCN (Full) 0.964 on an RTX 2080 Ti after 1000 epochs.

    python examples/train_synthetic_ba.py --epochs 300
"""
import argparse
import json
import os
import sys
import time

import networkx as nx
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphgym_amd as ga  # noqa: E402
from graphgym_amd import harness as H  # noqa: E402
from graphgym_amd.ego import ego_batch  # noqa: E402


def make_dataset(n_graphs=100, n=64, m=2, seed=0, label_dims=10):
    graphs = [nx.barabasi_albert_graph(n, m, seed=seed + i) for i in range(n_graphs)]
    cc = [np.array([nx.clustering(G, v) for v in range(n)]) for G in graphs]
    # balanced binning over the whole dataset (feature_augment.py:218-231), label = digitize - 1 (:141)
    allv = np.sort(np.concatenate(cc))
    bins = np.unique(allv[np.linspace(0, len(allv), num=label_dims, endpoint=False).astype(int)])
    labels = [np.digitize(c, bins) - 1 for c in cc]
    return graphs, labels, len(bins)


def union(graphs, labels, dev):
    """one batch = disjoint union of graphs (loader.py:247-251)"""
    parts, off = [], 0
    for G in graphs:
        e = np.array(list(G.edges()), dtype=np.int64) + off
        parts.append(np.concatenate([e, e[:, ::-1]]))
        off += G.number_of_nodes()
    ei = torch.from_numpy(np.concatenate(parts).T.copy()).to(dev)
    y = torch.from_numpy(np.concatenate(labels)).long().to(dev)
    return ei, y, off


def run(kind, epochs, dev, seed=0, radius=3, d=128, hipgraph=False):
    graphs, labels, n_cls = make_dataset(seed=0)
    torch.manual_seed(seed)
    split = int(0.8 * len(graphs))
    sets = {}
    for name, sl in (("train", slice(0, split)), ("val", slice(split, None))):
        ei, y, n = union(graphs[sl], labels[sl], dev)
        if kind.startswith("id"):   # ID-GNN Full: ego-net expansion, transform.py:11-38 with radius = layers_mp
            base = ga.CSRGraph.from_edge_index(ei, n)
            ei2, orig, ids, _ = ego_batch(base, torch.arange(n, device=dev), radius)
            sets[name] = dict(ei=ei2, x=torch.ones(orig.numel(), 1, device=dev), ids=ids, y=y,
                              label_index=torch.arange(n, device=dev), holder=H.Batch())
        else:
            sets[name] = dict(ei=ei, x=torch.ones(n, 1, device=dev), ids=None, y=y,
                              label_index=torch.arange(n, device=dev), holder=H.Batch())
    model = H.TfgNodeModel(kind, 1, d, n_cls).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=hipgraph)

    def logits(s):
        inputs = [s["x"], s["ei"]] + ([s["ids"]] if s["ids"] is not None else [])
        return model(inputs, holder=s["holder"])

    best = 0.0
    tr = sets["train"]
    train_loss = lambda: H.tfg_loss(logits(tr), tr["label_index"], tr["y"], model.kernel_parameters())
    model.train()
    graphed = H.GraphedTrainStep(model, opt, train_loss) if hipgraph else None   # full-batch: same shapes every epoch
    torch.cuda.synchronize()
    t0 = time.time()
    for ep in range(epochs):
        model.train()
        if graphed is not None:
            graphed()
        else:
            H.train_step(model, opt, train_loss)
        if ep % 10 == 0 or ep == epochs - 1:
            model.eval()
            with torch.no_grad():
                v = sets["val"]
                acc = float((logits(v)[v["label_index"]].argmax(1) == v["y"]).float().mean())
            best = max(best, acc)
    torch.cuda.synchronize()
    return {"model": kind, "epochs": epochs, "hipgraph": hipgraph, "best_val_acc": best, "seconds": time.time() - t0,
            "classes": n_cls, "train_nodes": int(sets["train"]["x"].size(0))}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--hipgraph", action="store_true", help="capture the training step into a HIP graph")
    ap.add_argument("--seeds", type=int, default=1, help="runs per model (weight initialisation seeds 0..seeds-1)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    # the reference's ScaleFree column (README.md:104-118; TF path, RTX 2080 Ti, 1000 epochs, its own dataset and class
    # count): a sanity band, not a parity target
    REF = {"gcn": 0.695, "sage": 0.470, "gat": 0.470, "gin": 0.639, "idgcn": 0.964, "idsage": 0.579, "idgat": 0.987,
           "idgin": 0.660}
    for kind in ("gcn", "idgcn", "sage", "idsage", "gat", "idgat", "gin", "idgin"):
        accs = []
        for seed in range(args.seeds):
            r = run(kind, args.epochs, dev, seed=seed, hipgraph=args.hipgraph)
            r["seed"] = seed
            accs.append(r["best_val_acc"])
            print(json.dumps(r), flush=True)
        if args.seeds > 1:
            mean = float(np.mean(accs))
            print(json.dumps({"model": kind, "summary": True, "seeds": args.seeds, "mean_best_val_acc": mean,
                              "min": float(min(accs)), "max": float(max(accs)), "reference_scalefree": REF[kind],
                              "delta_vs_reference": mean - REF[kind], "outside_band_0.08": abs(mean - REF[kind]) > 0.08}),
                  flush=True)

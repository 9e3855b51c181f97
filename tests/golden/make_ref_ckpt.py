"""Fixture generator (build container only; /root/reference does not travel to the GPU box).

The reference holds two trained checkpoints written by graphgym/checkpoint.py:43-53:

    run/results/node/1/ckpt/999.ckpt        gcnidconv, TU_BZR graph task, transform: ego   (config.yaml beside it)
    run/results/node-Copy1/1/ckpt/999.ckpt  gcnidconv, TU_PROTEINS node task, transform: ego

They are the only reference-held data that pins anything on this path: the STATE-DICT CONTRACT of the layer /
model boundary (module names, parameter names, shapes, which layers carry a bias, BatchNorm buffers).  Loaded with
torch.load(..., weights_only=True) — nothing from the file is executed.  Written out as plain arrays:

    tests/golden/ref_ckpt_node.npz, ref_ckpt_node_copy1.npz
        __keys__   the model_state keys in their stored order ('<U..' array)
        __cfg__    the config.yaml entries the model assembly reads, as 'section.key=value' strings
        <key>      one array per model_state tensor

This is data (tensors + key names + config scalars), not reference source.
"""
import os
import sys

import numpy as np
import torch
import yaml

REF = "/root/reference/run/results"
HERE = os.path.dirname(os.path.abspath(__file__))
RUNS = {"ref_ckpt_node": "node/1", "ref_ckpt_node_copy1": "node-Copy1/1"}
CFG_KEYS = [("gnn", k) for k in ("layer_type", "layers_pre_mp", "layers_mp", "layers_post_mp", "dim_inner", "batchnorm",
                                "act", "dropout", "agg", "normalize_adj", "l2norm", "stage_type", "self_msg")] + \
           [("bn", "eps"), ("bn", "mom"), ("dataset", "task"), ("dataset", "transform"), ("dataset", "name"),
            ("dataset", "augment_feature"), ("dataset", "augment_feature_dims"), ("model", "graph_pooling"),
            ("model", "loss_fun")]


def main():
    for out, run in RUNS.items():
        ck = torch.load(os.path.join(REF, run, "ckpt", "999.ckpt"), weights_only=True, map_location="cpu")
        with open(os.path.join(REF, run, "config.yaml")) as f:
            cfg = yaml.safe_load(f)
        state = ck["model_state"]
        arrays = {k: v.numpy() for k, v in state.items()}
        arrays["__keys__"] = np.array(list(state.keys()))
        arrays["__cfg__"] = np.array([f"{s}.{k}={cfg[s][k]!r}" for s, k in CFG_KEYS])
        arrays["__epoch__"] = np.array(int(ck["epoch"]))
        path = os.path.join(HERE, out + ".npz")
        np.savez_compressed(path, **arrays)
        print(path, len(state), "tensors,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("the reference checkout is not present: fixtures are generated in the build container only")
    main()

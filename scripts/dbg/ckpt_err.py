"""Where does the eval forward of the reference-checkpoint model lose accuracy against the float64 oracle?  Prints the
per-row relative error of the engine and of the float32 oracle for the plain path, layer by layer."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_reference_checkpoint as T
import graphgym_amd.graphgym_plugin  # noqa: F401 (registers the layer keys)
from graphgym_amd import harness as H
from oracle import ref_layers as RL
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "ref_ckpt_node_copy1"
keys, state, conf = T._load(name)
info = T.CKPTS[name]
with T._cfg_from(conf):
    for order in ("auto", "transform_first"):
        model = H.GNN(info["f_in"], info["classes"])
        model.load_state_dict(state, strict=True)
        model = model.to(dev).eval()
        for m in model.modules():
            if hasattr(m, "order"):
                m.order = order
        batch, n_graphs = T._ego_batch(dev, info["f_in"], seed=11)
        x0 = batch.node_feature.clone()
        feats = {}
        hooks = [mod.register_forward_hook(lambda m, i, o, k=k: feats.__setitem__(k, o.node_feature.detach().cpu().double().clone())
                                           if hasattr(o, "node_feature") else None)
                 for k, mod in [("pre", model.pre_mp), ("l0", model.mp.layer0), ("l1", model.mp.layer1), ("mp", model.mp)]]
        with torch.no_grad():
            pred, _ = model(batch)
        ei, ids = batch.edge_index.cpu(), batch.node_id_index.cpu()

        def oracle(dtype):
            t = lambda k: state[k].to(dtype)
            out = {}
            h = x0.cpu().to(dtype)
            bn = lambda h, p: torch.relu((h - t(p + ".running_mean")) / torch.sqrt(t(p + ".running_var") + conf["bn.eps"]) * t(p + ".weight") + t(p + ".bias"))
            h = bn(h @ t("pre_mp.Layer_0.layer.model.weight").t(), "pre_mp.Layer_0.post_layer.0"); out["pre"] = h
            for i in range(3):
                h = bn(RL.gcnid_conv(h, ei, ids, t(f"mp.layer{i}.layer.model.weight"), t(f"mp.layer{i}.layer.model.weight_id"), None),
                       f"mp.layer{i}.post_layer.0")
                out[f"l{i}"] = h
            out["mp"] = torch.nn.functional.normalize(h, p=2, dim=-1)
            return out
        torch.set_default_dtype(torch.float64); o64 = oracle(torch.float64); torch.set_default_dtype(torch.float32)
        o32 = oracle(torch.float32)
        print("order", order)
        for k in ("pre", "l0", "l1", "mp"):
            r = o64[k]; s = r.abs().amax(1).clamp(min=1e-300)
            e = (feats[k] - r).abs().amax(1) / s
            e32 = (o32[k].double() - r).abs().amax(1) / s
            print(f"  {k:4s} engine rel err: median {float(e.median()):.2e} p99 {float(e.quantile(0.99)):.2e} max {float(e.max()):.2e} | "
                  f"oracle32: median {float(e32.median()):.2e} p99 {float(e32.quantile(0.99)):.2e} max {float(e32.max()):.2e} | rows {r.size(0)}")
        for hk in hooks: hk.remove()

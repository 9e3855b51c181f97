"""Which module of the reference-checkpoint model turns a 1x error into a 20x one on a single output row?

VERDICT r3 #3: on `ref_ckpt_node_copy1` (gcnidconv, TU_PROTEINS node task: 1 pre-MP linear + 3 x GCNIDConv + BN +
l2norm + node head, graphgym/models/gnn.py:123-168, layer.py:16-47) one of 291 output rows of the PLAIN eval path sat at
1.4e-5 of its own magnitude, 20 x the float32 oracle's own error on that row.  This script evaluates the model three
ways on the same inputs — the engine, the oracle in float32, the oracle in float64 — captures the features behind EVERY
module (pre-MP linear, its BN+ReLU, each conv, each BN+ReLU, l2norm, head) and prints, for the worst output row and for
any rows given on the command line, that row's error after every module:

    err / max|row|            engine and float32 oracle, against float64
    err / sum|terms| (head)   the head's product h @ Wp^T + b measured against the row's sum of ABSOLUTE terms

    python scripts/ckpt_row_trace.py [ckpt name] [row ...] > profiles/r04_ckpt_row_trace.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_reference_checkpoint as T  # noqa: E402
import graphgym_amd.graphgym_plugin  # noqa: E402,F401  (registers the layer keys)
from graphgym_amd import harness as H  # noqa: E402
from oracle import ref_layers as RL  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    name = sys.argv[1] if len(sys.argv) > 1 else "ref_ckpt_node_copy1"
    extra_rows = [int(a) for a in sys.argv[2:]]
    keys, state, conf = T._load(name)
    info = T.CKPTS[name]
    n_pre, n_mp = conf["gnn.layers_pre_mp"], conf["gnn.layers_mp"]
    with T._cfg_from(conf):
        model = H.GNN(info["f_in"], info["classes"])
        model.load_state_dict(state, strict=True)
        model = model.to(dev).eval()
        batch, n_graphs = T._ego_batch(dev, info["f_in"], seed=11)
        x0 = batch.node_feature.clone()
        feats = {}

        def grab(k):
            def hook(m, i, o):
                t = o.node_feature if hasattr(o, "node_feature") else o
                if isinstance(t, tuple):
                    t = t[0]
                feats[k] = t.detach().cpu().double().clone()
            return hook

        hooks = []
        for i in range(n_pre):
            L = getattr(model.pre_mp, f"Layer_{i}")
            hooks += [L.layer.register_forward_hook(grab(f"pre{i}.linear")), L.register_forward_hook(grab(f"pre{i}.bn_relu"))]
        for i in range(n_mp):
            L = getattr(model.mp, f"layer{i}")
            hooks += [L.layer.register_forward_hook(grab(f"mp{i}.conv")), L.register_forward_hook(grab(f"mp{i}.bn_relu"))]
        hooks.append(model.mp.register_forward_hook(grab("l2norm")))
        hooks.append(model.post_mp.layer_post_mp.register_forward_hook(grab("head")))
        with torch.no_grad():
            pred, _ = model(batch)
        for h in hooks:
            h.remove()
        ei, ids = batch.edge_index.cpu(), batch.node_id_index.cpu()
        lab = batch.node_label_index.cpu()

        def oracle(dtype):
            t = lambda k: state[k].to(dtype)
            out = {}
            h = x0.cpu().to(dtype)

            def bn(h, p):
                h = (h - t(p + ".running_mean")) / torch.sqrt(t(p + ".running_var") + conf["bn.eps"])
                return torch.relu(h * t(p + ".weight") + t(p + ".bias"))
            for i in range(n_pre):
                h = h @ t(f"pre_mp.Layer_{i}.layer.model.weight").t(); out[f"pre{i}.linear"] = h
                h = bn(h, f"pre_mp.Layer_{i}.post_layer.0"); out[f"pre{i}.bn_relu"] = h
            for i in range(n_mp):
                h = RL.gcnid_conv(h, ei, ids, t(f"mp.layer{i}.layer.model.weight"), t(f"mp.layer{i}.layer.model.weight_id"), None)
                out[f"mp{i}.conv"] = h
                h = bn(h, f"mp.layer{i}.post_layer.0"); out[f"mp{i}.bn_relu"] = h
            h = torch.nn.functional.normalize(h, p=2, dim=-1); out["l2norm"] = h
            Wp, bp = t("post_mp.layer_post_mp.model.0.model.weight"), t("post_mp.layer_post_mp.model.0.model.bias")
            out["head"] = h @ Wp.t() + bp
            out["head.mag"] = h.abs() @ Wp.abs().t() + bp.abs()
            return out

        torch.set_default_dtype(torch.float64)
        o64 = oracle(torch.float64)
        torch.set_default_dtype(torch.float32)
        o32 = oracle(torch.float32)

        # the test compares pred (rows node_label_index of the head) — here node_label_index = arange(centres)
        if info["task"] == "node":
            r = o64["head"][lab]
            e = (pred.cpu().double() - r).abs().amax(1)
            s = r.abs().amax(1).clamp(min=1e-300)
            e32 = (o32["head"][lab].double() - r).abs().amax(1)
            order = torch.argsort(e / s, descending=True)
            rows = [int(lab[i]) for i in order[:3]] + extra_rows
            print(f"# {name}: {r.size(0)} output rows; engine rel err median {float((e / s).median()):.2e} "
                  f"p99 {float((e / s).quantile(0.99)):.2e} max {float((e / s).max()):.2e} | float32 oracle: median "
                  f"{float((e32 / s).median()):.2e} p99 {float((e32 / s).quantile(0.99)):.2e} max {float((e32 / s).max()):.2e}")
            print(f"# rows over 1e-5 of their own magnitude: engine {int((e / s > 1e-5).sum())}, float32 oracle "
                  f"{int((e32 / s > 1e-5).sum())}")
        else:
            rows = extra_rows or [0]
        stages = [k for k in o64 if not k.endswith(".mag")]
        for row in rows:
            print(f"\n## node {row}")
            print(f"{'module':14s} {'max|row|':>10s} {'engine err/|row|':>17s} {'f32 oracle err/|row|':>21s} {'ratio':>7s}")
            for k in stages:
                if k not in feats:
                    continue
                ref = o64[k][row]
                sc = float(ref.abs().max())
                ee = float((feats[k][row] - ref).abs().max())
                e3 = float((o32[k][row].double() - ref).abs().max())
                line = f"{k:14s} {sc:10.3e} {ee / max(sc, 1e-300):17.3e} {e3 / max(sc, 1e-300):21.3e} {ee / max(e3, 1e-300):7.1f}"
                if k == "head":
                    mg = float(o64["head.mag"][row].max())
                    line += (f"   | sum|terms| {mg:.3e}: engine err / sum|terms| {ee / mg:.2e}, float32 oracle {e3 / mg:.2e}; "
                             f"cancellation sum|terms| / max|row| = {mg / max(sc, 1e-300):.1f}")
                print(line)
        # population view per module: how the two float32 evaluations compare over ALL rows
        print("\n## all rows, per module: median / p99 / max of err/|row| (engine | float32 oracle)")
        for k in stages:
            if k not in feats:
                continue
            ref = o64[k]
            sc = ref.abs().amax(1).clamp(min=1e-300)
            ee = (feats[k] - ref).abs().amax(1) / sc
            e3 = (o32[k].double() - ref).abs().amax(1) / sc
            print(f"{k:14s} engine {float(ee.median()):.2e} / {float(ee.quantile(0.99)):.2e} / {float(ee.max()):.2e} | "
                  f"oracle32 {float(e3.median()):.2e} / {float(e3.quantile(0.99)):.2e} / {float(e3.max()):.2e}")
        # BatchNorm(eval) is an affine map per column, x -> (x - mean) * gain + beta with gain = gamma / sqrt(var + eps):
        # absolute errors of its input are multiplied by the column's gain while the row's magnitude may shrink
        print("\n## BatchNorm(eval) column gains |gamma| / sqrt(running_var + eps): median / max, and how much the row "
              "magnitudes change through the module (median over rows of max|out row| / max|in row|)")
        pairs = [(f"pre_mp.Layer_{i}.post_layer.0", f"pre{i}.linear", f"pre{i}.bn_relu") for i in range(n_pre)] + \
                [(f"mp.layer{i}.post_layer.0", f"mp{i}.conv", f"mp{i}.bn_relu") for i in range(n_mp)]
        for pre, kin, kout in pairs:
            gain = state[pre + ".weight"].double().abs() / torch.sqrt(state[pre + ".running_var"].double() + conf["bn.eps"])
            ratio = o64[kout].abs().amax(1) / o64[kin].abs().amax(1).clamp(min=1e-300)
            print(f"{kout:14s} gain {float(gain.median()):.2f} / {float(gain.max()):.2f}   row magnitude out / in: median "
                  f"{float(ratio.median()):.3f}, min {float(ratio.min()):.3f}")
        if info["task"] == "node":
            mg = o64["head.mag"][lab].amax(1)
            print(f"\n## head rows against their sum of absolute terms: engine max {float((e / mg).max()):.2e}, float32 oracle "
                  f"max {float((e32 / mg).max()):.2e}; cancellation factor sum|terms| / max|row|: median "
                  f"{float((mg / s).median()):.1f}, max {float((mg / s).max()):.1f}")


if __name__ == "__main__":
    main()

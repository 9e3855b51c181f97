"""The fresh-batch pipeline (graphgym_amd/pipeline.py): a batch built one step ahead on a side stream and a worker thread —
expansion writing the CSR itself, the graph flagged symmetric, the identity-branch shortcut, everything warmed — must
give the step EXACTLY what the plain path gives (ego_batch, then the model building its graph structures lazily from
edge_index): same logits, same gradients, bit for bit, step after step, whatever the caller does with its references;
and the step must build nothing itself."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _setup(dev, kind, n=60_000, f_in=64, d=64):
    import graphgym_amd as ga
    from graphgym_amd import graphgen, harness as H
    base = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n, 4, seed=2, device=dev), n)
    feats = torch.rand(n, f_in, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 2 - 1
    torch.manual_seed(3)
    model = H.TfgNodeModel(kind, f_in, d, 5).to(dev)
    labels = torch.randint(0, 5, (n,), generator=torch.Generator().manual_seed(4))
    return base, feats, model, labels


def _step(model, x, ei, ids, y, holder):
    for p in model.parameters():
        p.grad = None
    logits = model([x, ei, ids], holder=holder)
    F.cross_entropy(logits[ids], y).backward()
    return logits.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]


@pytest.mark.parametrize("kind", ["idgcn", "idgin"])
@pytest.mark.parametrize("threaded", [False, True])
def test_pipeline_batches_give_the_plain_path_results_bit_for_bit(dev, kind, threaded):
    from graphgym_amd import graph as G, harness as H
    from graphgym_amd.ego import ego_batch
    from graphgym_amd.pipeline import EgoBatchPipeline
    base, feats, model, labels = _setup(dev, kind)
    model.train()
    csr = "add" if kind == "idgcn" else "none"
    pipe = EgoBatchPipeline(base, feats, 2, prepare=model.prepare, device=dev, threaded=threaded, csr=csr)
    gen = torch.Generator().manual_seed(7)
    draws = [torch.randint(0, base.num_nodes, (96,), generator=gen) for _ in range(6)]
    draws[2][:2] = torch.tensor([0, 1])                                    # hub-centred egos (beyond the LDS table)
    pipe.submit(draws[0], labels[draws[0]])
    for k in range(6):
        b = pipe.get()
        assert b.prepared and getattr(b.holder, "_mp_graph_cache", None) is not None
        before = G.builds_by_this_thread()
        got = _step(model, b.x, b.edge_index, b.ids, b.y, b.holder)
        assert G.builds_by_this_thread() == before, "the step built graph structures the pipeline should have prepared"
        pipe.done()
        if k + 1 < 6:
            pipe.submit(draws[k + 1], labels[draws[k + 1]])
        del b                                                              # (the pipeline keeps the batch alive as long as needed)
        # the plain path on the same centres: COO from the expansion, CSR / norm / transpose built lazily by the layers
        cen = draws[k].to(dev)
        ei, orig, ids, _ = ego_batch(base, cen, 2)
        want = _step(model, feats.index_select(0, orig), ei, ids, labels[draws[k]].to(dev), H.Batch())
        assert torch.equal(got[0], want[0]), f"logits differ at step {k}"
        for a, w in zip(got[1], want[1]):
            assert torch.equal(a, w), f"a gradient differs at step {k}"
    pipe.close()


def test_pipeline_pauses_placement_and_resumes_it(dev):
    from graphgym_amd import placement
    from graphgym_amd.pipeline import EgoBatchPipeline
    base, feats, model, labels = _setup(dev, "idgcn", n=20_000)
    assert placement.enabled()
    pipe = EgoBatchPipeline(base, feats, 2, prepare=model.prepare, device=dev, csr="add")
    assert not placement.enabled()
    pipe.close()
    assert placement.enabled()

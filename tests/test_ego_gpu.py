"""GPU ego-net batcher vs the oracle's restatement of transform.py:11-38 (networkx) — node
numbering bit-exact, edge sets identical — on the committed golden graphs and fresh ones."""
import networkx as nx
import numpy as np
import pytest
import torch

from oracle import ref_layers as RL

pytestmark = pytest.mark.gpu


def directed(G):
    e = np.array(list(G.edges()), dtype=np.int64).reshape(-1, 2)
    return np.ascontiguousarray(np.concatenate([e, e[:, ::-1]], axis=0).T)


def expand_on_gpu(dev, base_ei, n, radius, centres=None):
    import graphgym_amd as ga
    from graphgym_amd.ego import ego_batch
    base = ga.CSRGraph.from_edge_index(torch.from_numpy(base_ei).to(dev), n)
    cen = torch.arange(n, device=dev) if centres is None else torch.as_tensor(centres, device=dev)
    return ego_batch(base, cen, radius)


def canon(ei, orig, ego_of):
    """expansion up to the relabelling inside each ego: {(centre, original u, original v)}"""
    ei, orig, ego_of = np.asarray(ei), np.asarray(orig), np.asarray(ego_of)
    assert (ego_of[ei[0]] == ego_of[ei[1]]).all()              # egos are disjoint components
    return set(zip(ego_of[ei[0]].tolist(), orig[ei[0]].tolist(), orig[ei[1]].tolist()))


def test_matches_committed_golden(dev, golden):
    z = golden("ego.npz")
    for k in range(4):
        n, radius = int(z[f"g{k}/base_n"]), int(z[f"g{k}/radius"])
        ei, orig, ids, ego_of = expand_on_gpu(dev, z[f"g{k}/base_edge_index"], n, radius)
        assert orig.numel() == int(z[f"g{k}/ego_n"])
        assert ids.cpu().tolist() == z[f"g{k}/node_id_index"].tolist()
        ref = z[f"g{k}/ego_edge_index"]
        assert ei.size(1) == ref.shape[1]
        assert canon(ei.cpu().numpy(), orig.cpu().numpy(), ego_of.cpu().numpy()) == \
            canon(ref, z[f"g{k}/orig_node"], z[f"g{k}/ego_of_node"])
        assert orig[:n].cpu().tolist() == list(range(n))        # centres keep their ids
        # the sizes of the egos, hence the first fresh id of each, are bit-exact
        assert np.array_equal(np.bincount(ego_of.cpu().numpy()), np.bincount(z[f"g{k}/ego_of_node"]))


@pytest.mark.parametrize("radius", [1, 2, 3, 5])
def test_against_networkx_restatement(dev, radius):
    G = nx.powerlaw_cluster_graph(40, 2, 0.3, seed=radius)
    G.add_edge(3, 3)                                            # a self loop survives the induced subgraph
    H, ids, h_orig, h_ego = RL.ego_nets(G, radius, return_map=True)
    ei, orig, idx, ego_of = expand_on_gpu(dev, directed(G), 40, radius)
    assert orig.numel() == H.number_of_nodes()
    hd = directed(H)
    assert canon(ei.cpu().numpy(), orig.cpu().numpy(), ego_of.cpu().numpy()) == \
        canon(hd, h_orig.numpy(), h_ego.numpy())
    # orig maps fresh ids back: members of ego c are exactly nx.ego_graph(G, c, radius)
    orig_c, ego_c = orig.cpu().numpy(), ego_of.cpu().numpy()
    for c in range(40):
        members = set(orig_c[ego_c == c].tolist())
        want_m = set(G.nodes) if radius > 4 else set(nx.ego_graph(G, c, radius=radius).nodes)
        assert members == want_m
        fresh = orig_c[40:][ego_c[40:] == c]
        assert (np.diff(fresh) > 0).all()                       # our fresh ids ascend with the original id


def test_subset_of_centres_on_a_large_graph(dev):
    import graphgym_amd as ga
    from graphgym_amd import graphgen
    from graphgym_amd.ego import ego_batch
    n = 200_000
    ei = graphgen.ba_edge_index(n, 5, seed=3, device=dev)
    base = ga.CSRGraph.from_edge_index(ei, n)
    cen = torch.tensor([0, 17, 150_000, 199_999, 4242], device=dev)
    e2, orig, idx, ego_of = ego_batch(base, cen, 2)
    # independent check with torch ops: 2-hop sets by boolean frontier expansion
    src, dst = ei[0], ei[1]
    for k, c in enumerate(cen.tolist()):
        seen = torch.zeros(n, dtype=torch.bool, device=dev)
        seen[c] = True
        for _ in range(2):
            nxt = torch.zeros_like(seen)
            nxt[dst[seen[src]]] = True
            seen |= nxt
        members = torch.sort(orig[ego_of == k]).values
        assert torch.equal(members, torch.nonzero(seen).view(-1))
        m = ego_of[e2[1]] == k
        induced = int((seen[src] & seen[dst]).sum())
        assert int(m.sum()) == induced
    assert orig[:5].tolist() == cen.tolist()
    # fresh ids of each ego ascend with the original id
    for k in range(5):
        o = orig[(ego_of == k)].cpu()
        rest = o[o != cen[k].item()] if k < 5 else o
        fresh = orig[5:][(ego_of[5:] == k)].cpu()
        assert torch.equal(fresh, torch.sort(fresh).values)


@pytest.mark.parametrize("radius", [1, 2, 3])
@pytest.mark.parametrize("loops", ["none", "add"])
def test_csr_written_by_the_expansion_equals_the_csr_built_from_its_edge_list(dev, radius, loops):
    """ego_batch(csr=...) hands back the batch's CSRGraph written by the expansion itself; it must be, entry for entry,
    what CSRGraph.from_edge_index builds from the returned edge_index (rowptr, col, eid) — with and without the added
    self loops — and its GCN normalisation and aggregation must give the same numbers"""
    import graphgym_amd as ga
    from graphgym_amd import graphgen, ops
    from graphgym_amd.ego import ego_batch
    n = 20_000
    ei0 = graphgen.ba_edge_index(n, 4, seed=radius, device=dev)
    base = ga.CSRGraph.from_edge_index(ei0, n)
    gen = torch.Generator().manual_seed(5)
    cen = torch.randint(0, n, (300,), generator=gen).to(dev)
    cen[:3] = torch.tensor([0, 1, 2], device=dev)                     # hub-centred egos (larger than the LDS table)
    ei, orig, ids, ego_of, g = ego_batch(base, cen, radius, csr=loops)
    assert g is not None and g.symmetric
    n2 = orig.numel()
    for dst_row in (0, 1):
        want = ga.CSRGraph.from_edge_index(ei, n2, dst_row=dst_row, add_self_loops=(loops == "add"))
        assert g.nnz == want.nnz and g.num_nodes == want.num_nodes
        assert torch.equal(g.rowptr, want.rowptr)
        assert torch.equal(g.col, want.col)
        if dst_row == 1:
            assert torch.equal(g.eid, want.eid)                       # positions in edge_index; -1 - row for added loops
    gn, wn = g.gcn_norm("row"), want.gcn_norm("row")
    assert torch.equal(gn.val, wn.val) and gn.symmetric
    x = torch.rand(n2, 64, device=dev)
    assert torch.equal(ops.spmm(gn, x, "sum"), ops.spmm(wn, x, "sum"))
    # the identity-branch operators through the ego-batch shortcut equal the general build's, field for field
    fast, slow = gn.id_branch(ids), gn._id_branch_build(ids)
    for f in ("rows", "crp", "slot", "val", "defer"):
        assert torch.equal(getattr(fast, f), getattr(slow, f)), f
    assert fast.n_rows == slow.n_rows and fast.t.nnz == slow.t.nnz and fast.t.num_nodes == slow.t.num_nodes
    assert torch.equal(fast.t.rowptr, slow.t.rowptr) and torch.equal(fast.t.col, slow.t.col)
    assert torch.equal(fast.t.val, slow.t.val)
    # its transpose is itself: the same numbers as the sorted transpose of the reference build
    assert gn.transpose() is gn
    assert torch.allclose(ops.spmm(gn.transpose(), x, "sum"), ops.spmm(wn.transpose(), x, "sum"), rtol=1e-6, atol=1e-6)
    # the 4-tuple form is unchanged, and a base graph with an explicit self loop falls back (fifth value None)
    assert len(ego_batch(base, cen, radius)) == 4
    loopy = ga.CSRGraph.from_edge_index(torch.cat([ei0, torch.tensor([[7], [7]], device=dev)], 1), n)
    assert ego_batch(loopy, cen[:5], radius, csr=loops)[4] is None


def test_attention_backward_on_the_expansion_s_own_csr(dev):
    """The CSR an expansion writes is flagged symmetric (transpose() is the graph itself), but attention scores and
    coefficients are per-entry values that are NOT symmetric: their backward passes permute them through the sorted
    transpose's entry map.  Scores, softmax and the weighted aggregation (TfgIDLayer.py:333-355), forward and all three
    gradients, must equal those on the CSR built from the returned edge list."""
    import graphgym_amd as ga
    from graphgym_amd import graphgen, ops
    from graphgym_amd.ego import ego_batch
    n = 5000
    base = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n, 3, seed=4, device=dev), n)
    cen = torch.randint(0, n, (64,), generator=torch.Generator().manual_seed(2)).to(dev)
    ei, orig, ids, ego_of, g = ego_batch(base, cen, 2, csr="add")
    assert g is not None and g.symmetric and g.transpose() is g
    n2 = orig.numel()
    want = ga.CSRGraph.from_edge_index(ei, n2, dst_row=1, add_self_loops=True)
    assert torch.equal(g.col, want.col) and not want.symmetric
    gen = torch.Generator().manual_seed(9)
    q0, k0, v0 = (torch.randn(n2, 32, generator=gen).to(dev) for _ in range(3))
    up = torch.randn(n2, 32, generator=gen).to(dev)
    outs = []
    for G in (g, want):
        q, k, v = (t.clone().requires_grad_(True) for t in (q0, k0, v0))
        for heads in (1, 4):
            sc = ops.sddmm_dot(G, q, k, heads=heads, scale=0.25)
            a = ops.edge_softmax(G, sc)
            y = ops.spmm_edge_values(G, a, v, heads=heads)
            y.backward(up)
        outs.append((y.detach(), q.grad, k.grad, v.grad))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("radius", [1, 2])
def test_repeated_isolated_and_single_centres(dev, radius):
    """A sampled batch may name a centre twice (two separate components, transform.py:24-36 builds one ego per listed
    node), a centre without any edge (an ego of one node), or a single centre; each ego of a batch must be what the
    batch of that centre alone gives, whatever else is in the batch."""
    import graphgym_amd as ga
    from graphgym_amd import graphgen
    from graphgym_amd.ego import ego_batch
    n = 3000
    ei = graphgen.ba_edge_index(n - 2, 3, seed=6, device=dev)               # nodes n-2, n-1 have no edges
    base = ga.CSRGraph.from_edge_index(ei, n)
    cen = torch.tensor([5, 0, 5, n - 1, 77, 0, n - 2, 5], device=dev)       # 0: a hub; repeats; isolated nodes
    for csr in (None, "add"):
        out = ego_batch(base, cen, radius, csr=csr)
        e2, orig, ids, ego_of = out[:4]
        assert ids.tolist() == list(range(cen.numel())) and orig[:cen.numel()].tolist() == cen.tolist()
        sizes = torch.bincount(ego_of, minlength=cen.numel())
        for k, c in enumerate(cen.tolist()):
            one = ego_batch(base, torch.tensor([c], device=dev), radius, csr=csr)
            assert int(sizes[k]) == one[1].numel()
            mem = orig[ego_of == k]
            assert torch.equal(torch.sort(mem).values, torch.sort(one[1]).values)
            m = ego_of[e2[0]] == k
            got = set(zip(orig[e2[0][m]].tolist(), orig[e2[1][m]].tolist()))
            want = set(zip(one[1][one[0][0]].tolist(), one[1][one[0][1]].tolist()))
            assert got == want and int(m.sum()) == one[0].size(1)
        assert int(sizes[3]) == 1 and int(sizes[6]) == 1                    # the isolated centres
        assert torch.equal(sizes[0], sizes[2]) and torch.equal(sizes[0], sizes[7]) and torch.equal(sizes[1], sizes[5])
        if csr is not None:
            g = out[4]
            want = ga.CSRGraph.from_edge_index(e2, orig.numel(), dst_row=1, add_self_loops=True)
            assert torch.equal(g.rowptr, want.rowptr) and torch.equal(g.col, want.col)

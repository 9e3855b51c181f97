"""`node_identity` features of ID-GNN Fast on the engine — graphgym/contrib/transform/identity.py:7-35,
called per graph from feature_augment.py:75-79 as compute_identity(edge_index, num_nodes, feature_dim).

The reference densifies the normalised adjacency (n x n) and multiplies it k - 1 times to read the diagonals of
A_hat^1 .. A_hat^k (weights of closed walks that return to the node).  Here the powers are never formed: with
X_0 = the indicator columns of a block of nodes, X_t = A_hat X_{t-1} is the aggregation kernel at width
`block`, and diag(A_hat^t) for those nodes is X_t[node, its column].  For a batch of disjoint graphs the column
of a node is its position inside its own graph, so ALL graphs advance together at width max-graph-size
(64 for the bundled synthetic sets): k aggregations for the whole batch instead of one dense n^3 chain per graph.
"""
import torch

from . import ops
from .graph import CSRGraph


def identity_graph(edge_index, n, edge_weight=None, improved=False):
    """identity.py:7-23: add_remaining_self_loops (fill 1 | 2), D^-1/2 A D^-1/2 with the degree scattered on
    edge_index[0]; a power's diagonal does not depend on which index is called the row"""
    g = CSRGraph.from_edge_index(edge_index, n, edge_weight, remove_self_loops=True, add_self_loops=True,
                                 keep_loop_weight=True, fill=2.0 if improved else 1.0)
    return g.gcn_norm("col")


def compute_identity(edge_index, n, k, batch=None, block=256, graph=None):
    """[n, k] with column t = diag(A_hat^(t+1)).  batch (optional, LongTensor [n], sorted graph ids of a disjoint
    union): nodes of different graphs share columns, so the width is the largest graph, not n."""
    n, k = int(n), int(k)
    dev = edge_index.device
    g = graph if graph is not None else identity_graph(edge_index, n)
    out = torch.empty((n, k), dtype=torch.float32, device=dev)
    if n == 0 or k == 0:
        return out
    node = torch.arange(n, device=dev)
    if batch is None:
        local, width = node, n
    else:
        b = batch.to(torch.int64)
        first = torch.zeros(int(b.max().item()) + 1, dtype=torch.int64, device=dev)
        first.scatter_reduce_(0, b, node, reduce="amin", include_self=False)
        local = node - first[b]                       # position inside the node's own graph
        width = int(local.max().item()) + 1
    block = max(1, min(int(block), width))
    for c0 in range(0, width, block):
        c1 = min(c0 + block, width)
        mine = (local >= c0) & (local < c1)           # nodes whose column lives in this block
        rows = node[mine]
        cols = local[mine] - c0
        x = torch.zeros((n, c1 - c0), dtype=torch.float32, device=dev)
        x[rows, cols] = 1.0
        for t in range(k):
            x, _ = ops._raw_spmm(g, x, 0)             # X_t = A_hat X_{t-1}  (sum aggregation, stored weights)
            out[rows, t] = x[rows, cols]
    return out

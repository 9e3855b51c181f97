"""Every kernel family of the path at the C4 shape (N = 10^7, 1.1*10^8 stored entries, d = 256) against its roofline:
algorithmic bytes (or flops) per launch / measured time.  Prints JSON lines and a markdown table."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops, nn as mpnn, placement
from graphgym_amd.graph import CSRGraph
dev = torch.device("cuda:0")
n, d = int(os.environ.get("NODES", "10000000")), 256
HBM, MFMA = 8000.0, 157.3
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
E_in = ei.size(1)

def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
rows = []
def rec(name, ms, gbytes=None, tflop=None, note=""):
    r = {"kernel": name, "ms": round(ms, 3)}
    if gbytes is not None:
        r["algorithmic_GB"] = round(gbytes, 2); r["GBps"] = round(gbytes / ms * 1e3, 0); r["frac_hbm_8TBps"] = round(gbytes / ms * 1e3 / HBM, 3)
    if tflop is not None:
        r["TFLOP"] = round(tflop, 3); r["TFLOPps"] = round(tflop / ms * 1e3, 1); r["frac_f32_mfma_157TF"] = round(tflop / ms * 1e3 / MFMA, 3)
    r["note"] = note
    rows.append(r); print(json.dumps(r), flush=True)

# build pipeline (sort-bound)
def build():
    return CSRGraph.from_edge_index(ei, n, add_self_loops=True)
t = timeit(build, iters=3, warm=1)
rec("csr_from_coo (keys, radix sort, rowptr, emit) + self loops", t, (E_in + n) * 24 * 2 / 1e9, note="~2 sort passes x 24 B per key (estimate); sort-bound")
g0 = build()
nnz = g0.nnz
t = timeit(lambda: g0.degree("row"))
rec("degree_row (weighted row sums)", t, (nnz * 4 + n * 8) / 1e9)
g = g0.gcn_norm()
t = timeit(lambda: g0.gcn_norm())
rec("gcn_norm = degree + inv_sqrt + norm_edges", t, (nnz * 4 + n * 8 + nnz * 16 + n * 8) / 1e9, note="dinv[col] is a random 4-byte gather per entry")
def tr():
    g._t = None
    src = getattr(g, "_pattern_of", None)
    if src is not None:
        src._t = None
    return g._transpose_sorted()
t_tr = timeit(tr, iters=3, warm=1)
gt = g._transpose_sorted()
rec("csr_transpose (keys, radix sort on the column bits, emit; only for operators that are not symmetric)", t_tr, nnz * 24 * 2 / 1e9, note="pattern shared with the un-normalised graph: values permuted only")
def sym():
    g.__dict__.pop("_sym_known", None)
    g.symmetric = False
    return g.is_symmetric(run=True)
t_sym = timeit(sym, iters=3, warm=1)
rec("is_symmetric (one binary search per entry + one host read; once per graph)", t_sym, nnz * 12 / 1e9, note=f"-> {g.is_symmetric(run=True)}; opt-in (MP_SYM_CHECK=1): it costs what the sorted transpose costs")
g.__dict__.pop("_sym_known", None)      # (the rest of the table runs as the default does: nobody asked)
g.symmetric = False
def pl():
    g._plan = None
    return g.plan()
rec("plan build (segments + hubs; once per graph)", timeit(pl, iters=3, warm=1), note="cached with the pattern (first build: ~50 us of kernels + one host read)")
del g0

x = placement.empty_or_torch((n, d), dev)            # operands through the engine's placement, as the operators allocate
x.uniform_(-1, 1)
y = placement.empty_or_torch((n, d), dev, reads=(x,))
agg_bytes = (nnz * (d * 4 + 8) + n * (d * 4 + 4)) / 1e9
for red, name in ((0, "sum"), (1, "mean"), (2, "max")):
    t = timeit(lambda: ops._raw_spmm(g, x, red, out=y, want_argmax=False))
    rec(f"agg_rows {name} (weighted; tile structure)", t, agg_bytes)
t = timeit(lambda: ops._raw_spmm(gt, x, 0, out=y))
rec("agg_rows sum on the transposed operator (backward)", t, agg_bytes)
t = timeit(lambda: ops._raw_spmm(g, x, 0, S=x, self_scale=1.0, out=y))
rec("agg_rows sum + self term (GIN combine)", t, agg_bytes + n * d * 4 / 1e9)
ids = torch.arange(0, n, 100, device=dev)
with torch.no_grad():
    t_two = timeit(lambda: ops.idgnn_aggregate(g, ids, x))
rec("agg_rows two-branch (P, Q; tile structure + identity rows, 1 % identity nodes)", t_two, agg_bytes + n * d * 4 / 1e9)
# attention pieces
s = None
t = timeit(lambda: ops._raw_sddmm_dot(g, x, x, 1, 1.0))
rec("sddmm_stream (dot-product scores, 1 head)", t, (nnz * (d * 4 + 12) + n * d * 4) / 1e9, note="K[col] row per entry, Q[row] once per row")
sc = ops._raw_sddmm_dot(g, x, x, 1, 1.0 / 16)
t = timeit(lambda: ops.edge_softmax(g, sc))
rec("row_softmax (per destination row)", t, (nnz * 8 + n * 4) / 1e9, note="a lane per short row (scores in registers), 16 lanes per medium row, the workgroup per hub row")
del sc
# dense transform kernels
W = torch.randn(d, d, device=dev) * 0.05
b = torch.randn(d, device=dev)
t = timeit(lambda: ops._dense_into(y, x, W, b, True))
rec("dense_x3: relu(P W + b), the streaming transform (LDS-DMA staged, bf16x3)", t, 2.0 * n * d * 4 / 1e9,
    tflop=2.0 * n * d * d / 1e12, note="reads P and writes out once; 6 bf16 MFMAs per fp32 product term")
t = timeit(lambda: ops.times_wt(x, W))
rec("dense_x3 with W^T: the input gradient g W^T", t, 2.0 * n * d * 4 / 1e9, tflop=2.0 * n * d * d / 1e12)
os.environ["MP_X3"] = "0"
t = timeit(lambda: ops._dense_into(y, x, W, b, True))
os.environ["MP_X3"] = "1"
rec("dense_fused, the general kernel (any shape, dual product)", t, tflop=2.0 * n * d * d / 1e12)
t = timeit(lambda: torch.mm(x, W, out=y))
rec("library fp32 GEMM (torch.mm) for comparison", t, tflop=2.0 * n * d * d / 1e12)
t = timeit(lambda: ops._raw_dense_wgrad(x, y, want_bias=True))
rec("dense_wgrad (+ bias gradient)", t, tflop=2.0 * n * d * d / 1e12)
t = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True, out=y))
rec("agg_dense (aggregate -> transform, one kernel; bf16x3 product)", t, agg_bytes, tflop=2.0 * n * d * d / 1e12, note="moves the aggregation's bytes AND does the transform's flops (fp32-accurate three-way bf16 split on the bf16 matrix pipe)")
t = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=False))
rec("agg_dense, exact-f32 MFMA product (round 1's form)", t, agg_bytes, tflop=2.0 * n * d * d / 1e12)
t = timeit(lambda: ops._raw_agg_dense(g, x, W, b, True, out=y, S=x, self_scale=1.0))
rec("agg_dense with a self term (GIN's (1 + eps) x + sum, then the first Dense)", t, agg_bytes + n * d * 4 / 1e9, tflop=2.0 * n * d * d / 1e12)
if os.environ.get("KT_F512", "1") == "1":     # config C5's width: two K halves over the row tile
    x5 = torch.empty((n, 512), device=dev).uniform_(-1, 1)
    y5 = placement.empty_or_torch((n, 512), dev, reads=(x5,))
    W5 = torch.randn(512, 512, device=dev) * 0.03
    b5 = torch.randn(512, device=dev)
    t = timeit(lambda: ops._raw_agg_dense(g, x5, W5, b5, True, out=y5))
    rec("agg_dense F = 512 -> 512 (two K halves per 64-row tile, both column blocks walk K together)", t,
        (nnz * (512 * 4 + 8) + n * (512 * 4 + 4)) / 1e9, tflop=2.0 * n * 512 * 512 / 1e12,
        note="index / value streams are read once per K half")
    del x5, y5, W5, b5
    torch.cuda.empty_cache()
# the ID layer: act(A (x W + S x W_id) + b), 1 % identity nodes
Wid = torch.randn(d, d, device=dev) * 0.05
with torch.no_grad():
    t_id = timeit(lambda: ops.agg_dense_id(g, x, W, Wid, ids, bias=b, relu=True))
    def id_two_kernels():
        P, Q = ops.idgnn_aggregate(g, ids, x)
        return ops.dense_fused(P, W, Q, Wid, b, relu=True)
    st0 = dict(placement.stats())
    t_id2 = timeit(id_two_kernels, iters=5, warm=4)
    st1 = placement.stats()
    print(json.dumps({"placement_during_round1_form": {k: st1[k] - st0[k] for k in ("allocations", "probed_pairs", "searches", "retries")
                                                       if isinstance(st1.get(k), (int, float))}}), flush=True)
rec("ID-GCN layer, one-kernel form (agg_dense + identity fix-up)", t_id, agg_bytes, tflop=2.0 * n * d * d / 1e12,
    note="round 1's form below: two-branch aggregation + dual GEMM")
rec("ID-GCN layer, round-1 form (two-branch aggregation, then P W + Q W_id)", t_id2, agg_bytes + 3 * n * d * 4 / 1e9, tflop=4.0 * n * d * d / 1e12)
# weight gradient with the ReLU backward folded in vs mask pass + weight gradient
yr = torch.relu(y)
def mask_then_wgrad():
    gm = torch.ops.aten.threshold_backward(x, yr, 0.0)
    return ops._raw_dense_wgrad(x, gm, want_bias=True)
t = timeit(lambda: ops._raw_dense_wgrad_relu(x, x, yr, want_bias=True))
rec("dense_wgrad with the ReLU mask folded in (+ masked gradient out)", t, tflop=2.0 * n * d * d / 1e12)
x1 = torch.ones(n, 1, device=dev)
t = timeit(lambda: ops._raw_dense_wgrad_relu(x1, x, yr, want_bias=True, want_gm=False))
rec("first-layer weight gradient (F = 1) with the ReLU mask, no masked-gradient output", t, 2.0 * n * d * 4 / 1e9)
del x1
t = timeit(mask_then_wgrad)
rec("threshold_backward pass + dense_wgrad (round 1's backward)", t, tflop=2.0 * n * d * d / 1e12)
del yr
# softmax cross-entropy over 10^7 labelled rows, 7 classes
z = torch.randn(n, 7, device=dev).requires_grad_(True)
lab = torch.randint(0, 7, (n,), device=dev)
def ce_engine():
    z.grad = None
    mpnn.softmax_cross_entropy(z, lab).backward()
def ce_torch():
    z.grad = None
    torch.nn.functional.cross_entropy(z, lab).backward()
rec("softmax cross-entropy fwd + bwd, engine", timeit(ce_engine), 4 * n * 7 * 4 / 1e9)
rec("softmax cross-entropy fwd + bwd, torch", timeit(ce_torch), 4 * n * 7 * 4 / 1e9)
del z, lab
# batch norm
bn = mpnn.BatchNorm1d(d, relu=True).to(dev)
xr = x.clone().requires_grad_(True)
t_f = timeit(lambda: bn(x))
rec("batchnorm + ReLU forward (statistics pass + apply pass)", t_f, 3 * n * d * 4 / 1e9)
out = bn(xr); up = torch.rand_like(out)
def bwd():
    o = bn(xr); o.backward(up); xr.grad = None
t_fb = timeit(bwd)
rec("batchnorm + ReLU backward (two passes; the mask is recomputed from x, y is not read)", t_fb - t_f, 5 * n * d * 4 / 1e9, note="time = (forward + backward) - forward")
del out, up, xr
# identity rows
u = torch.rand(ids.numel(), d, device=dev)
t = timeit(lambda: ops.gather_rows(x, ids))
rec("rows_gather (identity rows)", t, 2 * ids.numel() * d * 4 / 1e9, note="10^5 rows: launch / latency-bound")
del u
torch.cuda.empty_cache()
# per-batch graph work at the size of a training step's batch (6 * 10^6 stored entries): the general build, and what an
# ego batch pays instead (the expansion writes the CSR; symmetric: no transpose)
nb = 600_000
eb = graphgen.ba_edge_index(nb, 5, 7, device=dev)
def build_b():
    return CSRGraph.from_edge_index(eb, nb, add_self_loops=True)
t_b = timeit(build_b, iters=5, warm=2)
gb_ = build_b()
rec("batch-size csr_from_coo + self loops (6.6e6 entries)", t_b, (eb.size(1) + nb) * 24 * 2 / 1e9, note="general batches; sort-bound")
def tr_b():
    gb_._t = None
    return gb_._transpose_sorted()
rec("batch-size csr_transpose", timeit(tr_b, iters=5, warm=2), gb_.nnz * 24 * 2 / 1e9)
def sym_b():
    gb_.__dict__.pop("_sym_known", None)
    gb_.symmetric = False
    return gb_.is_symmetric(run=True)
rec("batch-size is_symmetric", timeit(sym_b, iters=5, warm=2), gb_.nnz * 12 / 1e9, note=f"-> {gb_.is_symmetric(run=True)}")
gb_.__dict__.pop("_sym_known", None)
gb_.symmetric = False
rec("batch-size gcn_norm", timeit(lambda: gb_.gcn_norm(), iters=5, warm=2), (gb_.nnz * 20 + nb * 16) / 1e9)
from graphgym_amd.ego import ego_batch
base = CSRGraph.from_edge_index(graphgen.ba_edge_index(2_000_000, 5, 11, device=dev), 2_000_000)
cen = torch.randint(0, 2_000_000, (4096,), device=dev)
t_e = timeit(lambda: ego_batch(base, cen, 2, csr="add"), iters=5, warm=2)
eg = ego_batch(base, cen, 2, csr="add")
rec("ego batch: expansion + its CSR with self loops (4096 centres, radius 2, base 2e6 nodes)", t_e,
    note=f"{eg[1].numel()} nodes, {eg[4].nnz} stored entries; replaces sampler + csr_from_coo + transpose for ego batches")
rec("ego batch: gcn_norm of the expansion's CSR", timeit(lambda: eg[4].gcn_norm("row"), iters=5, warm=2), (eg[4].nnz * 20 + eg[1].numel() * 16) / 1e9)
print("\n| kernel | ms | algorithmic GB | GB/s | % of 8 TB/s | TFLOP/s | % of 157 TF | note |\n|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['kernel']} | {r['ms']} | {r.get('algorithmic_GB', '')} | {r.get('GBps', '')} | "
          f"{'' if 'frac_hbm_8TBps' not in r else round(100 * r['frac_hbm_8TBps'], 1)} | {r.get('TFLOPps', '')} | "
          f"{'' if 'frac_f32_mfma_157TF' not in r else round(100 * r['frac_f32_mfma_157TF'], 1)} | {r['note']} |")

"""s_memtime stamps of the one-kernel layer at F = d = 128 on an EGO batch (4 096 centres of BA(2 * 10^6, 5), radius 2:
1.9 * 10^6 rows of ~3 entries) and, for comparison, on a BA graph of the same number of rows (~11 entries per row):
per item, how long the gathering waves gather, how long the multiplying waves multiply and store, and who waits at the
barrier.  Needs the timing build:  MP_ENGINE_LIB=graphgym_amd/csrc/libmpengine_timing.so  (fused.hip -DMP_FUSED_TIMING)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops, _lib
from graphgym_amd.ego import ego_batch
dev = torch.device("cuda:0")
d = 128
base = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(2_000_000, 5, seed=12345, device=dev), 2_000_000)
cen = torch.randint(0, 2_000_000, (4096,), generator=torch.Generator().manual_seed(1)).to(dev)
ei, orig, ids, ego_of, g_ego = ego_batch(base, cen, 2, csr="add")
n2 = orig.numel()
graphs = {"ego_batch": g_ego.gcn_norm("row"),
          "ba_same_rows": ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n2, 5, 7, device=dev), n2, add_self_loops=True).gcn_norm("row")}
L = C.CDLL(_lib.LIB_PATH)
L.mp_debug_read.argtypes = [C.c_void_p, C.c_size_t]
W = torch.randn(d, d, device=dev) * 0.05
b = torch.randn(d, device=dev)
for name, g in graphs.items():
    x = torch.empty((g.num_nodes, d), device=dev).uniform_(-1, 1)
    y = torch.empty((g.num_nodes, d), device=dev)
    for _ in range(3):
        ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=True); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(1 << 18, dtype=np.int64)
    assert L.mp_debug_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    t = buf[: 8 * 1000 * 8].reshape(8, 1000, 8).astype(np.float64)                  # s_memtime ticks
    # the stamp buffer keeps older launches' values behind this launch's last item: per sampled workgroup, the items of
    # THIS launch are the leading run of increasing P0 stamps
    raw = t[:, :, 0]
    runs = [int(np.argmax(np.diff(raw[w]) <= 0)) if (np.diff(raw[w]) <= 0).any() else 999 for w in range(8)]
    # ticks -> us: a sampled workgroup's stamps span the launch (the counter ran at ~1.6 GHz on the boxes of round 4,
    # not the 100 MHz round 3's script assumed), calibrated on the median workgroup against the launch's event time
    spans = sorted(float(raw[w, max(runs[w] - 1, 0)] - raw[w, 0]) for w in range(8))
    ticks_per_us = spans[3] / (e0.elapsed_time(e1) * 1e3)
    t = t / ticks_per_us
    items = int(min(900, min(runs) - 2))
    sl = slice(10, max(items, 20))
    P0, P1, P2, P3 = (t[:, sl, k] for k in range(4))
    C4, C5, C6, C7 = (t[:, sl, k] for k in range(4, 8))
    print(json.dumps({"graph": name, "rows": g.num_nodes, "entries_per_row": round(g.nnz / g.num_nodes, 2),
                      "items_per_sampled_workgroup": runs, "ticks_per_us": round(ticks_per_us, 1),
                      "kernel_ms": round(e0.elapsed_time(e1), 3), "lib": os.path.basename(_lib.LIB_PATH),
                      "item_period_us": round(float((P0[:, 1:] - P0[:, :-1]).mean()), 2),
                      "producer_gather_us": round(float((P1 - P0).mean()), 2),
                      "producer_wait_b1_us": round(float((P2 - P1).mean()), 2),
                      "producer_carry_b2_us": round(float((P3 - P2).mean()), 2),
                      "consumer_busy_us": round(float((C6[:, 1:] - C7[:, :-1]).mean()), 2),   # leaves b2 -> back at b1
                      "consumer_wait_b1_us": round(float((C7 - C6).mean()), 2)}), flush=True)
    del x, y

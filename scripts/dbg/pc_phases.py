"""s_memtime stamps of the producer/consumer one-kernel layer (library built with -DMP_FUSED_TIMING): per item, when the
producers finish gathering, when the consumers finish multiplying / storing, and who waits for whom at the barriers."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops, _lib, placement
dev = torch.device("cuda:0")
n, d = 10_000_000, int(os.environ.get("DIM", "512"))
g = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n, 5, 12345, device=dev), n, add_self_loops=True).gcn_norm("row")
torch.cuda.empty_cache()
x = torch.empty((n, d), device=dev).uniform_(-1, 1)
y = placement.empty_or_torch((n, d), dev, reads=(x,), tries=9, accept=-1.0)
W = torch.randn(d, d, device=dev) * 0.05
b = torch.randn(d, device=dev)
for _ in range(3):
    ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops._raw_agg_dense(g, x, W, b, True, out=y, bf16x3=True); e1.record(); torch.cuda.synchronize()
L = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(1 << 18, dtype=np.int64)
L.mp_debug_read.argtypes = [C.c_void_p, C.c_size_t]
assert L.mp_debug_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
t = buf[: 8 * 1000 * 8].reshape(8, 1000, 8).astype(np.float64) / 100.0          # us (s_memtime: 100 MHz)
# item `it` of the producers is consumed in iteration it + 1 of the consumers
sl = slice(50, 950)
P0, P1, P2, P3 = (t[:, sl, k] for k in range(4))
C4, C5, C6, C7 = (t[:, sl, k] for k in range(4, 8))
res = {"d": d, "kernel_ms": e0.elapsed_time(e1), "lib": os.path.basename(_lib.LIB_PATH),
       "item_period_us": float((P0[:, 1:] - P0[:, :-1]).mean()),
       "producer_gather_us": float((P1 - P0).mean()),            # top of the iteration -> arrival at b1
       "producer_wait_b1_us": float((P2 - P1).mean()),           # waiting for the other producers / the consumers
       "producer_carry_b2_us": float((P3 - P2).mean()),
       "consumer_mfma_us": float((C5 - C4).mean()),
       "consumer_store_us": float((C6 - C5).mean()),
       "consumer_work_us": float((C6 - C4).mean()),
       "consumer_wait_b1_us": float((C7 - C6).mean()),
       "frac_items_consumer_last_at_b1": float(((C6 > P1)).mean())}
print(json.dumps(res))

"""Per-phase s_memtime timestamps of sampled tiles of the one-kernel layer (library built with -DMP_FUSED_TIMING)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import graphgym_amd as ga
from graphgym_amd import graphgen, ops, _lib
dev = torch.device("cuda:0")
n, d = 10_000_000, 256
g = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n, 5, 12345, device=dev), n, add_self_loops=True).gcn_norm("row")
torch.cuda.empty_cache()
from graphgym_amd import placement
x = torch.empty((n, d), device=dev).uniform_(-1, 1)
y = placement.empty_or_torch((n, d), dev, reads=(x,), tries=9, accept=-1.0)     # the best of nine positions
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
t = buf[: 2048 * 64].reshape(2048, 4, 16)[:, :, :8].astype(np.float64)        # [tile, wave, slot]
ok = (t[:, :, 0] > 0).all(axis=1) & (t[:, :, 7] > t[:, :, 0]).all(axis=1)
t = t[ok]
t0 = t[:, :, 0].min(axis=1, keepdims=True)
names = ["start", "T init done", "after barrier 1", "phase A done", "after barrier 2", "carries+P done", "phase B done", "store done"]
rel = t - t0[:, :, None]
clk = 100e6   # s_memtime ticks at 100 MHz on gfx9 (constant clock)
res = {"place": getattr(y, "_mp_place", None), "lib": os.path.basename(_lib.LIB_PATH), "kernel_ms": e0.elapsed_time(e1), "tiles_sampled": int(ok.sum()),
       "mean_us_by_slot_and_wave": {names[s]: [round(float(rel[:, w, s].mean() / clk * 1e6), 2) for w in range(4)] for s in range(8)},
       "median_tile_total_us": float(np.median(rel[:, :, 7].max(axis=1)) / clk * 1e6),
       "phaseA_us_mean_per_wave": float((t[:, :, 3] - t[:, :, 2]).mean() / clk * 1e6),
       "phaseA_wait_at_barrier_us_mean": float((t[:, :, 4] - t[:, :, 3]).mean() / clk * 1e6),
       "phaseB_us_mean": float((t[:, :, 6] - t[:, :, 5]).mean() / clk * 1e6),
       "store_us_mean": float((t[:, :, 7] - t[:, :, 6]).mean() / clk * 1e6),
       "init_us_mean": float((t[:, :, 1] - t[:, :, 0]).mean() / clk * 1e6),
       "carries_us_mean": float((t[:, :, 5] - t[:, :, 4]).mean() / clk * 1e6)}
print(json.dumps(res))

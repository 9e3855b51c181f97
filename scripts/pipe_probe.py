"""Where do the 70-80 ms stalls of the batch build come from?  Builds 12 batches under different conditions and prints the
per-phase wall times (MP_PIPE_TIMING) with allocator / collector counters."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MP_PIPE_TIMING"] = "1"
import graphgym_amd as ga  # noqa: E402
from graphgym_amd import graphgen, harness as H  # noqa: E402
from graphgym_amd.pipeline import EgoBatchPipeline  # noqa: E402

dev = torch.device("cuda:0")
n0, B = 2_000_000, 4096
base = ga.CSRGraph.from_edge_index(graphgen.ba_edge_index(n0, 5, seed=12345, device=dev), n0)
x_base = torch.rand((n0, 128), device=dev)
model = H.TfgNodeModel("idgcn", 128, 128, 7).to(dev)
labels = torch.randint(0, 7, (n0,))
KEYS = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")
for mode in sys.argv[1:] or ["gc-on", "gc-off"]:
    if mode == "gc-off":
        gc.collect(); gc.freeze(); gc.disable()
    p = EgoBatchPipeline(base, x_base, 2, prepare=lambda i, h: model.prepare(i, h), device=dev)
    p.side = torch.cuda.current_stream(dev)
    for k in range(12):
        cen = torch.randint(0, n0, (B,), generator=torch.Generator().manual_seed(k))
        st0 = torch.cuda.memory_stats()
        g0 = [s["collections"] for s in gc.get_stats()]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p.submit(cen, labels[cen])
        b = p.get(); torch.cuda.synchronize(); t2 = time.perf_counter()
        st1 = torch.cuda.memory_stats()
        g1 = [s["collections"] for s in gc.get_stats()]
        print(mode, k, f"total {1e3*(t2-t0):7.2f} ms", {k_: round(v, 1) for k_, v in b.timing.items()},
              {k_: st1[k_] - st0[k_] for k_ in KEYS if st1[k_] - st0[k_]}, "gc runs", [a - b_ for a, b_ in zip(g1, g0)],
              "reserved GB", round(torch.cuda.memory_reserved() / 1e9, 2), flush=True)
        del b

import time, torch, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphgym_amd as ga
from graphgym_amd import graphgen, ego
from graphgym_amd.ego import ego_batch
dev = torch.device("cuda:0")
out = []
for n, B in ((2_000_000, 4096), (10_000_000, 256), (10_000_000, 4096)):
    ei = graphgen.ba_edge_index(n, 5, seed=12345, device=dev)
    base = ga.CSRGraph.from_edge_index(ei, n)
    del ei
    g = torch.Generator().manual_seed(99)
    cen = torch.randperm(n, generator=g)[:B].to(dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e2, orig, ids, ego_of = ego_batch(base, cen, 2)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = dict(ego.last_stats)
    deg = (base.rowptr[1:] - base.rowptr[:-1]).long()
    st["neighbour_slots"] = int(deg[orig].sum())          # what the edge phase walks
    st["wave_iterations"] = int(((deg[orig] + 63) // 64).sum())
    st.update(n=n, B=B, ms=dt * 1e3, bytes_per_node=st["scratch_peak_bytes"] / max(st["nodes"], 1))
    print(json.dumps(st), flush=True)
    del base

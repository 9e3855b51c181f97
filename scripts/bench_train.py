"""Training-step measurements for the BASELINE.json configs that are whole training runs, not bare kernels
(developer benchmark, JSON lines; the driver contract lives in bench.py):

  C4  ginconv_tf / sageconv_tf on a 10M-node / 100M-edge BA graph, d = 256, full-graph step
  C5  idgin_tf (ID-GNN Full) on ego batches sampled from a 10M-node scale-free graph, d = 512

One process per GPU (torchrun sets RANK/WORLD_SIZE); every rank owns its own graph / ego batch and
gradients are averaged with one all-reduce per step (graphgym_amd.dist.GradBucket).
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import dist as D, graphgen, harness as H
from graphgym_amd.ego import ego_batch


def run(kind, n, d, steps, rank, world, dev, ego_centres=0, radius=2):
    ei = graphgen.ba_edge_index(n, 5, 12345 + rank, device=dev)
    ids = None
    if ego_centres:
        base = ga.CSRGraph.from_edge_index(ei, n)
        gen = torch.Generator().manual_seed(100 + rank)
        cen = torch.randint(0, n, (ego_centres,), generator=gen).to(dev)
        t0 = time.perf_counter()
        ei, orig, ids, _ = ego_batch(base, cen, radius)
        torch.cuda.synchronize()
        t_ego = time.perf_counter() - t0
        del base
        n_nodes, label_index = int(orig.numel()), ids
    else:
        t_ego, n_nodes, label_index = 0.0, n, torch.arange(n, device=dev)
    x = torch.ones(n_nodes, 1, device=dev)                        # node_feature = ones, as the bundled datasets
    labels = torch.randint(0, 10, (label_index.numel(),), device=dev)
    torch.manual_seed(0)
    model = H.TfgNodeModel(kind, 1, d, 10).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    bucket = D.GradBucket(model.parameters()) if world > 1 else None
    holder = H.Batch()

    def fl():
        inputs = [x, ei] + ([ids] if model.with_id else [])
        return H.tfg_loss(model(inputs, holder=holder), label_index, labels, model.kernel_parameters())
    from graphgym_amd import placement
    for _ in range(int(os.environ.get("WARM", "2"))):
        H.train_step(model, opt, fl, bucket)
    st0 = placement.stats(dev) if placement.enabled() else {}
    D.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        H.train_step(model, opt, fl, bucket)
    torch.cuda.synchronize(); D.barrier()
    dt = D.all_reduce_max((time.perf_counter() - t0) / steps, dev)
    edges = D.all_reduce_sum(ei.size(1), dev)
    if rank == 0:
        print(json.dumps({"what": f"{kind} training step", "n_gpus": world, "nodes_per_gpu": n_nodes,
                          "edges_per_gpu": int(ei.size(1)), "d": d, "conv_layers": len(model.convs), "ms_per_step": dt * 1e3,
                          "edges_per_s_fwd_bwd_all_layers": 2 * len(model.convs) * edges / dt,
                          "ego_batch_build_ms": t_ego * 1e3, "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9,
                          "reserved_GB": torch.cuda.memory_reserved() / 1e9,
                          "placement_in_timed_steps": {k: (placement.stats(dev)[k] - st0[k]) for k in
                                                       ("allocations", "probed_pairs", "memo_hits", "retries")} if st0 else None}),
              flush=True)
    del model, opt, x, ei
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--kinds", default="sage,gin,idgin")
    a = ap.parse_args()
    rank, local, world = D.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    table = {"sage": (256, 0), "gin": (256, 0), "idgin": (512, 256), "gcn": (256, 0), "gat": (256, 0),
             "idgcn": (256, 256)}
    for kind in a.kinds.split(","):
        d, centres = table[kind]
        try:
            run(kind, a.nodes, d, a.steps, rank, world, dev, ego_centres=centres)
        except torch.OutOfMemoryError as e:
            if rank == 0:
                print(json.dumps({"what": f"{kind} training step", "error": "out of memory", "nodes": a.nodes}), flush=True)
            torch.cuda.empty_cache()

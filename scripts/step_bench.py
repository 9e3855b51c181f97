"""Developer benchmark: one training step (forward + backward + Adam) of the reference's model shapes
on the engine, plus where the time goes (rocprof-free: event timing per phase)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graphgym_amd as ga
from graphgym_amd import graphgen, harness as H
from graphgym_amd.harness import Batch

dev = torch.device("cuda:0")

def run(kind, n, d, f_in, iters=5):
    ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
    batch = Batch(edge_index=ei, node_id_index=torch.arange(0, n, 100, device=dev))
    x = torch.rand(n, f_in, device=dev) * 2 - 1
    labels = torch.randint(0, 10, (n,), device=dev)
    idx = torch.arange(n, device=dev)
    model = H.TfgNodeModel(kind, f_in, d, 10).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    def fl():
        inputs = [x, ei] + ([batch.node_id_index] if model.with_id else [])
        return H.tfg_loss(model(inputs, holder=batch), idx, labels, model.kernel_parameters())
    for _ in range(2):
        H.train_step(model, opt, fl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        H.train_step(model, opt, fl)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    nnz = ei.size(1)
    print(f"{kind:7s} n={n} d={d} f_in={f_in}: {dt*1e3:8.2f} ms/step  ({3*nnz/dt/1e9:.2f} G layer-edges/s fwd, x2 with bwd)", flush=True)

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    for kind in ("gcn", "idgcn", "sage", "idsage", "gin", "idgin", "gat", "idgat"):
        run(kind, n, 256, 256)
    run("gcn", n, 256, 1)

"""GCN training step at the C2 size under rocprofv3 --kernel-trace --stats (developer profile)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import graphgen, harness as H
dev = torch.device("cuda:0")
kind = os.environ.get("KIND", "gcn")
n, d = 1_000_000, 256
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
batch = H.Batch(edge_index=ei, node_id_index=torch.arange(0, n, 100, device=dev))
x = torch.rand(n, d, device=dev) * 2 - 1
labels = torch.randint(0, 10, (n,), device=dev)
idx = torch.arange(n, device=dev)
model = H.TfgNodeModel(kind, d, d, 10).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01)
def fl():
    inputs = [x, ei] + ([batch.node_id_index] if model.with_id else [])
    return H.tfg_loss(model(inputs, holder=batch), idx, labels, model.kernel_parameters())
for _ in range(int(os.environ.get("ITERS", "6"))):
    H.train_step(model, opt, fl)
torch.cuda.synchronize()

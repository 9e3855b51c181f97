import os
"""One TF-path model's training step at 10M nodes under rocprofv3 (KIND=sage|gin|gcn...)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import graphgen, harness as H
dev = torch.device("cuda:0")
n, d = int(os.environ.get("NODES", "10000000")), int(os.environ.get("DIM", "256"))
kind = os.environ.get("KIND", "sage")
ei = graphgen.ba_edge_index(n, 5, 12345, device=dev)
x = torch.ones(n, 1, device=dev)
labels = torch.randint(0, 10, (n,), device=dev)
idx = torch.arange(n, device=dev)
model = H.TfgNodeModel(kind, 1, d, 10).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01)
holder = H.Batch()
ids = torch.arange(0, n, 100, device=dev)
def fl():
    inputs = [x, ei] + ([ids] if model.with_id else [])
    return H.tfg_loss(model(inputs, holder=holder), idx, labels, model.kernel_parameters())
for _ in range(int(os.environ.get("STEPS", "5"))):      # scripts/step_window.py takes the last two
    H.train_step(model, opt, fl)
torch.cuda.synchronize()

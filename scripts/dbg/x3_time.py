"""time of the streaming transform at 10^7 x 256 x 256 through the loaded library (MP_ENGINE_LIB: ablation builds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graphgym_amd import ops
dev = torch.device("cuda:0")
M, F, d = 10_000_000, int(os.environ.get("DIM", "256")), 256
P = torch.randn(M, F, device=dev); W = torch.randn(F, d, device=dev) / 16; b = torch.randn(d, device=dev)
out = torch.empty(M, d, device=dev)
def run(): ops._dense_into(out, P, W, b, True)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get("MP_ENGINE_LIB", "libmpengine.so")), "MP_X3_PC=" + os.environ.get("MP_X3_PC", "1"), f"F={F}: {e0.elapsed_time(e1) / 10:.3f} ms")

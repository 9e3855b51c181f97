"""Multi-GPU execution of the path: one process per GPU, independent units per rank.

GraphGym batches are disjoint unions of graphs (graphgym/loader.py:247-251) and ID-GNN
ego nets are disjoint by construction (graphgym/models/transform.py:24-36), so the
aggregation itself needs no data-path collective: whole graphs / ego nets are dealt to
ranks by nnz (LPT), each rank builds its own CSR, and the only exchange of a training
step is the gradient all-reduce (RCCL over xGMI through torch.distributed's "nccl"
backend; "gloo" on CPU for the tests).  The reference has no distributed code at all.
"""
import os

import torch
import torch.distributed as dist


def lpt_partition(costs, world_size):
    """Greedy longest-processing-time assignment of units (cost = stored entries) to ranks.
    Returns a list of index lists, deterministic for equal inputs."""
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    loads = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(costs[i])
    for p in parts:
        p.sort()
    return parts


def init_from_env(device_type=None):
    """(rank, local_rank, world_size); initialises the process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type is None:
        device_type = "cuda" if torch.cuda.is_available() else "cpu"
    if device_type == "cuda":
        if os.environ.get("MP_SHARE_DEVICE") == "1":      # rehearsal of N ranks on a 1-GPU box (gloo only)
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("MP_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def _coll_device(device):
    """collectives on scalars run on the GPU under RCCL and on the host under gloo"""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return device


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def all_reduce_max(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class GradBucket:
    """One flat fp32 bucket holding every parameter's gradient, all-reduced once per step.

    The models on this path are small (0.2 M - 3.3 M parameters, <= 13 MB): a single
    bucket keeps the all-reduce at one RCCL launch, latency-bound either way."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def all_reduce_mean(self):
        """average gradients over ranks in place (no-op for a single process)"""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        if dist.get_backend() == "gloo" and self.flat.is_cuda:   # rehearsal mode: stage through the host
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)

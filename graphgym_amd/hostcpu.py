"""How many host CPUs this process may actually use.

os.cpu_count() reports the machine (256 on an MI355X host); a container is usually held to a CPU-time quota (cgroup
cpu.max: 16 CPUs on the 1-GPU boxes this engine is measured on).  PyTorch sizes its intra-op (OpenMP) pool by the former,
and a 256-thread pool that spins after each small CPU operator burns the quota of a whole scheduling period: the kernel
then freezes EVERY thread of the process for the rest of the period — 70-80 ms stalls that land in whichever host
synchronisation comes next (measured: every ~3rd batch build of a 7 ms build loop, scripts/pipe_probe.py; none with the
pool held to the quota).  The reference pins its own pool (cfg.num_threads = 6, graphgym/config.py:57, run/main.py:31).
"""
import os


def effective_cpus():
    """CPUs usable by this process: min(affinity mask, cgroup quota), at least 1"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:                                                   # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except Exception:
        pass
    if quota is None:
        try:                                               # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def fit_torch_threads():
    """hold torch's intra-op pool to the CPUs this process may use (no-op when OMP_NUM_THREADS / MP_KEEP_THREADS is set
    or the pool already fits).  Returns the pool size in effect."""
    import torch
    if os.environ.get("MP_KEEP_THREADS") == "1" or os.environ.get("OMP_NUM_THREADS"):
        return torch.get_num_threads()
    n = effective_cpus()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()

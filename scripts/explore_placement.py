"""Placement study, round 2: does a timed copy between two chunks of one slab show the region structure,
at which chunk size, and how do torch-allocated tensors sit against it?
Writes gpurun_out/explore_placement.log (committed as profiles/r02_placement_map.log)."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphgym_amd import _lib   # noqa: E402
from graphgym_amd._lib import check, lib   # noqa: E402

out = open(os.path.join(ROOT, "gpurun_out", "explore_placement.log"), "w")


def P(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    out.write(s + "\n")
    out.flush()


def probe(src, dst, nbytes, reps=3):
    ms = C.c_float(0)
    check(lib().mp_probe_copy_ms(C.c_void_p(src), C.c_void_p(dst), nbytes, reps, C.byref(ms),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return ms.value


def main():
    torch.cuda.init()
    L = lib()
    free, total = torch.cuda.mem_get_info()
    P("free/total GB", free / 1e9, total / 1e9)
    GB = 1 << 30
    arena_gb = int(os.environ.get("ARENA_GB", "160"))
    check(L.mp_arena_create(arena_gb * GB), "arena_create")
    base = C.c_void_p()
    nb = C.c_size_t()
    check(L.mp_arena_info(C.byref(base), C.byref(nb), None, None))   # (base, bytes, in_use, largest_free)
    base = base.value
    P("arena base", hex(base), "bytes", nb.value)

    # 1. chunk-size sensitivity: src at 0, dst at +1 GB .. (same region presumably) vs far away
    for cb in (256 << 20, 512 << 20, 1 << 30, 2 << 30):
        row = []
        for off_gb in (4, 8, 16, 24, 32, 36, 40, 48, 64, 72, 80, 100, 120, 140):
            if (off_gb + 2) * GB > nb.value:
                break
            row.append("%d:%.3f" % (off_gb, probe(base, base + off_gb * GB, cb)))
        P("chunk %4d MB, src@0, dst@GB:ms" % (cb >> 20), " ".join(row))

    # 2. pairwise map at 4 GB granularity, 1 GB chunks
    step = 4
    n = arena_gb // step
    cb = 1 << 30
    t0 = time.time()
    M = [[0.0] * n for _ in range(n)]
    for i in range(n):
        for j in range(n):
            if i == j:
                M[i][j] = probe(base + i * step * GB, base + i * step * GB + 2 * GB, cb, 2)
            else:
                M[i][j] = probe(base + i * step * GB, base + j * step * GB, cb, 2)
    P("pairwise scan took %.1f s" % (time.time() - t0))
    lo = min(min(r) for r in M)
    P("min ms", lo)
    P("     " + " ".join("%3d" % (j * step) for j in range(n)))
    for i in range(n):
        P("%3d: " % (i * step) + " ".join("%3d" % int(round(100 * (M[i][j] / lo - 1))) for j in range(n)))

    # 3. foreign (torch-allocated) tensors against arena anchors
    ts = [torch.empty(10 * GB, dtype=torch.uint8, device="cuda") for _ in range(4)]
    for k, t in enumerate(ts):
        row = []
        for j in range(n):
            row.append("%3d" % int(round(100 * (probe(t.data_ptr(), base + j * step * GB, cb, 2) / lo - 1))))
        P("torch tensor %d @%s head vs arena chunks: " % (k, hex(t.data_ptr())) + " ".join(row))
        row = []
        for j in range(n):
            row.append("%3d" % int(round(100 * (probe(t.data_ptr() + 9 * GB, base + j * step * GB, cb, 2) / lo - 1))))
        P("torch tensor %d tail vs arena chunks:           " % k + " ".join(row))
    del ts
    torch.cuda.empty_cache()



if __name__ == "__main__":
    main()

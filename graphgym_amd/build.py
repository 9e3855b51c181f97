"""Build libmpengine.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m graphgym_amd.build            # build if sources are newer than the .so
    python -m graphgym_amd.build --force

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only container.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libmpengine.so")
SOURCES = ["spmm.hip", "fused.hip", "csr_build.hip", "attn.hip", "ego.hip", "gemm.hip", "dense_x3.hip", "bn.hip", "util.hip", "probe.hip", "loss.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "vecio.h"), os.path.join(CSRC, "bf16x3.h"), os.path.join(ROOT, "include", "mp_engine.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c",
               os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode(errors='replace')}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

"""The role-split kernels meet at hardware barriers that pair up BY COUNT (csrc/common.h: role_barrier): the producer
waves and the consumer waves of a workgroup run separate loops over the same item sequence, and each loop must execute
the same number of s_barrier per item.  The language does not promise that a compiler leaves such barriers alone (it may
merge, hoist or duplicate a convergent call), so this test reads what was EMITTED: it disassembles the gfx950 code
objects of the built engine, rebuilds each kernel's control-flow graph, finds its loops (strongly connected components)
and checks, for every instantiation,

    agg_dense_pc_kernel (fused.hip)     two loops with 2 barriers each (b1, b2) — producers and consumers — and outside
                                        them 3 barriers (the prologue's, and b1 + b2 of the consumers' first pass), + 2 with
                                        a self term (b0, once per role)
    dense_wgrad_pc_kernel (gemm.hip)    the MFMA loop with 1 barrier per step, the loaders' loop with NSET (2 with the
                                        ReLU mask, 3 without: the loop is unrolled over its register sets), 4 outside
    dense_x3_pc_kernel (dense_x3.hip)   1 and 1, 2 outside

Runs on the CPU: it needs the built .o files (build() makes them) and llvm-objdump, no GPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "graphgym_amd", "csrc")
OBJDUMP = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"

SYM = re.compile(r'^([0-9a-f]{16}) <(.+)>:$')
INS = re.compile(r'^\s+(\S+)(?:\s+(.*?))?\s*//\s*([0-9A-F]+):')
TGT = re.compile(r'<(.+?)(?:\+0x([0-9a-f]+))?>\s*$')


def disassemble(obj_name, tmp_path):
    obj = os.path.join(CSRC, obj_name)
    if not os.path.exists(obj) or not os.path.exists(OBJDUMP):
        pytest.skip(f"{obj_name} not built or llvm-objdump missing")
    local = os.path.join(str(tmp_path), obj_name)
    shutil.copy(obj, local)                                   # (--offloading extracts next to its input)
    subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=str(tmp_path))
    co = [f for f in os.listdir(str(tmp_path)) if f.startswith(obj_name + ".") and "gfx950" in f]
    assert len(co) == 1, co
    return subprocess.run([OBJDUMP, "-d", os.path.join(str(tmp_path), co[0])], check=True, capture_output=True,
                          text=True).stdout


def kernels_of(dis, want):
    """{kernel symbol: [(offset, mnemonic, branch target offset or None)]}"""
    out, cur, base = {}, None, 0
    for line in dis.splitlines():
        m = SYM.match(line)
        if m:
            cur = m.group(2) if want in m.group(2) else None
            base = int(m.group(1), 16)
            if cur:
                out[cur] = []
            continue
        if cur is None:
            continue
        m = INS.match(line)
        if not m:
            continue
        op, addr, tgt = m.group(1), int(m.group(3), 16) - base, None
        if op.startswith("s_cbranch") or op == "s_branch":
            t = TGT.search(line)
            assert t and t.group(1) == cur, line                # no branch leaves the kernel
            tgt = int(t.group(2) or "0", 16)
        assert not op.startswith("s_setpc") and not op.startswith("s_swappc"), line   # no indirect control flow
        out[cur].append((addr, op, tgt))
    return out


def barriers_by_loop(ins):
    """(sorted s_barrier counts of the kernel's outermost loops that hold any, s_barrier outside every loop)"""
    idx = {a: i for i, (a, _, _) in enumerate(ins)}
    leaders = {0}
    for i, (_, op, t) in enumerate(ins):
        if t is not None:
            leaders.add(idx[t])
        if (t is not None or op == "s_endpgm") and i + 1 < len(ins):
            leaders.add(i + 1)
    leaders = sorted(leaders)
    nb = len(leaders)
    blk_of = {}
    for b, s in enumerate(leaders):
        for i in range(s, leaders[b + 1] if b + 1 < nb else len(ins)):
            blk_of[i] = b
    succ, nbar = [[] for _ in range(nb)], [0] * nb
    for b, s in enumerate(leaders):
        e = (leaders[b + 1] if b + 1 < nb else len(ins)) - 1
        nbar[b] = sum(1 for i in range(s, e + 1) if ins[i][1] == "s_barrier")
        _, op, t = ins[e]
        if t is not None:
            succ[b].append(blk_of[idx[t]])
        if op not in ("s_branch", "s_endpgm") and e + 1 < len(ins):
            succ[b].append(blk_of[e + 1])
    # Tarjan's strongly connected components, iterative
    index, low, on, stack, comp = [None] * nb, [0] * nb, [False] * nb, [], [None] * nb
    counter = ncomp = 0
    for root in range(nb):
        if index[root] is not None:
            continue
        work = [(root, 0)]
        while work:
            v, pi = work[-1]
            if pi == 0:
                index[v] = low[v] = counter
                counter += 1
                stack.append(v)
                on[v] = True
            if pi < len(succ[v]):
                work[-1] = (v, pi + 1)
                w = succ[v][pi]
                if index[w] is None:
                    work.append((w, 0))
                elif on[w]:
                    low[v] = min(low[v], index[w])
            else:
                work.pop()
                if work:
                    low[work[-1][0]] = min(low[work[-1][0]], low[v])
                if low[v] == index[v]:
                    while True:
                        w = stack.pop()
                        on[w] = False
                        comp[w] = ncomp
                        if w == v:
                            break
                    ncomp += 1
    size, bars, cyclic = [0] * ncomp, [0] * ncomp, [False] * ncomp
    for b in range(nb):
        size[comp[b]] += 1
        bars[comp[b]] += nbar[b]
        cyclic[comp[b]] |= b in succ[b]
    is_loop = [size[c] > 1 or cyclic[c] for c in range(ncomp)]
    return (sorted(bars[c] for c in range(ncomp) if bars[c] and is_loop[c]),
            sum(bars[c] for c in range(ncomp) if not is_loop[c]))


def _template_args(sym, name):
    """the leading template arguments of a mangled instantiation as a list of ints / bools: 'ILi4ELb1E...' -> [4, True, ...]"""
    body = sym.split(name + "I", 1)[1]
    return [int(v) if k == "i" else bool(int(v)) for k, v in re.findall(r"L([ib])(\d+)E", body.split("EvN")[0].split("EvP")[0])]


def test_agg_dense_pc_kernel_roles_hold_equal_barrier_counts(tmp_path):
    ks = kernels_of(disassemble("fused.o", tmp_path), "agg_dense_pc_kernel")
    assert len(ks) >= 40                                         # every instantiation the dispatcher can reach
    seen = set()
    for sym, ins in ks.items():
        # <W, WEIGHTED, U, KH, NCB, PF, NT_OUT, BF16X3, TR, NP, NC, HAS_S, AGG_ONLY, ...>
        targs = _template_args(sym, "agg_dense_pc_kernel")
        has_s = targs[11]
        loops, outside = barriers_by_loop(ins)
        assert loops == [2, 2], (sym, loops, outside)            # b1 + b2 per item in the producers' AND the consumers' loop
        assert outside == 3 + (2 if has_s else 0), (sym, loops, outside)
        seen.add(has_s)
    assert seen == {False, True}


def test_dense_wgrad_pc_kernel_roles_hold_equal_barrier_counts(tmp_path):
    ks = kernels_of(disassemble("gemm.o", tmp_path), "dense_wgrad_pc_kernel")
    assert len(ks) == 4
    for sym, ins in ks.items():
        relu = _template_args(sym, "dense_wgrad_pc_kernel")[0]
        loops, outside = barriers_by_loop(ins)
        # one barrier per step in the MFMA loop; the loaders' loop is unrolled over its NSET register sets (its range is
        # rounded up to whole sets on the host side of the kernel, so both roles execute the same number of steps)
        assert loops == [1, 2 if relu else 3], (sym, loops, outside)
        assert outside == 4, (sym, loops, outside)


def test_dense_x3_pc_kernel_roles_hold_equal_barrier_counts(tmp_path):
    ks = kernels_of(disassemble("dense_x3.o", tmp_path), "dense_x3_pc_kernel")
    assert len(ks) >= 1
    for sym, ins in ks.items():
        loops, outside = barriers_by_loop(ins)
        assert loops == [1, 1] and outside == 2, (sym, loops, outside)

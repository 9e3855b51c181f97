"""The C-ABI shared library: loads, exports every symbol include/mp_engine.h declares,
and its host-only entry points behave.  No device work here (CPU suite)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from graphgym_amd import _lib
from graphgym_amd import build as mpbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mp_engine.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    mpbuild.build()
    assert os.path.exists(_lib.LIB_PATH)
    assert os.path.dirname(_lib.LIB_PATH).startswith(ROOT)


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mp_engine.h but not exported"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype"
    for n in _lib.PROTOTYPES:
        assert n in names, f"ctypes binds {n} which the header does not declare"


def test_version_and_status_strings():
    lib = _lib.lib()
    assert lib.mp_version() >= 100
    assert lib.mp_status_str(0) == b"ok"
    assert b"workspace" in lib.mp_status_str(3)
    with pytest.raises(_lib.EngineError):
        _lib.check(1, "x")


def test_argument_validation_without_device():
    lib = _lib.lib()
    nb = C.c_size_t(0)
    assert lib.mp_spmm_plan_bytes(-1, 0, None, C.byref(nb)) == 1
    assert lib.mp_spmm_plan_bytes(10, 2**31, None, C.byref(nb)) == 2          # int32 index limit
    assert lib.mp_spmm_plan_bytes(1000, 10000, None, C.byref(nb)) == 0 and nb.value > 0
    default_bytes = nb.value
    assert lib.mp_spmm_plan_bytes(1000, 10000, (C.c_int32 * 4)(32, 4, 1024, 256), C.byref(nb)) == 1   # seg_cost too small
    assert lib.mp_spmm_plan_bytes(1000, 10000, (C.c_int32 * 4)(64, 1, 64, 64), C.byref(nb)) == 0 and nb.value > default_bytes
    counts = (C.c_int32 * 8)(10, 0, 3, 1, 4, 320, 1024, 256)
    assert lib.mp_spmm_ws_bytes(counts, 256, 0, 0, C.byref(nb)) == 0
    assert nb.value >= 3 * 256 * 4
    assert lib.mp_spmm_ws_bytes(counts, 256, 2, 0, C.byref(nb)) == 0    # max: values + argmax
    assert nb.value >= 2 * 3 * 256 * 4
    # null pointers are rejected before any launch
    assert lib.mp_spmm_csr_f32(None, None, None, 5, None, counts, None, 4, None, 4, 4, 0, None, 0, 0.0,
                               None, 0, None, None, 0, None) == 1


def test_ba_generator_host():
    from graphgym_amd import graphgen
    u, v = graphgen.ba_undirected_pairs(2000, 5, seed=3)
    assert u.size == 5 * (2000 - 5) and (u > v).all() and v.min() >= 0 and u.max() == 1999
    u2, v2 = graphgen.ba_undirected_pairs(2000, 5, seed=3)
    assert (u == u2).all() and (v == v2).all()                          # deterministic
    deg = np.bincount(np.concatenate([u, v]), minlength=2000)
    assert deg[5:].min() >= 5                                            # every new node made m links
    assert deg.max() > 5 * np.median(deg)                                # heavy tail
    ei = graphgen.ba_edge_index(500, 3, seed=1)
    key = (ei[1] * 500 + ei[0]).numpy()
    assert np.unique(key).size == key.size                               # deduplicated
    rev = (ei[0] * 500 + ei[1]).numpy()
    assert set(key.tolist()) == set(rev.tolist())                        # symmetric


def test_ops_refuse_cpu_tensors():
    import torch
    import graphgym_amd as ga
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(ga.EngineError):
        ga.CSRGraph.from_edge_index(ei, 2)


def test_powerlaw_cluster_generator_host():
    """Holme-Kim growth: same degree budget as BA, but triangles (clustering) like networkx's generator"""
    import networkx as nx
    from graphgym_amd import graphgen
    u, v = graphgen.powerlaw_cluster_undirected_pairs(4000, 4, 0.5, seed=2)
    assert (u > v).all() and v.min() >= 0 and u.max() == 3999 and u.size <= 4 * (4000 - 4)
    pairs = set(zip(u.tolist(), v.tolist()))
    assert len(pairs) == u.size                                          # links of one node are distinct
    G = nx.Graph(); G.add_edges_from(pairs)
    ref = nx.powerlaw_cluster_graph(4000, 4, 0.5, seed=2)
    ba_u, ba_v = graphgen.ba_undirected_pairs(4000, 4, seed=2)
    B = nx.Graph(); B.add_edges_from(zip(ba_u.tolist(), ba_v.tolist()))
    c, cref, cba = nx.average_clustering(G), nx.average_clustering(ref), nx.average_clustering(B)
    assert cba < 0.05 < c and abs(c - cref) < 0.5 * cref                 # triangle closing is really there
    u2, v2 = graphgen.powerlaw_cluster_undirected_pairs(4000, 4, 0.5, seed=2)
    assert (u == u2).all() and (v == v2).all()

"""ctypes access to oracle/libspmm_ref.so (the plain-C restatement).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libspmm_ref.so")
_lib = None


def build():
    src = os.path.join(_HERE, "spmm_ref.c")
    if not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
        subprocess.check_call(["make", "-C", _HERE, "libspmm_ref.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def coo_aggregate(row, col, w, x, num_nodes, reduce="sum", want_argmax=False):
    row = np.ascontiguousarray(row, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    d = x.shape[1]
    out = np.empty((num_nodes, d), dtype=np.float32)
    arg = np.empty((num_nodes, d), dtype=np.int64) if want_argmax else None
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib().ref_coo_aggregate_f32(p(row), p(col), p(w), C.c_int64(row.size), p(x), C.c_int64(d),
                                     C.c_int64(num_nodes), {"sum": 0, "add": 0, "mean": 1, "max": 2}[reduce],
                                     p(out), p(arg))
    if rc:
        raise RuntimeError(f"ref_coo_aggregate_f32 failed ({rc})")
    return (out, arg) if want_argmax else out


def gcn_norm(row, col, w, num_nodes, deg_by_col=False):
    row = np.ascontiguousarray(row, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty(row.size, dtype=np.float32)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib().ref_gcn_norm_f32(p(row), p(col), p(w), C.c_int64(row.size), C.c_int64(num_nodes),
                                int(deg_by_col), p(out))
    if rc:
        raise RuntimeError(f"ref_gcn_norm_f32 failed ({rc})")
    return out

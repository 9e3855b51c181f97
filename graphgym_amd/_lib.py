"""ctypes binding of libmpengine.so — the only way Python reaches the HIP kernels.

The prototypes below are a 1:1 transcription of include/mp_engine.h.  There is no
CPU fallback anywhere in this package: if the library is missing, importing it
raises, and ops called on non-HIP tensors raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MP_ENGINE_LIB") or os.path.join(_HERE, "csrc", "libmpengine.so")   # (MP_ENGINE_LIB: a variant build, for A/B studies)

MP_OK = 0
SUM, MEAN, MAX = 0, 1, 2
REDUCE = {"sum": SUM, "add": SUM, "mean": MEAN, "max": MAX}
COO_REMOVE_SELF_LOOPS, COO_ADD_SELF_LOOPS, COO_KEEP_LOOP_WEIGHT, COO_RECT = 1, 2, 4, 8
AXIS_ROW, AXIS_COL = 0, 1
ACT_NONE, ACT_RELU = 0, 1

_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_f32 = C.c_float
_sz = C.c_size_t
_psz = C.POINTER(C.c_size_t)
_pi32 = C.POINTER(C.c_int32)

# name -> (restype, argtypes); order and types follow mp_engine.h
PROTOTYPES = {
    "mp_version": (C.c_int, []),
    "mp_copy_probe_f32": (C.c_int, [_p, _p, _i64, _p]),
    "mp_read_probe_f32": (C.c_int, [_p, _i64, _p, _p]),
    "mp_status_str": (C.c_char_p, [C.c_int]),
    "mp_last_hip_error": (C.c_char_p, []),
    "mp_probe_copy_ms": (C.c_int, [_p, _p, _sz, _i32, C.POINTER(C.c_float), _p]),
    "mp_probe_gather_ms": (C.c_int, [_p, _sz, _p, _sz, _i32, _i32, C.POINTER(C.c_float), _p]),
    "mp_csr_from_coo_ws_bytes": (C.c_int, [_i64, _i64, _psz]),
    "mp_csr_from_coo": (C.c_int, [_p, _p, _p, _i64, _i64, C.c_int, _f32, _p, _p, _p, _p, _p, _sz, _p]),
    "mp_check_edge_index": (C.c_int, [_p, _p, _i64, _i64, _p, _p]),
    "mp_csr_row_ids": (C.c_int, [_p, _i64, _i64, _p, _p]),
    "mp_csr_is_symmetric": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "mp_csr_transpose_ws_bytes": (C.c_int, [_i64, _i64, _psz]),
    "mp_csr_transpose": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "mp_csr_degree": (C.c_int, [_p, _p, _p, _i64, _i64, C.c_int, _p, _p]),
    "mp_gcn_norm_edges": (C.c_int, [_p, _p, _p, _i64, _i64, C.c_int, _p, _p, _p]),
    "mp_stream_create_cu_mask": (C.c_int, [_p, C.c_int, _p]),
    "mp_stream_destroy": (C.c_int, [_p]),
    "mp_csr_scale_f32": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    "mp_mark_id_sources": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _p, _p]),
    "mp_spmm_plan_bytes": (C.c_int, [_i64, _i64, _pi32, _psz]),
    "mp_spmm_plan_build": (C.c_int, [_p, _i64, _i64, _pi32, _p, _sz, _pi32, _p]),
    "mp_spmm_ws_bytes": (C.c_int, [_pi32, _i32, C.c_int, C.c_int, _psz]),
    "mp_spmm_csr_f32": (C.c_int, [_p, _p, _p, _i64, _p, _pi32, _p, _i64, _p, _i64, _i32, C.c_int,
                                  _p, _i64, _f32, _p, C.c_int, _p, _p, _sz, _p]),
    "mp_spmm_csr_epilogue_f32": (C.c_int, [_p, _p, _p, _i64, _p, _pi32, _p, _i64, _p, _i64, _i32, C.c_int,
                                           _p, _i64, _f32, _p, _p, C.c_int, C.c_int, _f32, _p, _sz, _p]),
    "mp_idgnn_agg_f32": (C.c_int, [_p, _p, _p, _i64, _p, _pi32, _p, _i64, _p, _i64, _p, _i64, _i32,
                                   _p, _sz, _p]),
    "mp_spmm_max_bwd_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _p, _i64, _p]),
    "mp_bn_ws_bytes": (C.c_int, [_i64, _i32, _psz]),
    "mp_bn_train_fwd_f32": (C.c_int, [_p, _i64, _i64, _i32, _p, _p, _f32, C.c_int, _p, _i64, _p, _p, _p, _p, _sz, _p]),
    "mp_bn_train_bwd_f32": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _p, _i64, _p, _p, _p,
                                      _sz, _p]),
    "mp_bn_train_bwd_relu_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _i64, _p, _p, _p,
                                           _sz, _p]),
    "mp_agg_rows_tiles_f32": (C.c_int, [_p, _p, _p, _i64, C.c_int, _p, _i64, _i32, _p, _i64, C.c_float, _p, _i64, _p]),
    "mp_agg_dense_f32": (C.c_int, [_p, _p, _p, _i64, C.c_int, _p, _i64, _i32, _p, _i64, C.c_float, _p, _i64, _i32, _p, C.c_int,
                                   _p, _p, _i64, _p, _i64, _p, _p]),
    "mp_agg_dense_add_f32": (C.c_int, [_p, _p, _p, _i64, C.c_int, _p, _i64, _i32, _p, _i64, C.c_float, _p, _i64, _i32, _p,
                                       C.c_int, _p, _p, _i64, _p, _i64, _p, _p, _i64, _p]),
    "mp_id_fixup_f32": (C.c_int, [_p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i32, C.c_int, _p]),
    "mp_idgnn_agg_tiles_f32": (C.c_int, [_p, _p, _p, _i64, _p, _i64, _i32, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _p,
                                         _i64, _p]),
    "mp_id_rows_f32": (C.c_int, [_p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i32, _p]),
    "mp_dense_fused_f32": (C.c_int, [_p, _i64, _p, _p, _i64, _p, _p, C.c_int, _p, _i64, _i64, _i32, _i32, _p]),
    "mp_dense_x3_f32": (C.c_int, [_p, _i64, _p, _p, _i32, _p, _i64, _i64, _i32, _i32, _p]),
    "mp_split_w_bf16x3": (C.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p]),
    "mp_dense_wgrad_ws_bytes": (C.c_int, [_i64, _i32, _i32, _psz]),
    "mp_dense_wgrad_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _sz, _p]),
    "mp_dense_wgrad_relu_f32": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _sz, _p]),
    "mp_softmax_ce_rows_f32": (C.c_int, [_p, _i64, _i64, _p, _p, _i64, _i32, _p, _p]),
    "mp_softmax_ce_bwd_f32": (C.c_int, [_p, _i64, _i64, _p, _p, _i64, _i32, _p, _f32, _p, _i64, _p]),
    "mp_rows_gather_f32": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _i64, _p]),
    "mp_rows_scatter_add_f32": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _i64, _p]),
    "mp_sddmm_dot_f32": (C.c_int, [_p, _p, _i64, _i64, _p, _i64, _p, _i64, _i32, _i32, _f32, _p, _p]),
    "mp_sddmm_dot_stream_f32": (C.c_int, [_p, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _f32, _p, _p]),
    "mp_sddmm_add_f32": (C.c_int, [_p, _p, _i64, _i64, _p, _p, _f32, _p, _p]),
    "mp_gat_alpha_f32": (C.c_int, [_p, _p, _i64, _i64, _i32, _p, _p, _f32, _p, _p]),
    "mp_csr_row_softmax_f32": (C.c_int, [_p, _i64, _i32, _p, _p, _p]),
    "mp_csr_row_softmax_bwd_f32": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p]),
    "mp_sddmm_grad_f32": (C.c_int, [_p, _p, _i64, _i64, _p, _i64, _p, _i64, _i32, _i32, _p, _p]),
    "mp_spmm_csr_heads_f32": (C.c_int, [_p, _p, _p, _i64, _p, _pi32, _i32, _p, _i64, _p, _i64, _i32, _p, _sz, _p]),
    "mp_spmm_heads_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _p, _i64, _p, _i64, _i32, _p]),
    "mp_ego_expand": (C.c_int, [_p, _p, _i64, _p, _i64, _i32, _i32, _p, _p, _p, _p, _p]),
    "mp_lpt_partition_host": (C.c_int, [_p, _i64, _i32, _p]),
    "mp_gen_ba_edges_host": (C.c_int, [_i64, _i32, C.c_uint64, _p, _p, _p]),
    "mp_gen_powerlaw_cluster_edges_host": (C.c_int, [_i64, _i32, C.c_double, C.c_uint64, _p, _p, _p]),
}

# callback types of mp_ego_expand (mp_engine.h: mp_alloc_fn, mp_free_fn, mp_ego_result_t)
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p)
FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class EgoResult(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_edges", C.c_int64), ("src", C.c_void_p), ("dst", C.c_void_p),
                ("orig", C.c_void_p), ("ego_of", C.c_void_p), ("rowptr", C.c_void_p), ("col", C.c_void_p),
                ("eid", C.c_void_p), ("nnz", C.c_int64), ("candidates", C.c_int64),
                ("scratch_peak_bytes", C.c_size_t)]


_lib = None


class EngineError(RuntimeError):
    pass


def lib():
    """Load libmpengine.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m graphgym_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(h, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(status, what=""):
    if status != MP_OK:
        L = lib()
        msg = L.mp_status_str(status).decode()
        if status == 4:
            msg += " — " + L.mp_last_hip_error().decode()
        raise EngineError(f"{what or 'mp_engine'}: {msg} (status {status})")


def ptr(t):
    """device (or host) address of a tensor's first element, None -> NULL"""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())

/*
 * mp_engine.h — C ABI of the MI355X message-passing engine (libmpengine.so).
 *
 * This is the drop-in boundary for GraphGym's neighbour-aggregation hot path.
 * The reference has no native layer (it is pure Python over torch-scatter /
 * TensorFlow ops), so every entry point below cites the *Python* call site
 * whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers + sizes, no torch / C++ types; all pointers are DEVICE
 *     pointers unless the parameter name ends in `_host`;
 *   - every function returns an mp_status (0 = ok); nothing throws;
 *   - the caller owns every buffer (outputs and workspaces); nothing is
 *     allocated or freed inside the library;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*);
 *     functions do not synchronise unless their comment says so;
 *   - matrices are row-major fp32 with an explicit leading dimension (in
 *     elements); indices inside the engine are int32, the COO boundary takes
 *     the int64 `edge_index` tensors GraphGym batches carry;
 *   - row r of a CSR holds the in-edges of DESTINATION node r; `col` holds
 *     SOURCE node ids (SparseAdj: edge_index[0] = row, [1] = col,
 *     sparse_adj.py:50-56; PyG flow source_to_target: edge_index[1] = i).
 */
#ifndef MP_ENGINE_H
#define MP_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mp_stream_t; /* hipStream_t */

enum mp_status {
  MP_OK = 0,
  MP_ERR_INVALID_ARG = 1, /* null pointer, negative size, bad enum */
  MP_ERR_UNSUPPORTED = 2, /* e.g. nnz >= 2^31 */
  MP_ERR_WORKSPACE = 3,   /* workspace too small */
  MP_ERR_HIP = 4,         /* a HIP runtime call failed; see mp_last_hip_error */
  MP_ERR_ALIGNMENT = 5    /* pointer / leading dimension not aligned as required */
};

enum mp_reduce { MP_SUM = 0, MP_MEAN = 1, MP_MAX = 2 };

/* mp_csr_from_coo flags */
enum mp_coo_flags {
  MP_COO_REMOVE_SELF_LOOPS = 1, /* torch_geometric.utils.remove_self_loops (idconv.py:302,370) */
  MP_COO_ADD_SELF_LOOPS = 2,    /* append one (i,i) entry per node: tfg add_self_loop_edge via
                                   SparseAdj.add_self_loop (sparse_adj.py:58-63, TfgIDLayer.py:298,547);
                                   PyG add_self_loops (idconv.py:303) */
  MP_COO_KEEP_LOOP_WEIGHT = 4,  /* with REMOVE|ADD: an existing loop's weight replaces `fill` for
                                   that node = PyG add_remaining_self_loops (idconv.py:52-53,140-141,232-233) */
  MP_COO_RECT = 8               /* a rectangular operator (pooling.py:12-33: rows = graphs, columns = nodes): src ids
                                   may be any value below 2^32, not only [0, N).  Without it both ids share
                                   id_bits(N) bits of the sort key and the sort runs over twice that (N = 6e5: 40
                                   bits instead of 52); not with the self-loop flags */
};

/* degree axis for mp_csr_degree / mp_gcn_norm_edges */
enum mp_axis {
  MP_AXIS_ROW = 0, /* by destination: SparseAdj.reduce_sum(axis=-1) (sparse_adj.py:65-85, TfgIDLayer.py:549) */
  MP_AXIS_COL = 1  /* by source: scatter_add(edge_weight, edge_index[0]) (idconv.py:56,144) */
};

/* epilogue activation fused into the aggregation's row flush (K15) */
enum mp_act { MP_ACT_NONE = 0, MP_ACT_RELU = 1 };

int mp_version(void);
/* streaming device copy of n floats (n % 4 == 0, 16-byte aligned), 16 B per lane: the measured-bandwidth
 * yardstick bench.py reports beside the nominal HBM peak */
int mp_copy_probe_f32(const float* src, float* dst, int64_t n, mp_stream_t stream);
/* read-only stream of n floats; sink must hold 256 * 8 * 256 floats (one per launched lane) */
int mp_read_probe_f32(const float* src, int64_t n, float* sink, mp_stream_t stream);
/* a stream restricted to the compute units set in mask (n_words x 32 bits; hipExtStreamCreateWithCUMask):
 * lets a caller run the HBM-bound aggregation and the MFMA-bound transform side by side on disjoint CUs */
int mp_stream_create_cu_mask(const uint32_t* mask, int n_words, mp_stream_t* stream);
int mp_stream_destroy(mp_stream_t stream);
const char* mp_status_str(int status);
/* text of the last failing HIP call on this thread ("" if none) */
const char* mp_last_hip_error(void);

/* ------------------------------------------------------------------ *
 * Placement probe (no counterpart in the reference: this belongs to the  *
 * path's data layout in HBM).  On MI355X a launch that reads X and       *
 * writes Y is up to ~13 % slower depending on which physical memory      *
 * backs the two (high address bits are hashed into the DRAM bank /       *
 * channel selection).  The library allocates nothing: outputs are the    *
 * caller's buffers.  The host side (graphgym_amd/placement.py) times a   *
 * copy between sample chunks of what a launch reads and a candidate      *
 * output buffer with this entry point and re-allocates on conflict.      *
 * ------------------------------------------------------------------ */
/* timed streaming copy (16 B per lane, non-temporal stores): one untimed launch, then `reps` launches between
 * two events; SYNCHRONISES `stream`; *ms_host = mean milliseconds per launch */
int mp_probe_copy_ms(const void* src, void* dst, size_t bytes, int32_t reps, float* ms_host, mp_stream_t stream);
/* timed gather probe — the access pattern of the aggregation itself: each of the dst_bytes / 1024 rows of dst is the
 * sum of `fan` (1..64) pseudo-random 1 KiB rows taken from ALL of [src, src + src_bytes); one untimed launch, then
 * `reps`; SYNCHRONISES `stream`.  Shows the placement effect at full size (+12-13 % good vs bad position of dst). */
int mp_probe_gather_ms(const void* src, size_t src_bytes, void* dst, size_t dst_bytes, int32_t fan, int32_t reps,
                       float* ms_host, mp_stream_t stream);

/* ------------------------------------------------------------------ *
 * Graph construction: COO edge list -> destination-sorted CSR         *
 * replaces: SparseAdj.__init__/add_self_loop (sparse_adj.py:18-63),   *
 *   add_remaining_self_loops / remove_self_loops / add_self_loops     *
 *   call sites in idconv.py:52,140,232,302-304,370                    *
 * ------------------------------------------------------------------ */

/* bytes of workspace mp_csr_from_coo needs for E input edges, N nodes */
int mp_csr_from_coo_ws_bytes(int64_t E, int64_t N, size_t* bytes_host);

/*
 * dst/src: [E] int64 node ids in [0,N) (src: any id below 2^32 with MP_COO_RECT).  w: [E] fp32 or NULL (= all ones).
 * Outputs (capacity E+N entries each; nnz = rowptr[N] afterwards):
 *   rowptr [N+1] int32, col [cap] int32 (sorted by (row, col, input order)),
 *   val [cap] fp32 (may be NULL when w == NULL and fill == 1: unweighted),
 *   eid [cap] int32: input position of each entry, or -1-i for an inserted
 *   loop of node i (may be NULL).
 * Out-of-range node ids make the result undefined; validate on the host
 * (mp_check_edge_index) in debug paths.
 */
int mp_csr_from_coo(const int64_t* dst, const int64_t* src, const float* w,
                    int64_t E, int64_t N, int flags, float fill,
                    int32_t* rowptr, int32_t* col, float* val, int32_t* eid,
                    void* ws, size_t ws_bytes, mp_stream_t stream);

/* counts entries with dst or src outside [0,N) into *bad (device int32) */
int mp_check_edge_index(const int64_t* dst, const int64_t* src, int64_t E,
                        int64_t N, int32_t* bad, mp_stream_t stream);

/* row id of every stored entry: row_of[e] = r for rowptr[r] <= e < rowptr[r+1] */
int mp_csr_row_ids(const int32_t* rowptr, int64_t N, int64_t nnz,
                   int32_t* row_of, mp_stream_t stream);

/* flag[0] (device) = 0 if the stored square operator equals its transpose — every entry (r, c, v) has its mirror
 * (c, r, v) with the same bits and no (r, c) is stored twice — else 1.  The undirected graphs of the reference's
 * datasets (both directions stored, loader.py / transform.py:11-38) are their own transpose: the backward pass
 * dL/dX = A^T dL/dY then needs no second CSR (mp_csr_transpose: a sort of all entries). */
int mp_csr_is_symmetric(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int64_t nnz,
                        int32_t* flag, mp_stream_t stream);

/*
 * Transpose (CSR of A^T, i.e. out-edges by source) for the backward pass
 * dL/dX = A^T dL/dY (autodiff of gather + unsorted_segment_sum,
 * sparse_adj.py:91-97; PyG propagate's backward).
 * pos[k] = index in the source CSR of transposed entry k (so per-edge values
 * such as attention coefficients can be permuted with one gather).
 */
/* n_rows x n_cols source (square graphs: both N; pooling operators are rectangular);
 * the workspace query takes N = n_cols */
int mp_csr_transpose_ws_bytes(int64_t nnz, int64_t N, size_t* bytes_host);
int mp_csr_transpose(const int32_t* rowptr, const int32_t* col, const float* val,
                     int64_t n_rows, int64_t n_cols, int64_t nnz,
                     int32_t* t_rowptr, int32_t* t_col, float* t_val, int32_t* pos,
                     void* ws, size_t ws_bytes, mp_stream_t stream);

/* weighted degree: deg[i] = sum of val over row i (axis ROW) or column i (axis COL);
 * val == NULL counts entries.  (K6: sparse_adj.py:84-85, idconv.py:56,144)
 * AXIS_ROW is a fixed-order (Kahan) row sum: bitwise reproducible.  AXIS_COL on a WEIGHTED operator adds with float
 * atomics (order-dependent in the last ulp; exact for val == NULL): for reproducible by-source degrees run AXIS_ROW on
 * the transposed CSR (mp_csr_transpose), as graphgym_amd.graph.CSRGraph.degree('col') does. */
int mp_csr_degree(const int32_t* rowptr, const int32_t* col, const float* val,
                  int64_t N, int64_t nnz, int axis, float* deg, mp_stream_t stream);

/*
 * GCN symmetric normalisation of the stored entries (K6-K8):
 *   dinv = deg^-1/2 with inf/nan -> 0 ; val_out[e] = dinv[row] * val[e] * dinv[col]
 * replaces gcn_norm_adj (TfgIDLayer.py:528-566) and GCNIDConvLayer.norm
 * (idconv.py:132-148).  dinv_out [N] may be NULL.  val may be NULL (ones).
 */
int mp_gcn_norm_edges(const int32_t* rowptr, const int32_t* col, const float* val,
                      int64_t N, int64_t nnz, int deg_axis,
                      float* val_out, float* dinv_out, mp_stream_t stream);

/* diagonal scalings of the stored entries (A3): val_out[e] = row_scale[row] * val[e] * col_scale[col];
 * either scale NULL = ones.  diag @ A = SparseAdj.rmatmul_diag / diag_sparse_matmul (sparse_adj.py:116-119,
 * sparse_ops.py:11-12); A @ diag = matmul_diag / sparse_diag_matmul (sparse_adj.py:110-113, sparse_ops.py:6-7) */
int mp_csr_scale_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int64_t nnz,
                     const float* row_scale, const float* col_scale, float* val_out, mp_stream_t stream);

/* mark entries whose SOURCE is an identity node: col_out[e] = col[e] | 0x80000000
 * when is_id[col[e]] != 0.  is_id: [N] uint8 scratch filled from id_index. */
int mp_mark_id_sources(const int32_t* col, int64_t nnz, const int64_t* id_index,
                       int64_t n_id, int64_t N, uint8_t* is_id, int32_t* col_out,
                       mp_stream_t stream);

/* ------------------------------------------------------------------ *
 * Aggregation plan: nnz-balanced row segments + split of hub rows     *
 * (no counterpart in the reference; built once per graph and cached)  *
 * ------------------------------------------------------------------ */
/* cfg_host: {seg_cost, row_cost, hub_deg, piece_edges} or NULL for the defaults {320, 4, 1024, 256}: a segment is a
 * run of whole rows of cost ~seg_cost (1 per stored entry + row_cost per row); rows with more than hub_deg entries
 * are split into pieces of piece_edges.  The configuration travels with the plan (counts_host[5..7]); there is no
 * process-wide state. */
int mp_spmm_plan_bytes(int64_t N, int64_t nnz, const int32_t* cfg_host, size_t* bytes_host);
/* counts_host[8] <- {n_seg, n_hub, n_piece, cap_hub, cap_piece, seg_cost, hub_deg,
 * piece_edges}; SYNCHRONISES `stream` (once per graph) */
int mp_spmm_plan_build(const int32_t* rowptr, int64_t N, int64_t nnz, const int32_t* cfg_host,
                       int32_t* plan, size_t plan_bytes, int32_t* counts_host,
                       mp_stream_t stream);
/* workspace bytes for an aggregation of width d with this plan */
int mp_spmm_ws_bytes(const int32_t* counts_host, int32_t d, int reduce,
                     int two_branch, size_t* bytes_host);

/* ------------------------------------------------------------------ *
 * The hot path: Y = reduce_{e in row r} val[e] * X[col[e], :]         *
 * replaces SparseAdj.matmul (sparse_adj.py:91-97) = tf.gather * w ->  *
 * unsorted_segment_sum; PyG MessagePassing.propagate + torch_scatter  *
 * (idconv.py:89,177,235,315,371; generalconv.py:86); mean_reducer     *
 * (TfgIDLayer.py:98); aggr='max' (generalconv.py:18).                 *
 * ------------------------------------------------------------------ */
/*
 * X [n_src, d] (ldx), Y [N, d] (ldy).  val NULL = unweighted.
 * reduce: MP_SUM | MP_MEAN (sum / row entry count; empty row -> 0) |
 *         MP_MAX (empty row -> 0; argmax [N, d] int32 receives the CSR entry
 *         index of the winner, -1 for empty rows; may be NULL).
 * Epilogue, applied per output row before the store (K15, K17):
 *   y = act( y + self_scale * S[r, :] + bias )
 *   S (lds) NULL = no self term (GIN's (1+eps)*x_i: TfgIDLayer.py:157-159,
 *   idconv.py:371); bias [d] NULL = none.
 */
int mp_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                    int64_t N, const int32_t* plan, const int32_t* counts_host,
                    const float* X, int64_t ldx, float* Y, int64_t ldy, int32_t d,
                    int reduce, const float* S, int64_t lds, float self_scale,
                    const float* bias, int act, int32_t* argmax,
                    void* ws, size_t ws_bytes, mp_stream_t stream);

/*
 * Inference-time fusion of the layer's post-ops (graphgym/models/layer.py:26-47,
 * gnn.py:79-80) into the row flush:
 *   y = act( (reduce + self_scale * S[r]) * col_scale + col_shift ) ; optionally y /= max(||y||_2, l2_eps)
 * col_scale / col_shift [d] fold conv bias + BatchNorm1d in eval mode (gamma / sqrt(var + eps), ...).
 * l2_normalize needs the whole row in one wave: d <= 256 (MP_ERR_UNSUPPORTED otherwise).
 */
int mp_spmm_csr_epilogue_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                             int64_t N, const int32_t* plan, const int32_t* counts_host,
                             const float* X, int64_t ldx, float* Y, int64_t ldy, int32_t d,
                             int reduce, const float* S, int64_t lds, float self_scale,
                             const float* col_scale, const float* col_shift, int act,
                             int l2_normalize, float l2_eps,
                             void* ws, size_t ws_bytes, mp_stream_t stream);

/*
 * ID-GNN two-branch aggregation in one pass over the edges (A7):
 *   P[r,:] = sum_e val[e] * X[col[e],:]
 *   Q[r,:] = sum_{e : source is an identity node} val[e] * X[col[e],:]
 * so that  P W + Q W_id  ==  A_hat (X W + S X W_id)  of gcn_id
 * (TfgIDLayer.py:510-517) / GCNIDConvLayer.forward (idconv.py:152-177).
 * col must carry the identity mark of mp_mark_id_sources.
 */
int mp_idgnn_agg_f32(const int32_t* rowptr, const int32_t* col_marked, const float* val,
                     int64_t N, const int32_t* plan, const int32_t* counts_host,
                     const float* X, int64_t ldx, float* P, int64_t ldp,
                     float* Q, int64_t ldq, int32_t d,
                     void* ws, size_t ws_bytes, mp_stream_t stream);

/* backward of MP_MAX: dX[col[e], c] += val[e] * dY[r,c] with e = argmax[r,c] >= 0
 * (dX pre-zeroed by the caller; val NULL = ones) */
int mp_spmm_max_bwd_f32(const int32_t* col, const float* val, const int32_t* argmax,
                        const float* dY, int64_t ldy, int64_t N, int32_t d,
                        float* dX, int64_t ldx, mp_stream_t stream);

/* ------------------------------------------------------------------ *
 * BatchNorm1d over the node axis in training mode with the activation  *
 * fused (K20 / SURVEY §8f rank 3): graphgym/models/layer.py:26-35,      *
 * keras BatchNormalization of main_zd.py:181-186.                       *
 *   fwd: mean / biased var over the N rows, y = act((x - mean) * invstd *
 *        * gamma + beta); var_unbiased feeds the running statistics.    *
 *   bwd: g = dy * [y > 0] (y NULL = no activation); dgamma, dbeta, dx.  *
 * Column sums are combined in double precision in a fixed order.        *
 * ------------------------------------------------------------------ */
int mp_bn_ws_bytes(int64_t N, int32_t d, size_t* bytes_host);
int mp_bn_train_fwd_f32(const float* x, int64_t ldx, int64_t N, int32_t d,
                        const float* gamma, const float* beta, float eps, int relu,
                        float* y, int64_t ldy, float* mean, float* invstd, float* var_unbiased,
                        void* ws, size_t ws_bytes, mp_stream_t stream);
int mp_bn_train_bwd_f32(const float* dy, int64_t lddy, const float* y, int64_t ldy,
                        const float* x, int64_t ldx, int64_t N, int32_t d,
                        const float* gamma, const float* mean, const float* invstd,
                        float* dx, int64_t lddx, float* dgamma, float* dbeta,
                        void* ws, size_t ws_bytes, mp_stream_t stream);
/* The backward of BatchNorm + ReLU without the forward's output: the mask [y > 0] is recomputed from x —
 * y = relu(fmaf(x - mean, float(gamma * invstd), beta)), the forward's own expression on the forward's own operands,
 * so the mask is the forward's bit for bit — and the two passes read dy and x only (50 instead of 70 GB at
 * 10^7 x 256).  gamma / beta NULL = 1 / 0. */
int mp_bn_train_bwd_relu_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t N, int32_t d,
                             const float* gamma, const float* beta, const float* mean, const float* invstd,
                             float* dx, int64_t lddx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                             mp_stream_t stream);

/* ------------------------------------------------------------------ *
 * Aggregate -> transform in one kernel                                *
 * ------------------------------------------------------------------ */
/* out[N, d_out] = act( (A X + self_scale * S) W + bias ): SparseAdj.matmul (sparse_adj.py:91-97) followed by
 * the layer's kernel product (gcn_id in the aggregate-first order, TfgIDLayer.py:510-523; GIN's
 * (1 + eps) x + sum -> first Linear, idconv.py:371-399, TfgIDLayer.py:447-456) without the [N, F] intermediate
 * going through HBM: a workgroup reduces a 32-row tile into LDS and multiplies it by W on the matrix cores
 * while the other workgroups of the compute unit gather.  reduce = MP_SUM | MP_MEAN (mean: rows divided by
 * their entry count, S must be NULL); stored values (val NULL = ones).
 * F must be 64, 128, 256 or 512 and d_out even (F = 512: two K halves over the same row tile, d_out <= 512;
 * MP_ERR_UNSUPPORTED otherwise: use mp_spmm_csr_f32 + mp_dense_fused_f32).  P (optional, [N, F]) receives the
 * aggregated rows (kept for the weight gradient).  defer_act (optional, [N] uint8): rows with a nonzero flag are
 * stored WITHOUT the activation (mp_id_fixup_f32 finishes them).  A sign-bit identity mark on col
 * (mp_mark_id_sources) is ignored.  No plan, no workspace, no process-wide state; bitwise reproducible.
 * W_split (optional): W split three ways into bf16 — [3][F / 8][d_out][8] bf16 (element [s][k / 8][c][k % 8] =
 * plane s of W[k][c]), plane s holding bf16(W - sum of the planes before it) — switches the product to the bf16 matrix pipe with all six significant
 * cross terms (fp32-accurate to ~2^-24 of the result; 3/8 of the MFMA cycles of the exact-fp32 form, which gfx950 runs at
 * 1/16 of the bf16 rate).  NULL = v_mfma_f32_32x32x2_f32 on W itself. */
int mp_agg_dense_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int reduce,
                     const float* X, int64_t ldx, int32_t F, const float* S, int64_t lds, float self_scale, const float* W,
                     int64_t ldw, int32_t d_out, const float* bias, int act, const uint8_t* defer_act, float* P,
                     int64_t ldp, float* out, int64_t ldo, const void* W_split, mp_stream_t stream);

/* The same with a residual added before the activation: out = act((A X + self_scale * S) W + bias + R), R [N, d_out]
 * (ldr; 8-byte aligned rows; may be `out` itself: every element is read before it is written, by the same lane).  The
 * input gradient of a concatenating layer — dx = g_self W_self^T + (A^T g_nbr) W_nbr^T (MeanGraphSage,
 * TfgIDLayer.py:100-117) — is this call with R = g_self W_self^T, instead of an extra read-read-write pass over
 * [N, d]. */
int mp_agg_dense_add_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int reduce,
                         const float* X, int64_t ldx, int32_t F, const float* S, int64_t lds, float self_scale,
                         const float* W, int64_t ldw, int32_t d_out, const float* bias, int act,
                         const uint8_t* defer_act, float* P, int64_t ldp, float* out, int64_t ldo, const void* W_split,
                         const float* R, int64_t ldr, mp_stream_t stream);

/* The aggregation ALONE on the same workgroup structure (round 3): out[N, F] = reduce_j w_ij X[j] (+ self_scale * S) —
 * SparseAdj.matmul (sparse_adj.py:91-97), the same contract as mp_spmm_csr_f32 with reduce = MP_SUM | MP_MEAN | MP_MAX
 * (max: values only, S must be NULL, rows without entries are 0; bit for bit the rows of mp_spmm_csr_f32, whose argmax
 * the backward pass uses) and no epilogue.  Four waves of a workgroup gather 64-row tiles into LDS while four others store the finished tile with
 * full-line non-temporal stores; tiles are drawn from a counter (no plan, no workspace).  F = 128, 256 or 512, 16-byte
 * aligned rows, rows of at most 2^18 stored entries (longer rows: mp_spmm_csr_f32, whose plan spreads them over many
 * waves); MP_ERR_UNSUPPORTED otherwise.  Faster than the plan-based kernel at these widths (19.6 against 20.6 ms at
 * 10^7 rows, 1.1e8 entries, F = 256).  Bitwise reproducible; a row cut between two waves of a tile is summed in wave
 * order, so results agree with mp_spmm_csr_f32 to fp32 rounding of the row sum, not bit for bit. */
int mp_agg_rows_tiles_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int reduce,
                          const float* X, int64_t ldx, int32_t F, const float* S, int64_t lds, float self_scale,
                          float* out, int64_t ldo, mp_stream_t stream);

/* The identity branch of the ID layers on top of mp_agg_dense_f32: out = act(A (X W + S X W_id) + b)
 * (gcn_id, TfgIDLayer.py:510-523; GCNIDConvLayer.forward, idconv.py:150-177) equals
 * act((A X) W + b + A_id Z) with Z = X[id] W_id (n_id rows: a small product the caller makes) and A_id the stored
 * entries whose source is an identity node.  mp_agg_dense_f32 is run with defer_act[r] != 0 on the rows that own
 * such an entry (they are stored without the activation), then this call finishes exactly those rows:
 *   out[rows[k], :] = act(out[rows[k], :] + sum_{e in [crp[k], crp[k+1])} val[e] * Z[slot[e], :])
 * rows [n_rows] ascending row ids, crp [n_rows + 1], slot [crp[n_rows]] = row of Z, val NULL = ones. */
int mp_id_fixup_f32(const int32_t* rows, const int32_t* crp, const int32_t* slot, const float* val, int64_t n_rows,
                    const float* Z, int64_t ldz, float* out, int64_t ldo, int32_t d, int act, mp_stream_t stream);

/* The two-branch aggregation of gcn_id (TfgIDLayer.py:510-517; the contract of mp_idgnn_agg_f32: P = A X, Q = A S X
 * with S selecting the identity nodes' rows) on the tile structure of mp_agg_rows_tiles_f32:
 *   P = A X (col: plain or sign-marked indices) and, in the same pass, the rows of Q written as zeros — except the rows
 *   with id_rows[r] != 0 (the rows that own an entry from an identity node), which a small kernel launched behind the
 *   tile kernel writes as
 *   Q[rows[k], :] = sum_{e in [crp[k], crp[k+1])} val_id[e] * Z[slot[e], :]  with Z = X[id]
 *   (rows / crp / slot / val_id: as for mp_id_fixup_f32; id_rows [N] uint8 = 1 exactly on `rows`).
 * Every row of Q is written exactly once; the identity entries (1 % of the operator at 1 % identity nodes) are gathered
 * from an [n_id, F] matrix.  n_rows = 0: Q = 0.  F = 128, 256 or 512, 16-byte aligned rows; MP_ERR_UNSUPPORTED otherwise.
 * mp_id_rows_f32 is the small kernel alone (the row's old contents are not read). */
int mp_idgnn_agg_tiles_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, const float* X,
                           int64_t ldx, int32_t F, const uint8_t* id_rows, const int32_t* rows, const int32_t* crp,
                           const int32_t* slot, const float* val_id, int64_t n_rows, const float* Z, int64_t ldz, float* P,
                           int64_t ldp, float* Q, int64_t ldq, mp_stream_t stream);
int mp_id_rows_f32(const int32_t* rows, const int32_t* crp, const int32_t* slot, const float* val, int64_t n_rows,
                   const float* Z, int64_t ldz, float* out, int64_t ldo, int32_t d, mp_stream_t stream);

/* ------------------------------------------------------------------ *
 * Dense transform after the aggregation, fused (K10 / K11 / K15):       *
 *   out = act( P @ W [+ Q @ W_id] + bias )                              *
 * P, Q [M, F] (ldp, ldq), W, W_id [F, d] row-major contiguous, bias [d] *
 * or NULL, out [M, d] (ldo).  Q == W_id == NULL gives the single        *
 * product.  fp32 on the matrix cores (v_mfma_f32_32x32x2_f32).          *
 * Replaces x @ kernel, x_id @ kernel_id, scatter-add, + bias, activation *
 * of gcn_id in its post-aggregation form (TfgIDLayer.py:510-523;        *
 * idconv.py:152-184).  Any F, d and leading dimensions; 16-byte loads  *
 * when F % 8 == 0, d % 4 == 0 and rows are 16-byte aligned.             *
 * ------------------------------------------------------------------ */
int mp_dense_fused_f32(const float* P, int64_t ldp, const float* W,
                       const float* Q, int64_t ldq, const float* W_id,
                       const float* bias, int act, float* out, int64_t ldo,
                       int64_t M, int32_t F, int32_t d, mp_stream_t stream);

/* The same transform at its hot shape, as a streaming kernel (dense_x3.hip): out = act(P @ W + bias) with
 * d = 64 / 128 / 256, F % 32 == 0 and F >= 64 (MP_ERR_UNSUPPORTED otherwise: use mp_dense_fused_f32), P and out 16-byte
 * aligned with ldp % 4 == 0 and ldo % 4 == 0 (MP_ERR_ALIGNMENT).  The Linear / kernel product of every layer (x @ kernel, TfgIDLayer.py:510-523;
 * GeneralLayer's Linear, layer.py:136-147; the GIN MLPs, idconv.py:371-399) and, with W_split made from W^T, the
 * input gradient of that product.  W_split = W three-way split into bf16 by mp_split_w_bf16x3 (the layout
 * mp_agg_dense_f32 takes); the product runs on the bf16 matrix pipe with all six significant cross terms
 * (fp32-accurate), P staged by LDS-DMA and read from HBM once. */
int mp_dense_x3_f32(const float* P, int64_t ldp, const void* W_split, const float* bias, int32_t act, float* out,
                    int64_t ldo, int64_t M, int32_t F, int32_t d, mp_stream_t stream);

/* W_split [3][K / 8][n][8] bf16 (6 K n bytes, 16-byte aligned) from fp32 weights: element [s][k / 8][c][k % 8] =
 * plane s of B[k][c], plane s = bf16(B - sum of the planes before it).  trans == 0: B = W, W [K, n] with leading
 * dimension ldw; trans != 0: B = W^T, W [n, K].  K % 8 == 0. */
int mp_split_w_bf16x3(const float* W, int64_t ldw, int32_t K, int32_t n, int32_t trans, void* W_split,
                      mp_stream_t stream);

/* weight gradient of the transform: dW [F, d] = P^T @ G with P [M, F], G [M, d] (backward of K11 under
 * loss.backward(), graphgym/train.py:24).  Split over the node axis into slabs in `ws`
 * (mp_dense_wgrad_ws_bytes), summed in a fixed order: bitwise reproducible.  Any F, d.
 * dbias (optional, [d]) receives the bias gradient sum_m G[m, :] from the same pass over G. */
int mp_dense_wgrad_ws_bytes(int64_t M, int32_t F, int32_t d, size_t* bytes_host);
int mp_dense_wgrad_f32(const float* P, int64_t ldp, const float* G, int64_t ldg, int64_t M,
                       int32_t F, int32_t d, float* dW, float* dbias, void* ws, size_t ws_bytes,
                       mp_stream_t stream);

/* The same pass with the backward of a ReLU epilogue folded in: G is masked by [Y > 0] (Y = the forward output of the
 * transform whose gradient this is) as it is read, dW = P^T (G * [Y > 0]), dbias its column sums, and GM (optional,
 * [M, d], may alias G) receives the masked gradient for the input-gradient launch that follows — the separate
 * elementwise masking pass of loss.backward() (graphgym/train.py:24) disappears. */
int mp_dense_wgrad_relu_f32(const float* P, int64_t ldp, const float* G, int64_t ldg, const float* Y, int64_t ldy,
                            float* GM, int64_t ldgm, int64_t M, int32_t F, int32_t d, float* dW, float* dbias,
                            void* ws, size_t ws_bytes, mp_stream_t stream);

/* ------------------------------------------------------------------ *
 * Loss of the training step that drives the path: softmax cross-entropy  *
 * over the labelled rows (graphgym/loss.py:53-68 masked_logits =          *
 * logits[node_label_index], mean softmax-CE; loss.py:20-37 torch path).   *
 * logits [n_rows, C]; index [n_sel] int64 selects rows of logits (NULL  *
 * = rows 0..n_sel-1); labels [n_sel] int64 in [0, C).  A label outside  *
 * [0, C) (torch's ignore_index is NOT implemented) or an index outside  *
 * [0, n_rows) is never dereferenced: that row's loss is NaN (so the mean *
 * is NaN) and it receives no gradient.                                   *
 *   rows:  row_loss[k] = logsumexp(z_i) - z_i[y_k]                        *
 *   bwd:   dlogits[i, :] (+)= (softmax(z_i) - onehot(y_k)) * gscale[0] * inv_n *
 *          (gscale: device scalar, the upstream gradient).  With an index *
 *          the caller zeroes dlogits first and rows ACCUMULATE, so a row  *
 *          listed k times gets k terms, like F.cross_entropy(logits[index]) *
 * ------------------------------------------------------------------ */
int mp_softmax_ce_rows_f32(const float* logits, int64_t ld, int64_t n_rows, const int64_t* labels,
                           const int64_t* index, int64_t n_sel, int32_t C, float* row_loss, mp_stream_t stream);
int mp_softmax_ce_bwd_f32(const float* logits, int64_t ld, int64_t n_rows, const int64_t* labels,
                          const int64_t* index, int64_t n_sel, int32_t C, const float* gscale, float inv_n,
                          float* dlogits, int64_t ldd, mp_stream_t stream);

/* ------------------------------------------------------------------ *
 * Identity-row update (K10): H[id[k], :] += U[k, :]                   *
 * replaces tf.tensor_scatter_nd_add (TfgIDLayer.py:107,165,330,515)   *
 * and x.index_add_(0, id, x_id) (idconv.py:67,155,251,310,375).       *
 * Duplicate ids accumulate (atomics); GraphGym's ids are unique.      *
 * ------------------------------------------------------------------ */
int mp_rows_gather_f32(const float* X, int64_t ldx, const int64_t* idx, int64_t n,
                       int32_t d, float* out, int64_t ldo, mp_stream_t stream);
int mp_rows_scatter_add_f32(float* H, int64_t ldh, const int64_t* idx, int64_t n,
                            int32_t d, const float* U, int64_t ldu, mp_stream_t stream);

/* ------------------------------------------------------------------ *
 * Attention pieces (K12, K13) for the GAT layers                      *
 * ------------------------------------------------------------------ */
/* dot-product scores: s[e*H+h] = scale * <Qm[row, h-th slice], Km[col, h-th slice]>
 * (TfgIDLayer.py:333-339); H heads split the feature axis evenly. */
int mp_sddmm_dot_f32(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t nnz,
                     const float* Qm, int64_t ldq, const float* Km, int64_t ldk,
                     int32_t d, int32_t heads, float scale, float* s,
                     mp_stream_t stream);
/* the same scores on the entry-balanced kernel: needs row_of (mp_csr_row_ids) instead of rowptr;
 * heads > 1 needs d <= 256 and (d / heads / vector width) a power of two (MP_ERR_UNSUPPORTED
 * otherwise: use mp_sddmm_dot_f32) */
int mp_sddmm_dot_stream_f32(const int32_t* row_of, const int32_t* col, int64_t nnz,
                            const float* A, int64_t lda, const float* B, int64_t ldb,
                            int32_t d, int32_t heads, float scale, float* s, mp_stream_t stream);
/* additive scores: s[e] = leaky_relu(ai[row] + aj[col], slope) (idconv.py:319-326
 * with ai = <z, att[:d]>, aj = <z, att[d:]> precomputed per node) */
int mp_sddmm_add_f32(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t nnz,
                     const float* ai, const float* aj, float slope, float* s,
                     mp_stream_t stream);
/* additive attention coefficients in one pass, all heads (idconv.py:319-327; torch_geometric GATConv [3P]):
 * alpha[e*H+h] = softmax over the entries e of row r of leaky_relu(a_dst[r*H+h] + a_src[col[e]*H+h], slope);
 * the scores are never stored.  a_dst, a_src [N, H]. */
int mp_gat_alpha_f32(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t nnz, int32_t heads,
                     const float* a_dst, const float* a_src, float slope, float* alpha, mp_stream_t stream);
/* softmax over each row's entries, per head: segment_softmax (sparse_adj.py:136-151),
 * torch_geometric.utils.softmax (idconv.py:327).  In-place allowed. */
int mp_csr_row_softmax_f32(const int32_t* rowptr, int64_t N, int32_t heads,
                           const float* s, float* out, mp_stream_t stream);
/* backward of the row softmax: ds = p * (dp - sum_row(p*dp)) */
int mp_csr_row_softmax_bwd_f32(const int32_t* rowptr, int64_t N, int32_t heads,
                               const float* p, const float* dp, float* ds,
                               mp_stream_t stream);
/* per-entry dot: g[e*H+h] = <A[row, slice h], B[col, slice h]> — gradient of an
 * aggregation w.r.t. its edge values (A = dY, B = X) */
int mp_sddmm_grad_f32(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t nnz,
                      const float* A, int64_t lda, const float* B, int64_t ldb,
                      int32_t d, int32_t heads, float* g, mp_stream_t stream);
/* multi-head weighted aggregation on the segment plan — ALL heads in one launch of the hot kernel (full-row loads; a lane
 * applies the weight of the head its columns belong to): Y[r, slice h] = sum_e a[e*H+h] * V[col[e], slice h]
 * (TfgIDLayer.py:340-355 with split_value_heads; idconv.py:317-332).  heads in {1, 2, 4, 8}, d % heads == 0
 * (MP_ERR_UNSUPPORTED otherwise: use mp_spmm_heads_f32).  Workspace as mp_spmm_csr_f32 (mp_spmm_ws_bytes, reduce SUM). */
int mp_spmm_csr_heads_f32(const int32_t* rowptr, const int32_t* col, const float* a, int64_t N, const int32_t* plan,
                          const int32_t* counts_host, int32_t heads, const float* V, int64_t ldv, float* Y, int64_t ldy,
                          int32_t d, void* ws, size_t ws_bytes, mp_stream_t stream);
/* the same without a plan (one wave per row; any head count): fallback for head layouts the plan kernel does not take */
int mp_spmm_heads_f32(const int32_t* rowptr, const int32_t* col, const float* a,
                      int64_t N, int32_t heads, const float* V, int64_t ldv,
                      float* Y, int64_t ldy, int32_t d, mp_stream_t stream);

/* ------------------------------------------------------------------ *
 * Ego-net batcher: graphgym/models/transform.py:11-38 *
 * for a batch of centre nodes, on the device.  The base CSR must be    *
 * symmetric (the reference's graphs are undirected nx.Graph).          *
 *   new ids: centre c -> c (0..B-1); other members of ego c -> fresh    *
 *   consecutive ids in ascending original id, egos one after another    *
 *   (transform.py:24-36); radius > 4 takes the whole graph (:17-18).    *
 * Cost follows the ego nets: the work and every buffer are sized by the *
 * members and candidate neighbours of the level at hand, nothing by N   *
 * (rounds 1-3 kept B x N bitmaps).  Sizes are data, so the engine asks  *
 * the CALLER for memory through a callback (e.g. torch's caching        *
 * allocator) — scratch is handed back level by level, the four outputs  *
 * stay with the caller.  One call; SYNCHRONISES the stream (radius + 2  *
 * reads of counters).                                                   *
 * ------------------------------------------------------------------ */
enum mp_ego_tag { MP_EGO_TAG_SCRATCH = 0, MP_EGO_TAG_EDGES = 1, MP_EGO_TAG_ORIG = 2, MP_EGO_TAG_EGO_OF = 3,
                  MP_EGO_TAG_ROWPTR = 4, MP_EGO_TAG_COL = 5, MP_EGO_TAG_EID = 6 };
/* flags of mp_ego_expand.  The edge list always comes out in the engine's CSR order (rows = destinations by new id,
 * inside a row by new source id), so MP_EGO_CSR can hand back the batch's CSR itself — rowptr / col / eid as
 * mp_csr_from_coo would build them from that list, without the sort.  MP_EGO_CSR_SELF_LOOPS adds one self entry per row
 * to the CSR at its sorted place (eid = -1 - row: mp_csr_from_coo's MP_COO_ADD_SELF_LOOPS on a loop-free input; the
 * base graph must hold no explicit self loops).  Both are ignored when radius > 4. */
enum mp_ego_flags { MP_EGO_CSR = 1, MP_EGO_CSR_SELF_LOOPS = 2 };
/* device memory of `bytes` bytes, 256-byte aligned, usable on the call's stream; NULL = failure (MP_ERR_WORKSPACE) */
typedef void* (*mp_alloc_fn)(size_t bytes, int32_t tag, void* user);
/* a MP_EGO_TAG_SCRATCH block is no longer needed (work that uses it is already enqueued on the call's stream: the
 * allocator must be stream-ordered, as torch's is); may be NULL (scratch then lives until the caller drops it) */
typedef void (*mp_free_fn)(void* ptr, void* user);
typedef struct mp_ego_result {
  int64_t n_nodes;            /* nodes of the expanded graph = sum of the ego nets' sizes                       */
  int64_t n_edges;            /* directed edges (both directions of an undirected edge)                         */
  int64_t* src;               /* [n_edges] new ids; ordered by (dst, src) in the new ids.  src and dst are the two rows */
  int64_t* dst;               /* of ONE [2, n_edges] block (MP_EGO_TAG_EDGES): dst == src + n_edges                 */
  int64_t* orig;              /* [n_nodes] original id of every new node, MP_EGO_TAG_ORIG                       */
  int32_t* ego_of;            /* [n_nodes] owning centre (index into `centres`), MP_EGO_TAG_EGO_OF              */
  int32_t* rowptr;            /* MP_EGO_CSR: [n_nodes + 1], MP_EGO_TAG_ROWPTR (else NULL)                       */
  int32_t* col;               /* MP_EGO_CSR: [nnz] source ids, MP_EGO_TAG_COL                                   */
  int32_t* eid;               /* MP_EGO_CSR: [nnz] position in the edge list, -1 - row for a self entry, TAG_EID */
  int64_t nnz;                /* MP_EGO_CSR: n_edges (+ n_nodes with MP_EGO_CSR_SELF_LOOPS), else 0              */
  int64_t candidates;         /* neighbour entries pushed over all levels                                       */
  size_t scratch_peak_bytes;  /* most scratch held at once                                                      */
} mp_ego_result_t;
int mp_ego_expand(const int32_t* rowptr, const int32_t* col, int64_t N,
                  const int64_t* centres, int64_t n_centres, int32_t radius, int32_t flags,
                  mp_alloc_fn alloc, mp_free_fn release, void* user,
                  mp_ego_result_t* out, mp_stream_t stream);

/* ------------------------------------------------------------------ *
 * Host-side sharding of independent units over ranks (not a device op) *
 * graphgym/loader.py:247-251 (a batch is a disjoint union of graphs),  *
 * graphgym/models/transform.py:24-36 (ego nets are disjoint): units by  *
 * descending cost, each to the least-loaded rank (LPT; ties: lower      *
 * index / lower rank).  owner_host[i] <- rank of unit i.                *
 * ------------------------------------------------------------------ */
int mp_lpt_partition_host(const int64_t* costs_host, int64_t n, int32_t world, int32_t* owner_host);

/* ------------------------------------------------------------------ *
 * Host-side synthetic graph generator (bench / tests; not a device op) *
 * Barabasi-Albert preferential attachment, m links per new node,      *
 * family of datasets/syn_graph.py:42.  Writes m*(n-m) undirected      *
 * (u > v) pairs to HOST arrays; returns the count via *n_edges_host.  *
 * ------------------------------------------------------------------ */
int mp_gen_ba_edges_host(int64_t n, int32_t m, uint64_t seed,
                         int64_t* u_host, int64_t* v_host, int64_t* n_edges_host);

/* Holme-Kim powerlaw-cluster growth (networkx.powerlaw_cluster_graph(n, m, p), datasets/syn_graph.py:42):
 * preferential attachment with probability-p triangle closing.  At most m*(n-m) pairs (u > v). */
int mp_gen_powerlaw_cluster_edges_host(int64_t n, int32_t m, double p, uint64_t seed,
                                       int64_t* u_host, int64_t* v_host, int64_t* n_edges_host);

#ifdef __cplusplus
}
#endif
#endif /* MP_ENGINE_H */

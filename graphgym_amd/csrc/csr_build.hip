// Graph construction on the device: COO edge list -> destination-sorted CSR with
// self-loop edits fused in, transpose, degrees and the GCN edge normalisation.
// Replaces what the reference redoes in Python on every layer call:
//   SparseAdj.__init__/add_self_loop        sparse_adj.py:18-63
//   gcn_norm_adj                            TfgIDLayer.py:528-566
//   GCNIDConvLayer.norm                     idconv.py:132-148
//   add_remaining_self_loops / remove_self_loops / add_self_loops
//                                           idconv.py:52,140,232,302-304,370
// The sort is rocPRIM's device radix sort (rocprim::radix_sort_pairs, called directly): a
// once-per-graph preprocessing step, not the hot path; everything around it is ours.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace mp {

static int id_bits(int64_t N) {
  int b = 1;
  while (((int64_t)1 << b) <= N) ++b;  // ids 0..N need b bits (N itself marks "removed")
  return b;
}

struct CooWs {
  uint64_t* keys_a; uint64_t* keys_b;
  uint32_t* pay_a; uint32_t* pay_b;
  float* loop_w;
  void* cub; size_t cub_bytes;
  size_t total;
};

static int coo_ws_layout(int64_t M, int64_t N, void* base, CooWs* w) {
  size_t cub_bytes = 0;
  rocprim::double_buffer<uint64_t> dk(nullptr, nullptr);
  rocprim::double_buffer<uint32_t> dv(nullptr, nullptr);
  hipError_t e = rocprim::radix_sort_pairs(nullptr, cub_bytes, dk, dv, (size_t)M, 0u, 64u, (hipStream_t)0);
  if (e != hipSuccess) { set_hip_error(e, "rocprim::radix_sort_pairs(size query)"); return MP_ERR_HIP; }
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes, 256); return r; };
  w->keys_a = (uint64_t*)take((size_t)M * 8);
  w->keys_b = (uint64_t*)take((size_t)M * 8);
  w->pay_a = (uint32_t*)take((size_t)M * 4);
  w->pay_b = (uint32_t*)take((size_t)M * 4);
  w->loop_w = (float*)take((size_t)(N > 0 ? N : 1) * 4);
  w->cub = take(cub_bytes);
  w->cub_bytes = cub_bytes;
  w->total = off;
  return MP_OK;
}

__global__ __launch_bounds__(kBlock) void fill_f32_kernel(float* p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// key = dst << shift | src, shift = id_bits(N): the sort then runs over 2 * shift bits instead of 32 + shift (N = 6e5:
// 40 bits, five 8-bit passes instead of seven); removed self loops get row N (sorts behind every real row)
__global__ __launch_bounds__(kBlock) void coo_keys_kernel(const int64_t* __restrict__ dst,
                                                          const int64_t* __restrict__ src,
                                                          const float* __restrict__ w, int64_t E,
                                                          int64_t N, int flags, int shift, uint64_t* keys,
                                                          uint32_t* pay, float* loop_w) {
  const bool rm = flags & MP_COO_REMOVE_SELF_LOOPS;
  const bool add = flags & MP_COO_ADD_SELF_LOOPS;
  const bool keep = (flags & MP_COO_KEEP_LOOP_WEIGHT) && rm && add;
  const int64_t M = E + (add ? N : 0);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t key;
    if (i < E) {
      const int64_t d = dst[i], s = src[i];
      if (rm && d == s) {
        key = (uint64_t)N << shift;
        if (keep) loop_w[d] = w ? w[i] : 1.f;  // an existing loop's weight survives (last one wins)
      } else {
        key = ((uint64_t)d << shift) | (uint64_t)(uint32_t)s;
      }
    } else {
      const uint64_t n = (uint64_t)(i - E);
      key = (n << shift) | n;
    }
    keys[i] = key;
    pay[i] = (uint32_t)i;
  }
}

// Row starts from the sorted keys, written by the threads that walk the keys anyway (round 4; rounds 1-3 ran a
// separate launch with one binary search over all keys per row): position k opens every row in (row(k-1), row(k)] — its
// own and the empty rows in front of it — and the last position closes the rows behind it; rows are capped at N (the
// "removed" marker of the COO build), so rowptr[N] = the count of kept entries.
__device__ __forceinline__ void row_starts_from_keys(const uint64_t* __restrict__ keys, int64_t k, int64_t M, int64_t N,
                                                      int shift, uint64_t key, int32_t* __restrict__ rowptr) {
  int64_t row = (int64_t)(key >> shift);
  if (row > N) row = N;
  int64_t prev = -1;
  if (k > 0) {
    prev = (int64_t)(keys[k - 1] >> shift);
    if (prev > N) prev = N;
  }
  for (int64_t r = prev + 1; r <= row; ++r) rowptr[r] = (int32_t)k;
  if (k == M - 1)
    for (int64_t r = row + 1; r <= N; ++r) rowptr[r] = (int32_t)M;
}

__global__ __launch_bounds__(kBlock) void coo_emit_kernel(const uint64_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ pay,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ loop_w,
                                                          int64_t M, int64_t E, int64_t N, float fill,
                                                          int keep, int shift, int32_t* rowptr, int32_t* col,
                                                          float* val, int32_t* eid) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < M;
       k += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[k];
    row_starts_from_keys(keys, k, M, N, shift, key, rowptr);
    if ((int64_t)(key >> shift) >= N) continue;  // removed entry
    const uint32_t p = pay[k];
    col[k] = (int32_t)(uint32_t)(key & (((uint64_t)1 << shift) - 1));
    if ((int64_t)p < E) {
      if (val) val[k] = w ? w[p] : 1.f;
      if (eid) eid[k] = (int32_t)p;
    } else {
      const int64_t n = (int64_t)p - E;
      if (val) val[k] = keep ? loop_w[n] : fill;
      if (eid) eid[k] = (int32_t)(-1 - n);
    }
  }
}

__global__ __launch_bounds__(kBlock) void check_edges_kernel(const int64_t* __restrict__ dst,
                                                             const int64_t* __restrict__ src,
                                                             int64_t E, int64_t N, int32_t* bad) {
  int local = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t d = dst[i], s = src[i];
    if (d < 0 || d >= N || s < 0 || s >= N) ++local;
  }
  if (local) atomicAdd(bad, local);
}

// row of entry e by binary search: the last r with rowptr[r] <= e
__device__ __forceinline__ int row_of_entry(const int32_t* __restrict__ rowptr, int32_t N, int32_t e) {
  int lo = 0, hi = N;  // invariant: rowptr[lo] <= e < rowptr[hi]
  while (hi - lo > 1) {
    const int mid = lo + ((hi - lo) >> 1);
    if (rowptr[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// Rows of CONSECUTIVE entries, a workgroup at a time.  One global binary search per entry (24 dependent loads at 10^7
// rows) made row_ids run at 0.4 TB/s and the transpose's key pass at 1 TB/s (profiles/r03_kernel_table.md).  A tile of
// kRowTile consecutive entries spans few rows: two searches per TILE find its first and last row, the row starts in
// between are staged in LDS (coalesced), and every entry finds its row among them in LDS.  A tile that spans more rows
// than the stage holds (long runs of empty rows) searches the global array between the tile's two rows instead.
constexpr int kRowTile = 2048;

template <class Body>   // body(entry, row)
__device__ __forceinline__ void for_entries_with_rows(const int32_t* __restrict__ rowptr, int32_t N, int64_t nnz, Body body) {
  __shared__ int32_t rp_s[kRowTile + 2];
  __shared__ int32_t lim_s[2];
  const int64_t n_tiles = (nnz + kRowTile - 1) / kRowTile;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int64_t e0 = t * kRowTile;
    const int64_t e1 = e0 + kRowTile < nnz ? e0 + kRowTile : nnz;
    if (threadIdx.x < 2) lim_s[threadIdx.x] = row_of_entry(rowptr, N, (int32_t)(threadIdx.x == 0 ? e0 : e1 - 1));
    __syncthreads();
    const int r_lo = lim_s[0], r_hi = lim_s[1];
    const int R = r_hi - r_lo + 1;                       // rows the tile touches: starts rowptr[r_lo .. r_hi]
    const bool staged = R <= kRowTile + 1;
    if (staged)
      for (int i = threadIdx.x; i < R; i += kBlock) rp_s[i] = rowptr[r_lo + i];
    __syncthreads();
    for (int64_t e = e0 + threadIdx.x; e < e1; e += kBlock) {
      int lo = 0, hi = R;                                // invariant: start[lo] <= e < start[hi] (start[R] = +inf)
      if (staged) {
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (rp_s[mid] <= (int32_t)e) lo = mid; else hi = mid;
        }
      } else {
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (rowptr[r_lo + mid] <= (int32_t)e) lo = mid; else hi = mid;
        }
      }
      body(e, r_lo + lo);
    }
    __syncthreads();                                     // the stage is rewritten by the next tile
  }
}

__global__ __launch_bounds__(kBlock) void row_ids_kernel(const int32_t* __restrict__ rowptr, int32_t N,
                                                         int64_t nnz, int32_t* row_of) {
  for_entries_with_rows(rowptr, N, nnz, [&](int64_t e, int r) { row_of[e] = r; });
}

__global__ __launch_bounds__(kBlock) void transpose_keys_kernel(const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ col,
                                                                int32_t N, int64_t nnz, uint64_t* keys,
                                                                uint32_t* pay) {
  for_entries_with_rows(rowptr, N, nnz, [&](int64_t e, int r) {
    keys[e] = ((uint64_t)(uint32_t)col[e] << 32) | (uint64_t)(uint32_t)r;
    pay[e] = (uint32_t)e;
  });
}

// Is the stored operator its own transpose?  Every entry (r, c, v) looks for (c, r) in row c (columns ascend within a
// row: binary search) and compares the values bit for bit; a repeated (r, c) counts as "no" (A[r, c] is then a sum the
// search cannot see).  flag[0] is set to 1 by any entry that fails.
__global__ __launch_bounds__(kBlock) void csr_symmetric_kernel(const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ col,
                                                               const float* __restrict__ val, int32_t N, int64_t nnz,
                                                               int32_t* flag) {
  bool bad = false;
  for_entries_with_rows(rowptr, N, nnz, [&](int64_t e, int r) {
    const int c = col[e];
    if ((unsigned)c >= (unsigned)N) { bad = true; return; }   // (not a square operator)
    if (e + 1 < rowptr[r + 1] && col[e + 1] == c) bad = true;
    if (c == r) return;
    int lo = rowptr[c], hi = rowptr[c + 1];
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (col[mid] < r) lo = mid + 1; else hi = mid;
    }
    if (lo >= rowptr[c + 1] || col[lo] != r) { bad = true; return; }
    if (val && __float_as_uint(val[lo]) != __float_as_uint(val[e])) bad = true;
  });
  if (bad) flag[0] = 1;
}

__global__ __launch_bounds__(kBlock) void transpose_emit_kernel(const uint64_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ pay,
                                                                const float* __restrict__ val,
                                                                int64_t nnz, int64_t N, int32_t* t_rowptr, int32_t* t_col,
                                                                float* t_val, int32_t* pos) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t p = pay[k];
    row_starts_from_keys(keys, k, nnz, N, 32, keys[k], t_rowptr);
    t_col[k] = (int32_t)(uint32_t)keys[k];
    if (t_val && val) t_val[k] = val[p];
    if (pos) pos[k] = (int32_t)p;
  }
}

// Rows are walked by groups of kRowLanes lanes (4 rows per wave): short rows keep most lanes busy and a
// 10^4-entry hub row is split over the group instead of serialising one thread.
constexpr int kRowLanes = 16;

template <bool INV_SQRT = false>   // INV_SQRT: deg[r] = 1 / sqrt(row sum), 0 where that is not finite (inv_sqrt_kernel's rule)
__global__ __launch_bounds__(kBlock) void degree_row_kernel(const int32_t* __restrict__ rowptr,
                                                            const float* __restrict__ val, int64_t N,
                                                            float* deg) {
  const int sub = threadIdx.x % kRowLanes;
  const int64_t groups = (int64_t)gridDim.x * (kBlock / kRowLanes);
  for (int64_t r = (int64_t)blockIdx.x * (kBlock / kRowLanes) + threadIdx.x / kRowLanes; r < N; r += groups) {
    const int s = rowptr[r], e = rowptr[r + 1];
    float acc = 0.f;
    if (val) {
      // per-lane compensated (Kahan) partial sums, then a 16-lane tree: hub rows add 10^4 terms
      float comp = 0.f;
      for (int k = s + sub; k < e; k += kRowLanes) {
        const float yv = val[k] - comp;
        const float t = acc + yv;
        comp = (t - acc) - yv;
        acc = t;
      }
      for (int off = kRowLanes >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kRowLanes);
    } else {
      acc = (float)(e - s);
    }
    if constexpr (INV_SQRT) {
      float y = 1.0f / sqrtf(acc);
      if (isinf(y) || isnan(y)) y = 0.f;
      acc = y;
    }
    if (sub == 0) deg[r] = acc;
  }
}

__global__ __launch_bounds__(kBlock) void degree_col_kernel(const int32_t* __restrict__ col,
                                                            const float* __restrict__ val, int64_t nnz,
                                                            float* deg) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz;
       e += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&deg[col[e]], val ? val[e] : 1.f);
}

// deg -> deg^-1/2 with inf / nan -> 0 (TfgIDLayer.py:550-555; idconv.py:57-58,145-146)
__global__ __launch_bounds__(kBlock) void inv_sqrt_kernel(float* d, int64_t N) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float x = d[i];
    float y = 1.0f / sqrtf(x);
    if (isinf(y) || isnan(y)) y = 0.f;
    d[i] = y;
  }
}

__global__ __launch_bounds__(kBlock) void norm_edges_kernel(const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ val,
                                                            const float* __restrict__ dinv, int64_t N,
                                                            float* val_out) {
  const int sub = threadIdx.x % kRowLanes;
  const int64_t groups = (int64_t)gridDim.x * (kBlock / kRowLanes);
  for (int64_t r = (int64_t)blockIdx.x * (kBlock / kRowLanes) + threadIdx.x / kRowLanes; r < N; r += groups) {
    const int s = rowptr[r], e = rowptr[r + 1];
    const float dr = dinv[r];
    for (int k = s + sub; k < e; k += kRowLanes) {
      const float w = val ? val[k] : 1.f;
      val_out[k] = dr * w * dinv[col[k]];  // (D^-1/2 A) D^-1/2: TfgIDLayer.py:558, idconv.py:60,148
    }
  }
}

// val_out[e] = rs[row(e)] * val[e] * cs[col(e)]   (either scale may be null = ones)
__global__ __launch_bounds__(kBlock) void scale_edges_kernel(const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             const float* __restrict__ val,
                                                             const float* __restrict__ rs,
                                                             const float* __restrict__ cs, int64_t N,
                                                             float* val_out) {
  const int sub = threadIdx.x % kRowLanes;
  const int64_t groups = (int64_t)gridDim.x * (kBlock / kRowLanes);
  for (int64_t r = (int64_t)blockIdx.x * (kBlock / kRowLanes) + threadIdx.x / kRowLanes; r < N; r += groups) {
    const int s = rowptr[r], e = rowptr[r + 1];
    const float dr = rs ? rs[r] : 1.f;
    for (int k = s + sub; k < e; k += kRowLanes) {
      const float w = val ? val[k] : 1.f;
      val_out[k] = dr * w * (cs ? cs[col[k]] : 1.f);
    }
  }
}

__global__ __launch_bounds__(kBlock) void set_flags_kernel(const int64_t* __restrict__ idx, int64_t n,
                                                           int64_t N, uint8_t* flag) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx[i];
    if (j >= 0 && j < N) flag[j] = 1;
  }
}

__global__ __launch_bounds__(kBlock) void mark_cols_kernel(const int32_t* __restrict__ col, int64_t nnz,
                                                           const uint8_t* __restrict__ flag,
                                                           int32_t* col_out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t c = col[e] & 0x7fffffff;
    col_out[e] = flag[c] ? (int32_t)((uint32_t)c | 0x80000000u) : c;
  }
}

__global__ __launch_bounds__(kBlock) void rows_gather_kernel(const float* __restrict__ X, int64_t ldx,
                                                             const int64_t* __restrict__ idx, int64_t n,
                                                             int32_t d, float* out, int64_t ldo) {
  const int64_t total = n * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i / d;
    const int c = (int)(i - k * d);
    out[k * ldo + c] = X[idx[k] * ldx + c];
  }
}

__global__ __launch_bounds__(kBlock) void rows_scatter_add_kernel(float* H, int64_t ldh,
                                                                  const int64_t* __restrict__ idx,
                                                                  int64_t n, int32_t d,
                                                                  const float* __restrict__ U,
                                                                  int64_t ldu) {
  const int64_t total = n * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i / d;
    const int c = (int)(i - k * d);
    atomicAdd(&H[idx[k] * ldh + c], U[k * ldu + c]);
  }
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_csr_from_coo_ws_bytes(int64_t E, int64_t N, size_t* bytes_host) {
  if (!bytes_host || E < 0 || N < 0) return MP_ERR_INVALID_ARG;
  if (E + N >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  CooWs w;
  int st = coo_ws_layout(E + N, N, nullptr, &w);
  if (st != MP_OK) return st;
  *bytes_host = w.total;
  return MP_OK;
}

int mp_csr_from_coo(const int64_t* dst, const int64_t* src, const float* w, int64_t E, int64_t N,
                    int flags, float fill, int32_t* rowptr, int32_t* col, float* val, int32_t* eid,
                    void* ws, size_t ws_bytes, mp_stream_t stream) {
  if (E < 0 || N < 0 || !rowptr || (E > 0 && (!dst || !src))) return MP_ERR_INVALID_ARG;
  if (E + N >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  if (flags & ~15) return MP_ERR_INVALID_ARG;
  if ((flags & MP_COO_RECT) && (flags & (MP_COO_REMOVE_SELF_LOOPS | MP_COO_ADD_SELF_LOOPS))) return MP_ERR_INVALID_ARG;
  const bool add = flags & MP_COO_ADD_SELF_LOOPS;
  const bool keep = (flags & MP_COO_KEEP_LOOP_WEIGHT) && (flags & MP_COO_REMOVE_SELF_LOOPS) && add;
  const int64_t M = E + (add ? N : 0);
  if (M > 0 && !col) return MP_ERR_INVALID_ARG;
  if (!val && (w || (add && fill != 1.f))) return MP_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  CooWs L;
  int rc = coo_ws_layout(E + N, N, ws, &L);
  if (rc != MP_OK) return rc;
  if (!ws || ws_bytes < L.total) return MP_ERR_WORKSPACE;

  if (M == 0) {
    MP_HIP(hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * 4, st));
    return MP_OK;
  }
  if (keep) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3(flat_grid(N)), dim3(kBlock), 0, st, L.loop_w, N, fill);
    MP_LAUNCH_CHECK();
  }
  const int shift = (flags & MP_COO_RECT) ? 32 : id_bits(N);
  hipLaunchKernelGGL(coo_keys_kernel, dim3(flat_grid(M)), dim3(kBlock), 0, st, dst, src, w, E, N, flags, shift,
                     L.keys_a, L.pay_a, L.loop_w);
  MP_LAUNCH_CHECK();
  rocprim::double_buffer<uint64_t> dk(L.keys_a, L.keys_b);
  rocprim::double_buffer<uint32_t> dv(L.pay_a, L.pay_b);
  size_t cub_bytes = L.cub_bytes;
  MP_HIP(rocprim::radix_sort_pairs(L.cub, cub_bytes, dk, dv, (size_t)M, 0u, (unsigned)(shift + id_bits(N)), st));
  hipLaunchKernelGGL(coo_emit_kernel, dim3(flat_grid(M)), dim3(kBlock), 0, st, dk.current(), dv.current(),
                     w, L.loop_w, M, E, N, fill, keep ? 1 : 0, shift, rowptr, col, val, eid);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_check_edge_index(const int64_t* dst, const int64_t* src, int64_t E, int64_t N, int32_t* bad,
                        mp_stream_t stream) {
  if (!bad || E < 0 || (E > 0 && (!dst || !src))) return MP_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  MP_HIP(hipMemsetAsync(bad, 0, 4, st));
  if (E == 0) return MP_OK;
  hipLaunchKernelGGL(check_edges_kernel, dim3(flat_grid(E)), dim3(kBlock), 0, st, dst, src, E, N, bad);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_csr_row_ids(const int32_t* rowptr, int64_t N, int64_t nnz, int32_t* row_of, mp_stream_t stream) {
  if (!rowptr || N < 0 || nnz < 0 || (nnz > 0 && !row_of)) return MP_ERR_INVALID_ARG;
  if (nnz == 0) return MP_OK;
  hipLaunchKernelGGL(row_ids_kernel, dim3(flat_grid(ceil_div(nnz, kRowTile) * kBlock)), dim3(kBlock), 0, as_stream(stream), rowptr,
                     (int32_t)N, nnz, row_of);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_csr_is_symmetric(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int64_t nnz,
                        int32_t* flag, mp_stream_t stream) {
  if (!rowptr || !flag || N < 0 || nnz < 0 || (nnz > 0 && !col)) return MP_ERR_INVALID_ARG;
  if (nnz >= INT32_MAX || N >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  MP_HIP(hipMemsetAsync(flag, 0, 4, st));
  if (nnz == 0) return MP_OK;
  hipLaunchKernelGGL(csr_symmetric_kernel, dim3(flat_grid(ceil_div(nnz, kRowTile) * kBlock)), dim3(kBlock), 0, st, rowptr, col,
                     val, (int32_t)N, nnz, flag);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_csr_transpose_ws_bytes(int64_t nnz, int64_t N, size_t* bytes_host) {
  if (!bytes_host || nnz < 0 || N < 0) return MP_ERR_INVALID_ARG;
  if (nnz >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  CooWs w;
  int st = coo_ws_layout(nnz, N, nullptr, &w);
  if (st != MP_OK) return st;
  *bytes_host = w.total;
  return MP_OK;
}

int mp_csr_transpose(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                     int64_t n_cols, int64_t nnz, int32_t* t_rowptr, int32_t* t_col, float* t_val,
                     int32_t* pos, void* ws, size_t ws_bytes, mp_stream_t stream) {
  if (!rowptr || !t_rowptr || n_rows < 0 || n_cols < 0 || nnz < 0 || (nnz > 0 && (!col || !t_col)))
    return MP_ERR_INVALID_ARG;
  if (nnz >= INT32_MAX || n_rows >= INT32_MAX || n_cols >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  const int64_t N = n_cols;   // the transpose has one row per column of the source
  if (nnz == 0) {
    MP_HIP(hipMemsetAsync(t_rowptr, 0, (size_t)(N + 1) * 4, st));
    return MP_OK;
  }
  CooWs L;
  int rc = coo_ws_layout(nnz, N, ws, &L);
  if (rc != MP_OK) return rc;
  if (!ws || ws_bytes < L.total) return MP_ERR_WORKSPACE;
  hipLaunchKernelGGL(transpose_keys_kernel, dim3(flat_grid(ceil_div(nnz, kRowTile) * kBlock)), dim3(kBlock), 0, st, rowptr, col,
                     (int32_t)n_rows, nnz, L.keys_a, L.pay_a);
  MP_LAUNCH_CHECK();
  rocprim::double_buffer<uint64_t> dk(L.keys_a, L.keys_b);
  rocprim::double_buffer<uint32_t> dv(L.pay_a, L.pay_b);
  size_t cub_bytes = L.cub_bytes;
  // the entries arrive in (row, column) order and the sort is stable: sorting on the column bits alone leaves the rows of
  // a column ascending — 20 bits at 6e5 columns (three passes) instead of 52 (seven)
  MP_HIP(rocprim::radix_sort_pairs(L.cub, cub_bytes, dk, dv, (size_t)nnz, 32u, (unsigned)(32 + id_bits(N)), st));
  hipLaunchKernelGGL(transpose_emit_kernel, dim3(flat_grid(nnz)), dim3(kBlock), 0, st, dk.current(),
                     dv.current(), val, nnz, N, t_rowptr, t_col, t_val, pos);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_csr_degree(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int64_t nnz,
                  int axis, float* deg, mp_stream_t stream) {
  if (!rowptr || !deg || N < 0 || nnz < 0) return MP_ERR_INVALID_ARG;
  if (N == 0) return MP_OK;
  hipStream_t st = as_stream(stream);
  if (axis == MP_AXIS_ROW) {
    hipLaunchKernelGGL(degree_row_kernel<false>, dim3(flat_grid(N * kRowLanes)), dim3(kBlock), 0, st, rowptr, val, N, deg);
    MP_LAUNCH_CHECK();
  } else if (axis == MP_AXIS_COL) {
    if (nnz > 0 && !col) return MP_ERR_INVALID_ARG;
    MP_HIP(hipMemsetAsync(deg, 0, (size_t)N * 4, st));
    if (nnz > 0) {
      hipLaunchKernelGGL(degree_col_kernel, dim3(flat_grid(nnz)), dim3(kBlock), 0, st, col, val, nnz, deg);
      MP_LAUNCH_CHECK();
    }
  } else {
    return MP_ERR_INVALID_ARG;
  }
  return MP_OK;
}

int mp_gcn_norm_edges(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int64_t nnz,
                      int deg_axis, float* val_out, float* dinv_out, mp_stream_t stream) {
  if (!rowptr || !dinv_out || N < 0 || nnz < 0 || (nnz > 0 && (!col || !val_out))) return MP_ERR_INVALID_ARG;
  if (N == 0) return MP_OK;
  hipStream_t st = as_stream(stream);
  if (deg_axis == MP_AXIS_ROW) {   // row sums and their inverse square roots in one launch (the same operations: same bits)
    hipLaunchKernelGGL(degree_row_kernel<true>, dim3(flat_grid(N * kRowLanes)), dim3(kBlock), 0, st, rowptr, val, N, dinv_out);
    MP_LAUNCH_CHECK();
  } else {
    int rc = mp_csr_degree(rowptr, col, val, N, nnz, deg_axis, dinv_out, stream);
    if (rc != MP_OK) return rc;
    hipLaunchKernelGGL(inv_sqrt_kernel, dim3(flat_grid(N)), dim3(kBlock), 0, st, dinv_out, N);
    MP_LAUNCH_CHECK();
  }
  if (nnz > 0) {
    hipLaunchKernelGGL(norm_edges_kernel, dim3(flat_grid(N * kRowLanes)), dim3(kBlock), 0, st, rowptr, col, val,
                       dinv_out, N, val_out);
    MP_LAUNCH_CHECK();
  }
  return MP_OK;
}

int mp_csr_scale_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int64_t nnz,
                     const float* row_scale, const float* col_scale, float* val_out, mp_stream_t stream) {
  if (!rowptr || N < 0 || nnz < 0 || (nnz > 0 && (!col || !val_out))) return MP_ERR_INVALID_ARG;
  if (N == 0 || nnz == 0) return MP_OK;
  hipLaunchKernelGGL(scale_edges_kernel, dim3(flat_grid(N * kRowLanes)), dim3(kBlock), 0, as_stream(stream), rowptr,
                     col, val, row_scale, col_scale, N, val_out);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_mark_id_sources(const int32_t* col, int64_t nnz, const int64_t* id_index, int64_t n_id, int64_t N,
                       uint8_t* is_id, int32_t* col_out, mp_stream_t stream) {
  if (N < 0 || nnz < 0 || n_id < 0 || !is_id || (nnz > 0 && (!col || !col_out)) || (n_id > 0 && !id_index))
    return MP_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  if (N > 0) MP_HIP(hipMemsetAsync(is_id, 0, (size_t)N, st));
  if (n_id > 0) {
    hipLaunchKernelGGL(set_flags_kernel, dim3(flat_grid(n_id)), dim3(kBlock), 0, st, id_index, n_id, N, is_id);
    MP_LAUNCH_CHECK();
  }
  if (nnz > 0) {
    hipLaunchKernelGGL(mark_cols_kernel, dim3(flat_grid(nnz)), dim3(kBlock), 0, st, col, nnz, is_id, col_out);
    MP_LAUNCH_CHECK();
  }
  return MP_OK;
}

int mp_rows_gather_f32(const float* X, int64_t ldx, const int64_t* idx, int64_t n, int32_t d, float* out,
                       int64_t ldo, mp_stream_t stream) {
  if (n < 0 || d <= 0 || (n > 0 && (!X || !idx || !out)) || ldx < d || ldo < d) return MP_ERR_INVALID_ARG;
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(rows_gather_kernel, dim3(flat_grid(n * d)), dim3(kBlock), 0, as_stream(stream), X, ldx,
                     idx, n, d, out, ldo);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_rows_scatter_add_f32(float* H, int64_t ldh, const int64_t* idx, int64_t n, int32_t d, const float* U,
                            int64_t ldu, mp_stream_t stream) {
  if (n < 0 || d <= 0 || (n > 0 && (!H || !idx || !U)) || ldh < d || ldu < d) return MP_ERR_INVALID_ARG;
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(rows_scatter_add_kernel, dim3(flat_grid(n * d)), dim3(kBlock), 0, as_stream(stream), H,
                     ldh, idx, n, d, U, ldu);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // extern "C"

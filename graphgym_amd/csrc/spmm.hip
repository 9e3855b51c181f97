// Neighbour aggregation (CSR SpMM with sum / mean / max and the ID-GNN two-branch
// form) for gfx950.  Replaces SparseAdj.matmul (sparse_adj.py:91-97) and PyG
// propagate + torch_scatter (idconv.py:89,177,235,315,371) — see mp_engine.h.
//
// Shape of the kernel:
//   * one wavefront walks one *segment*: a run of consecutive destination rows
//     whose cost (entries + row_cost per row) is ~seg_cost, found at plan time
//     by a binary search over rowptr, so waves carry equal work on power-law
//     degree distributions;
//   * the wave reads the segment's column indices 64 at a time (one coalesced
//     load), broadcasts one index per step through v_readlane into an SGPR and
//     issues a fully coalesced row load of X (64 lanes x W floats: 1 KiB per
//     instruction at d = 256), U rows in flight per wave;
//   * rows are reduced in registers by the whole wave (a segmented reduction
//     whose segment boundaries are wave-uniform scalars), each output row is
//     written exactly once with the epilogue fused: no atomics, no memset,
//     bitwise reproducible;
//   * rows longer than hub_deg are cut into pieces that separate waves reduce
//     into a small partial buffer, summed in piece order by a finalize kernel.
#include "common.h"
#include "vecio.h"
#include <limits.h>

namespace mp {

struct AggArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const float* val;
  const int32_t* seg_row;
  int32_t n_seg;
  int32_t hub_deg;
  const float* X; int64_t ldx;
  float* Y; int64_t ldy;
  float* Q; int64_t ldq;
  const float* S; int64_t lds; float self_scale;
  const float* bias;
  const float* col_scale;   // per-column multiplier applied before bias (BatchNorm in eval mode folded)
  int32_t act;
  int32_t l2norm;           // normalise the finished row to unit L2 norm (single column tile only)
  float l2_eps;
  int32_t* argmax;
  int32_t d;
  int32_t head_width;       // NH > 1: val is [nnz, NH] and head h owns columns [h * head_width, (h + 1) * head_width)
  // hub path
  const int32_t* header;
  const int32_t* hub_row;
  const int32_t* hub_base;
  const int32_t* hub_np;
  const int32_t* piece_hub;
  const int32_t* piece_k;
  int32_t piece_edges;
  float* part;
  float* part2;
  int32_t* part_arg;
};

// Running reduction of one output row, W columns per lane.
template <int W, int REDUCE, bool BRANCH2>
struct RowAcc {
  float a[W];
  float b[BRANCH2 ? W : 1];
  int arg[REDUCE == MP_MAX ? W : 1];

  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int k = 0; k < W; ++k) {
      a[k] = (REDUCE == MP_MAX) ? -INFINITY : 0.f;
      if constexpr (BRANCH2) b[k] = 0.f;
      if constexpr (REDUCE == MP_MAX) arg[k] = -1;
    }
  }
  // one neighbour row v scaled by w; `marked`: source is an identity node; e: entry index
  __device__ __forceinline__ void add(const float (&v)[W], float w, bool marked, int e) {
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if constexpr (REDUCE == MP_MAX) {
        float m = w * v[k];
        if (m > a[k]) { a[k] = m; arg[k] = e; }
      } else {
        a[k] = fmaf(w, v[k], a[k]);
      }
    }
    if constexpr (BRANCH2) {
      if (marked) {   // wave-uniform
#pragma unroll
        for (int k = 0; k < W; ++k) b[k] = fmaf(w, v[k], b[k]);
      }
    }
  }
};

// Epilogue + store of one finished output row (K15/K17 fused into the flush).
template <int W, int REDUCE, bool BRANCH2, bool NT = true>
__device__ __forceinline__ void finish_row(const AggArgs& a, int row, int deg,
                                           RowAcc<W, REDUCE, BRANCH2>& acc,
                                           int c0, int c0ld, bool lane_on) {
  float out[W];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    if (REDUCE == MP_MEAN) out[k] = deg > 0 ? acc.a[k] / (float)deg : 0.f;
    else if (REDUCE == MP_MAX) out[k] = deg > 0 ? acc.a[k] : 0.f;
    else out[k] = acc.a[k];
  }
  if (a.S != nullptr) {
    float s[W];
    load_vec<W>(a.S + (int64_t)row * a.lds + c0ld, s);
#pragma unroll
    for (int k = 0; k < W; ++k) out[k] = fmaf(a.self_scale, s[k], out[k]);
  }
  if (a.col_scale != nullptr) {
    float sv[W];
    load_vec<W>(a.col_scale + c0ld, sv);
#pragma unroll
    for (int k = 0; k < W; ++k) out[k] *= sv[k];
  }
  if (a.bias != nullptr) {
    float bv[W];
    load_vec<W>(a.bias + c0ld, bv);
#pragma unroll
    for (int k = 0; k < W; ++k) out[k] += bv[k];
  }
  if (a.act == MP_ACT_RELU) {
#pragma unroll
    for (int k = 0; k < W; ++k) out[k] = fmaxf(out[k], 0.f);
  }
  if (a.l2norm) {
    // F.normalize(p=2, dim=-1) (layer.py:43-46, gnn.py:79-80): the wave holds the whole row
    float ss = 0.f;
    if (lane_on) {
#pragma unroll
      for (int k = 0; k < W; ++k) ss = fmaf(out[k], out[k], ss);
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, kWave);
    const float inv = 1.0f / fmaxf(sqrtf(ss), a.l2_eps);
#pragma unroll
    for (int k = 0; k < W; ++k) out[k] *= inv;
  }
  if (lane_on) {
    if constexpr (NT) store_vec_nt<W>(a.Y + (int64_t)row * a.ldy + c0, out);
    else store_vec<W>(a.Y + (int64_t)row * a.ldy + c0, out);
    if constexpr (BRANCH2) store_vec<W>(a.Q + (int64_t)row * a.ldq + c0, acc.b);
    if constexpr (REDUCE == MP_MAX) {
      if (a.argmax != nullptr) store_ivec<W>(a.argmax + (int64_t)row * a.d + c0, acc.arg);
    }
  }
  acc.reset();
}

// Main kernel: one wave per segment of whole rows.  Kept from the round-1 variant study: U = 8 rows
// in flight, non-temporal stores of Y (-1.2 %); non-temporal index loads, index prefetch and an LDS-staged index
// tile measured within 0.3 % and are not built.
// NH > 1 (multi-head attention, TfgIDLayer.py:333-355): every entry carries NH weights (val [nnz, NH]); a lane applies
// the weight of the head its columns belong to, so all heads aggregate in one launch on full 1 KiB row loads.
template <int W, int REDUCE, bool WEIGHTED, bool BRANCH2, int U, int NH = 1>
__global__ __launch_bounds__(kBlock) void agg_rows_kernel(AggArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int seg = blockIdx.x * kWavesPerBlock + wave;
  if (seg >= a.n_seg) return;
  const int c0 = (blockIdx.y * kWave + lane) * W;
  const bool lane_on = c0 < a.d;
  const int c0ld = lane_on ? c0 : 0;  // idle lanes re-read column 0, never store

  const int r0 = a.seg_row[seg];
  int r1 = a.seg_row[seg + 1];
  if (r0 >= r1) return;
  const int e0 = a.rowptr[r0];
  int e1 = a.rowptr[r1];
  {
    // a hub row can only be the last row that starts in a segment; the hub path owns it
    const int last_start = a.rowptr[r1 - 1];
    if (e1 - last_start > a.hub_deg) { r1 -= 1; e1 = last_start; }
  }
  if (r0 >= r1) return;

  const float* __restrict__ xlane = a.X + c0ld;
  const int myh = NH > 1 ? c0ld / a.head_width : 0;

  // row ends of up to 64 rows live in one VGPR; the current one is broadcast to an SGPR
  int rbase = r0;
  int rendv = (rbase + lane < r1) ? a.rowptr[rbase + 1 + lane] : INT_MAX;
  int r = r0;
  int rstart = e0;
  int rend = bcast_i(rendv, 0);

  RowAcc<W, REDUCE, BRANCH2> acc;
  acc.reset();

  auto advance = [&]() {
    r += 1;
    rstart = rend;
    if (r - rbase == kWave) {
      rbase = r;
      rendv = (rbase + lane < r1) ? a.rowptr[rbase + 1 + lane] : INT_MAX;
    }
    rend = (r < r1) ? bcast_i(rendv, r - rbase) : INT_MAX;
  };

  for (int ec = e0; ec < e1; ec += kWave) {
    const int me = min(ec + lane, e1 - 1);
    const int cv = a.col[me];
    float wv[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) wv[h] = WEIGHTED ? a.val[(int64_t)me * NH + h] : 1.f;
    const int n = min(kWave, e1 - ec);
    for (int jb = 0; jb < n; jb += U) {
      float v[U][W];
      int cj[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        cj[j] = bcast_i(cv, jb + j);
        const int c = BRANCH2 ? (cj[j] & 0x7fffffff) : cj[j];
        load_vec<W>(xlane + (int64_t)c * a.ldx, v[j]);
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int e = ec + jb + j;
        if (e < e1) {
          while (e >= rend) {
            finish_row<W, REDUCE, BRANCH2, true>(a, r, rend - rstart, acc, c0, c0ld, lane_on);
            advance();
          }
          float w = WEIGHTED ? bcast_f(wv[0], jb + j) : 1.f;
#pragma unroll
          for (int h = 1; h < NH; ++h) {
            const float wh = bcast_f(wv[h], jb + j);
            w = myh == h ? wh : w;
          }
          acc.add(v[j], w, BRANCH2 && cj[j] < 0, e);
        }
      }
    }
  }
  while (r < r1) {
    finish_row<W, REDUCE, BRANCH2, true>(a, r, rend - rstart, acc, c0, c0ld, lane_on);
    advance();
  }
}

// Hub path 1/2: one wave reduces one piece (<= piece_edges entries) of a hub row.
template <int W, int REDUCE, bool WEIGHTED, bool BRANCH2, int U, int NH = 1>
__global__ __launch_bounds__(kBlock) void agg_hub_pieces_kernel(AggArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c0 = (blockIdx.y * kWave + lane) * W;
  const bool lane_on = c0 < a.d;
  const int c0ld = lane_on ? c0 : 0;
  const float* __restrict__ xlane = a.X + c0ld;
  const int myh = NH > 1 ? c0ld / a.head_width : 0;
  const int n_piece = a.header[PW_NPIECE];

  for (int p = blockIdx.x * kWavesPerBlock + wave; p < n_piece; p += gridDim.x * kWavesPerBlock) {
    const int h = a.piece_hub[p];
    const int k = a.piece_k[p];
    const int row = a.hub_row[h];
    const int rs = a.rowptr[row];
    const int re = a.rowptr[row + 1];
    const int e0 = rs + k * a.piece_edges;
    const int e1 = min(e0 + a.piece_edges, re);

    RowAcc<W, REDUCE, BRANCH2> acc;
    acc.reset();
    for (int ec = e0; ec < e1; ec += kWave) {
      const int me = min(ec + lane, e1 - 1);
      const int cv = a.col[me];
      float wv[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) wv[h] = WEIGHTED ? a.val[(int64_t)me * NH + h] : 1.f;
      const int n = min(kWave, e1 - ec);
      for (int jb = 0; jb < n; jb += U) {
        float v[U][W];
        int cj[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
          cj[j] = bcast_i(cv, jb + j);
          const int c = BRANCH2 ? (cj[j] & 0x7fffffff) : cj[j];
          load_vec<W>(xlane + (int64_t)c * a.ldx, v[j]);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const int e = ec + jb + j;
          if (e < e1) {
            float w = WEIGHTED ? bcast_f(wv[0], jb + j) : 1.f;
#pragma unroll
            for (int h = 1; h < NH; ++h) {
              const float wh = bcast_f(wv[h], jb + j);
              w = myh == h ? wh : w;
            }
            acc.add(v[j], w, BRANCH2 && cj[j] < 0, e);
          }
        }
      }
    }
    if (lane_on) {
      store_vec<W>(a.part + (int64_t)p * a.d + c0, acc.a);
      if constexpr (BRANCH2) store_vec<W>(a.part2 + (int64_t)p * a.d + c0, acc.b);
      if constexpr (REDUCE == MP_MAX) store_ivec<W>(a.part_arg + (int64_t)p * a.d + c0, acc.arg);
    }
  }
}

// Hub path 2/2: combine a hub row's pieces in piece order, run the epilogue, store.
template <int W, int REDUCE, bool BRANCH2>
__global__ __launch_bounds__(kBlock) void agg_hub_finalize_kernel(AggArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c0 = (blockIdx.y * kWave + lane) * W;
  const bool lane_on = c0 < a.d;
  const int c0ld = lane_on ? c0 : 0;
  const int n_hub = a.header[PW_NHUB];

  for (int h = blockIdx.x * kWavesPerBlock + wave; h < n_hub; h += gridDim.x * kWavesPerBlock) {
    const int row = a.hub_row[h];
    const int base = a.hub_base[h];
    const int np = a.hub_np[h];
    const int deg = a.rowptr[row + 1] - a.rowptr[row];
    RowAcc<W, REDUCE, BRANCH2> acc;
    acc.reset();
    for (int p = base; p < base + np; ++p) {
      float v[W];
      load_vec<W>(a.part + (int64_t)p * a.d + c0ld, v);
      if constexpr (REDUCE == MP_MAX) {
        int ai[W];
        load_ivec<W>(a.part_arg + (int64_t)p * a.d + c0ld, ai);
#pragma unroll
        for (int k = 0; k < W; ++k)
          if (v[k] > acc.a[k]) { acc.a[k] = v[k]; acc.arg[k] = ai[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < W; ++k) acc.a[k] += v[k];
      }
      if constexpr (BRANCH2) {
        float v2[W];
        load_vec<W>(a.part2 + (int64_t)p * a.d + c0ld, v2);
#pragma unroll
        for (int k = 0; k < W; ++k) acc.b[k] += v2[k];
      }
    }
    finish_row<W, REDUCE, BRANCH2>(a, row, deg, acc, c0, c0ld, lane_on);
  }
}

// ---- plan ---------------------------------------------------------------

// The segmentation's tunables travel with the plan (its header words and counts_host carry them): a segment is a
// run of whole rows of cost ~seg_cost (1 per stored entry + row_cost per row); rows with more than hub_deg entries
// are split into pieces of piece_edges.
static const PlanCfg kDefaultCfg = {320, 4, 1024, 256};

static int cfg_from(const int32_t* cfg_host, PlanCfg* c) {
  if (!cfg_host) { *c = kDefaultCfg; return MP_OK; }
  PlanCfg v = {cfg_host[0], cfg_host[1], cfg_host[2], cfg_host[3]};
  if (v.seg_cost < 64 || v.row_cost < 0 || v.hub_deg < v.seg_cost || v.piece_edges < 64) return MP_ERR_INVALID_ARG;
  *c = v;
  return MP_OK;
}

static int32_t n_seg_of(int64_t N, int64_t nnz, const PlanCfg& c) {
  int64_t total = nnz + (int64_t)c.row_cost * N;
  int64_t s = ceil_div(total, c.seg_cost);
  return (int32_t)(s < 1 ? 1 : s);
}

static size_t plan_words(int64_t N, int64_t nnz, const PlanCfg& c) {
  int64_t n_seg = n_seg_of(N, nnz, c);
  int64_t cap_hub = nnz / c.hub_deg + 1;
  int64_t cap_piece = nnz / c.piece_edges + cap_hub + 1;
  return (size_t)(PW_HEADER_WORDS + (n_seg + 1) + 3 * cap_hub + 2 * cap_piece);
}

static PlanView plan_view(const int32_t* plan, int64_t N, int64_t nnz, const PlanCfg& c) {
  PlanView v;
  v.n_seg = n_seg_of(N, nnz, c);
  v.cap_hub = (int32_t)(nnz / c.hub_deg + 1);
  v.cap_piece = (int32_t)(nnz / c.piece_edges + v.cap_hub + 1);
  v.header = plan;
  v.seg_row = plan + PW_HEADER_WORDS;
  v.hub_row = v.seg_row + (v.n_seg + 1);
  v.hub_base = v.hub_row + v.cap_hub;
  v.hub_np = v.hub_base + v.cap_hub;
  v.piece_hub = v.hub_np + v.cap_hub;
  v.piece_k = v.piece_hub + v.cap_piece;
  return v;
}

__global__ void plan_header_kernel(int32_t* plan, int32_t n_seg, PlanCfg c, int32_t cap_hub,
                                   int32_t cap_piece) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    plan[PW_MAGIC] = kPlanMagic;
    plan[PW_NSEG] = n_seg;
    plan[PW_SEG_COST] = c.seg_cost;
    plan[PW_ROW_COST] = c.row_cost;
    plan[PW_HUB_DEG] = c.hub_deg;
    plan[PW_PIECE_EDGES] = c.piece_edges;
    plan[PW_NHUB] = 0;
    plan[PW_NPIECE] = 0;
    plan[PW_CAP_HUB] = cap_hub;
    plan[PW_CAP_PIECE] = cap_piece;
  }
}

// seg_row[s] = first row r with rowptr[r] + row_cost * r >= s * seg_cost  (s < n_seg);
// seg_row[n_seg] = N.
__global__ __launch_bounds__(kBlock) void plan_seg_kernel(const int32_t* __restrict__ rowptr,
                                                          int32_t N, int32_t n_seg, int seg_cost,
                                                          int row_cost, int32_t* seg_row) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= n_seg;
       s += (int64_t)gridDim.x * blockDim.x) {
    if (s == n_seg) { seg_row[s] = N; continue; }
    const int64_t target = s * (int64_t)seg_cost;
    int lo = 0, hi = N;  // answer in [0, N]
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      const int64_t p = (int64_t)rowptr[mid] + (int64_t)row_cost * mid;
      if (p >= target) hi = mid; else lo = mid + 1;
    }
    seg_row[s] = lo;
  }
}

__global__ __launch_bounds__(kBlock) void plan_hub_kernel(const int32_t* __restrict__ rowptr,
                                                          int32_t N, int hub_deg, int piece_edges,
                                                          int32_t* header, int32_t* hub_row,
                                                          int32_t* hub_base, int32_t* hub_np,
                                                          int32_t* piece_hub, int32_t* piece_k) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int deg = rowptr[r + 1] - rowptr[r];
    if (deg > hub_deg) {
      const int np = (deg + piece_edges - 1) / piece_edges;
      const int h = atomicAdd(&header[PW_NHUB], 1);
      const int base = atomicAdd(&header[PW_NPIECE], np);
      hub_row[h] = (int32_t)r;
      hub_base[h] = base;
      hub_np[h] = np;
      for (int k = 0; k < np; ++k) {
        piece_hub[base + k] = h;
        piece_k[base + k] = k;
      }
    }
  }
}

// ---- dispatch -------------------------------------------------------------

template <int W, int REDUCE, bool WEIGHTED, bool BRANCH2, int NH = 1>
static int launch_agg(const AggArgs& a, int64_t N, const int32_t* counts, hipStream_t st) {
  constexpr int U = 8;
  const int tiles = (int)ceil_div(a.d, kWave * W);
  dim3 grid((unsigned)ceil_div(a.n_seg, kWavesPerBlock), (unsigned)tiles);
  hipLaunchKernelGGL((agg_rows_kernel<W, REDUCE, WEIGHTED, BRANCH2, U, NH>), grid, dim3(kBlock), 0, st, a);
  MP_LAUNCH_CHECK();
  const int n_hub = counts[1], n_piece = counts[2];
  if (n_hub > 0) {
    int pb = (int)ceil_div(n_piece, kWavesPerBlock);
    if (pb > kNumCU * 8) pb = kNumCU * 8;
    hipLaunchKernelGGL((agg_hub_pieces_kernel<W, REDUCE, WEIGHTED, BRANCH2, U, NH>), dim3(pb, tiles),
                       dim3(kBlock), 0, st, a);
    MP_LAUNCH_CHECK();
    int hb = (int)ceil_div(n_hub, kWavesPerBlock);
    if (hb > kNumCU * 8) hb = kNumCU * 8;
    hipLaunchKernelGGL((agg_hub_finalize_kernel<W, REDUCE, BRANCH2>), dim3(hb, tiles), dim3(kBlock),
                       0, st, a);
    MP_LAUNCH_CHECK();
  }
  (void)N;
  return MP_OK;
}

template <int W, bool BRANCH2>
static int dispatch_reduce(const AggArgs& a, int64_t N, const int32_t* counts, int reduce,
                           hipStream_t st) {
  const bool weighted = a.val != nullptr;
  if constexpr (BRANCH2) {
    return weighted ? launch_agg<W, MP_SUM, true, true>(a, N, counts, st)
                    : launch_agg<W, MP_SUM, false, true>(a, N, counts, st);
  }
  switch (reduce) {
    case MP_SUM:
      return weighted ? launch_agg<W, MP_SUM, true, false>(a, N, counts, st)
                      : launch_agg<W, MP_SUM, false, false>(a, N, counts, st);
    case MP_MEAN:
      return weighted ? launch_agg<W, MP_MEAN, true, false>(a, N, counts, st)
                      : launch_agg<W, MP_MEAN, false, false>(a, N, counts, st);
    case MP_MAX:
      return weighted ? launch_agg<W, MP_MAX, true, false>(a, N, counts, st)
                      : launch_agg<W, MP_MAX, false, false>(a, N, counts, st);
  }
  return MP_ERR_INVALID_ARG;
}

static bool aligned(const void* p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; }

// widest per-lane vector every operand allows, then no wider than the row needs
static int pick_width(const AggArgs& a) {
  auto ok = [&](int w) {
    const size_t bytes = 4u * w;
    if (a.d % w) return false;
    if (a.head_width > 0 && a.head_width % w) return false;   // a lane's columns stay inside one head
    if (a.ldx % w || a.ldy % w) return false;
    if (a.Q && a.ldq % w) return false;
    if (a.S && a.lds % w) return false;
    return aligned(a.X, bytes) && aligned(a.Y, bytes) && aligned(a.Q, bytes) &&
           aligned(a.S, bytes) && aligned(a.bias, bytes) && aligned(a.col_scale, bytes) &&
           aligned(a.argmax, bytes) &&
           aligned(a.part, bytes) && aligned(a.part2, bytes) && aligned(a.part_arg, bytes);
  };
  int w = 4;
  while (w > 1 && !ok(w)) w >>= 1;
  while (w > 1 && kWave * (w / 2) >= a.d) w >>= 1;  // d=128 -> 2, d=64 -> 1: keep all lanes busy
  return w;
}

static int agg_common(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N,
                      const int32_t* plan, const int32_t* counts, const float* X, int64_t ldx,
                      float* Y, int64_t ldy, float* Q, int64_t ldq, int32_t d, int reduce,
                      const float* S, int64_t lds, float self_scale, const float* bias, int act,
                      int32_t* argmax, void* ws, size_t ws_bytes, hipStream_t st,
                      const float* col_scale = nullptr, int l2norm = 0, float l2_eps = 1e-12f, int heads = 1) {
  if (!rowptr || !plan || !counts || !X || !Y) return MP_ERR_INVALID_ARG;
  if (N < 0 || d <= 0 || ldx < d || ldy < d) return MP_ERR_INVALID_ARG;
  if (reduce < MP_SUM || reduce > MP_MAX) return MP_ERR_INVALID_ARG;
  if (act != MP_ACT_NONE && act != MP_ACT_RELU) return MP_ERR_INVALID_ARG;
  if (Q && ldq < d) return MP_ERR_INVALID_ARG;
  if (S && lds < d) return MP_ERR_INVALID_ARG;
  if (N >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  if (N == 0) return MP_OK;
  const int32_t n_seg = counts[0], n_piece = counts[2];
  if (n_seg < 1) return MP_ERR_INVALID_ARG;
  if (!col && n_piece > 0) return MP_ERR_INVALID_ARG;

  size_t need = 0;
  mp_spmm_ws_bytes(counts, d, reduce, Q != nullptr, &need);
  if (need > 0 && (!ws || ws_bytes < need)) return MP_ERR_WORKSPACE;

  AggArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val;
  a.header = plan;
  a.seg_row = plan + PW_HEADER_WORDS;
  a.n_seg = n_seg;
  a.hub_deg = counts[6];       // the config the plan was built under
  a.piece_edges = counts[7];
  a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy; a.Q = Q; a.ldq = ldq;
  a.S = S; a.lds = lds; a.self_scale = self_scale; a.bias = bias; a.act = act;
  a.col_scale = col_scale; a.l2norm = l2norm; a.l2_eps = l2_eps;
  a.argmax = argmax; a.d = d;
  a.head_width = heads > 1 ? d / heads : 0;
  a.hub_row = a.hub_base = a.hub_np = a.piece_hub = a.piece_k = nullptr;
  a.part = a.part2 = nullptr; a.part_arg = nullptr;
  if (n_piece > 0) {
    // the hub arrays sit behind seg_row; counts[3], counts[4] = their capacities
    const int32_t cap_hub = counts[3], cap_piece = counts[4];
    a.hub_row = a.seg_row + (n_seg + 1);
    a.hub_base = a.hub_row + cap_hub;
    a.hub_np = a.hub_base + cap_hub;
    a.piece_hub = a.hub_np + cap_hub;
    a.piece_k = a.piece_hub + cap_piece;
    char* w = (char*)ws;
    const size_t slab = align_up((size_t)n_piece * d * 4, 256);
    a.part = (float*)w; w += slab;
    if (Q) { a.part2 = (float*)w; w += slab; }
    if (reduce == MP_MAX) { a.part_arg = (int32_t*)w; w += slab; }
  }

  const int w = pick_width(a);
  if (l2norm && d > kWave * w) return MP_ERR_UNSUPPORTED;   // the row must sit in one wave
  if (heads > 1) {
    if (!val || Q || reduce != MP_SUM || d % heads) return MP_ERR_INVALID_ARG;
#define MP_HEADS(WV)                                                                           \
    switch (heads) {                                                                           \
      case 2: return launch_agg<WV, MP_SUM, true, false, 2>(a, N, counts, st);                \
      case 4: return launch_agg<WV, MP_SUM, true, false, 4>(a, N, counts, st);                \
      case 8: return launch_agg<WV, MP_SUM, true, false, 8>(a, N, counts, st);                \
      default: return MP_ERR_UNSUPPORTED;                                                      \
    }
    if (w == 4) { MP_HEADS(4) } else if (w == 2) { MP_HEADS(2) } else { MP_HEADS(1) }
#undef MP_HEADS
  }
  const bool two = Q != nullptr;
  switch (w) {
    case 4: return two ? dispatch_reduce<4, true>(a, N, counts, reduce, st)
                       : dispatch_reduce<4, false>(a, N, counts, reduce, st);
    case 2: return two ? dispatch_reduce<2, true>(a, N, counts, reduce, st)
                       : dispatch_reduce<2, false>(a, N, counts, reduce, st);
    default: return two ? dispatch_reduce<1, true>(a, N, counts, reduce, st)
                        : dispatch_reduce<1, false>(a, N, counts, reduce, st);
  }
}

__global__ __launch_bounds__(kBlock) void max_bwd_kernel(const int32_t* __restrict__ col,
                                                         const float* __restrict__ val,
                                                         const int32_t* __restrict__ argmax,
                                                         const float* __restrict__ dY, int64_t ldy,
                                                         int64_t N, int32_t d, float* dX, int64_t ldx) {
  const int64_t total = N * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / d;
    const int c = (int)(i - r * d);
    const int e = argmax[i];
    if (e >= 0) atomicAdd(&dX[(int64_t)col[e] * ldx + c], (val ? val[e] : 1.f) * dY[r * ldy + c]);
  }
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_spmm_plan_bytes(int64_t N, int64_t nnz, const int32_t* cfg_host, size_t* bytes_host) {
  if (!bytes_host || N < 0 || nnz < 0) return MP_ERR_INVALID_ARG;
  if (nnz >= INT32_MAX || N >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  PlanCfg c;
  if (int st = cfg_from(cfg_host, &c)) return st;
  *bytes_host = plan_words(N, nnz, c) * sizeof(int32_t);
  return MP_OK;
}

// counts_host: {n_seg, n_hub, n_piece, cap_hub, cap_piece, seg_cost, hub_deg, piece_edges}
int mp_spmm_plan_build(const int32_t* rowptr, int64_t N, int64_t nnz, const int32_t* cfg_host, int32_t* plan,
                       size_t plan_bytes, int32_t* counts_host, mp_stream_t stream) {
  if (!rowptr || !plan || !counts_host || N < 0 || nnz < 0) return MP_ERR_INVALID_ARG;
  if (nnz >= INT32_MAX || N >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  PlanCfg c;
  if (int cst = cfg_from(cfg_host, &c)) return cst;
  if (plan_bytes < plan_words(N, nnz, c) * sizeof(int32_t)) return MP_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  PlanView v = plan_view(plan, N, nnz, c);
  hipLaunchKernelGGL(plan_header_kernel, dim3(1), dim3(64), 0, st, plan, v.n_seg, c, v.cap_hub,
                     v.cap_piece);
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(plan_seg_kernel, dim3(flat_grid(v.n_seg + 1)), dim3(kBlock), 0, st, rowptr,
                     (int32_t)N, v.n_seg, c.seg_cost, c.row_cost, (int32_t*)v.seg_row);
  MP_LAUNCH_CHECK();
  if (N > 0) {
    hipLaunchKernelGGL(plan_hub_kernel, dim3(flat_grid(N)), dim3(kBlock), 0, st, rowptr, (int32_t)N,
                       c.hub_deg, c.piece_edges, plan, (int32_t*)v.hub_row, (int32_t*)v.hub_base,
                       (int32_t*)v.hub_np, (int32_t*)v.piece_hub, (int32_t*)v.piece_k);
    MP_LAUNCH_CHECK();
  }
  int32_t hdr[PW_HEADER_WORDS];
  MP_HIP(hipMemcpyAsync(hdr, plan, sizeof(hdr), hipMemcpyDeviceToHost, st));
  MP_HIP(hipStreamSynchronize(st));
  counts_host[0] = hdr[PW_NSEG];
  counts_host[1] = hdr[PW_NHUB];
  counts_host[2] = hdr[PW_NPIECE];
  counts_host[3] = hdr[PW_CAP_HUB];
  counts_host[4] = hdr[PW_CAP_PIECE];
  counts_host[5] = hdr[PW_SEG_COST];
  counts_host[6] = hdr[PW_HUB_DEG];
  counts_host[7] = hdr[PW_PIECE_EDGES];
  return MP_OK;
}

int mp_spmm_ws_bytes(const int32_t* counts_host, int32_t d, int reduce, int two_branch,
                     size_t* bytes_host) {
  if (!counts_host || !bytes_host || d <= 0) return MP_ERR_INVALID_ARG;
  const size_t slab = align_up((size_t)counts_host[2] * d * 4, 256);
  size_t n = counts_host[2] > 0 ? slab : 0;
  if (counts_host[2] > 0 && two_branch) n += slab;
  if (counts_host[2] > 0 && reduce == MP_MAX) n += slab;
  *bytes_host = n;
  return MP_OK;
}

int mp_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N,
                    const int32_t* plan, const int32_t* counts_host, const float* X, int64_t ldx,
                    float* Y, int64_t ldy, int32_t d, int reduce, const float* S, int64_t lds,
                    float self_scale, const float* bias, int act, int32_t* argmax, void* ws,
                    size_t ws_bytes, mp_stream_t stream) {
  return agg_common(rowptr, col, val, N, plan, counts_host, X, ldx, Y, ldy, nullptr, 0, d, reduce, S,
                    lds, self_scale, bias, act, argmax, ws, ws_bytes, as_stream(stream));
}

int mp_spmm_csr_epilogue_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N,
                             const int32_t* plan, const int32_t* counts_host, const float* X, int64_t ldx,
                             float* Y, int64_t ldy, int32_t d, int reduce, const float* S, int64_t lds,
                             float self_scale, const float* col_scale, const float* col_shift, int act,
                             int l2_normalize, float l2_eps, void* ws, size_t ws_bytes, mp_stream_t stream) {
  return agg_common(rowptr, col, val, N, plan, counts_host, X, ldx, Y, ldy, nullptr, 0, d, reduce, S, lds,
                    self_scale, col_shift, act, nullptr, ws, ws_bytes, as_stream(stream), col_scale,
                    l2_normalize ? 1 : 0, l2_eps);
}

int mp_idgnn_agg_f32(const int32_t* rowptr, const int32_t* col_marked, const float* val, int64_t N,
                     const int32_t* plan, const int32_t* counts_host, const float* X, int64_t ldx,
                     float* P, int64_t ldp, float* Q, int64_t ldq, int32_t d, void* ws,
                     size_t ws_bytes, mp_stream_t stream) {
  if (!Q) return MP_ERR_INVALID_ARG;
  return agg_common(rowptr, col_marked, val, N, plan, counts_host, X, ldx, P, ldp, Q, ldq, d, MP_SUM,
                    nullptr, 0, 0.f, nullptr, MP_ACT_NONE, nullptr, ws, ws_bytes, as_stream(stream));
}

int mp_spmm_csr_heads_f32(const int32_t* rowptr, const int32_t* col, const float* a, int64_t N, const int32_t* plan,
                          const int32_t* counts_host, int32_t heads, const float* V, int64_t ldv, float* Y, int64_t ldy,
                          int32_t d, void* ws, size_t ws_bytes, mp_stream_t stream) {
  if (heads < 1 || !a) return MP_ERR_INVALID_ARG;
  return agg_common(rowptr, col, a, N, plan, counts_host, V, ldv, Y, ldy, nullptr, 0, d, MP_SUM, nullptr, 0, 0.f,
                    nullptr, MP_ACT_NONE, nullptr, ws, ws_bytes, as_stream(stream), nullptr, 0, 1e-12f, heads);
}

int mp_spmm_max_bwd_f32(const int32_t* col, const float* val, const int32_t* argmax, const float* dY,
                        int64_t ldy, int64_t N, int32_t d, float* dX, int64_t ldx, mp_stream_t stream) {
  if (!col || !argmax || !dY || !dX || N < 0 || d <= 0) return MP_ERR_INVALID_ARG;
  if (N == 0) return MP_OK;
  hipLaunchKernelGGL(max_bwd_kernel, dim3(flat_grid(N * d)), dim3(kBlock), 0, as_stream(stream), col,
                     val, argmax, dY, ldy, N, d, dX, ldx);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // extern "C"

// Aggregate -> transform in one kernel:   out = act( (A X [+ s * S]) W + bias )
// i.e. SparseAdj.matmul followed by the layer's kernel product (TfgIDLayer.py:510-523 in the
// aggregate-first order, GIN's (1 + eps) x + sum -> first Linear, idconv.py:371-399) without the
// [N, F] intermediate making a round trip through HBM, and with the MFMA work of one row tile
// running while the other workgroups of the compute unit are still gathering theirs.
//
// Measured background (profiles/r01_overlap.log): the gather kernel reaches 88 % of its full-chip
// rate on half of the compute units — it is bound by HBM, not by issue slots — so the matrix cores
// of every CU are idle most of the time; a separate GEMM then needs its own 11 ms.  Here:
//   * a workgroup (4 waves) owns a tile of 32 consecutive destination rows;
//   * phase A: the tile's stored entries are split into four equal runs, one per wave (a row cut
//     by a run boundary is finished through a carry row, added in wave order => bitwise
//     reproducible); each wave walks its run exactly like the aggregation kernel (64 indices per
//     coalesced load, one v_readlane broadcast per entry, 1 KiB row loads, U in flight) and
//     leaves the reduced rows in LDS;
//   * phase B: the 32 x F tile in LDS times W on the matrix cores (v_mfma_f32_32x32x2_f32, exact
//     fp32 fma chain): each wave owns 64 output columns, B fragments come straight from W in L2
//     (256 KiB, resident in every XCD's L2) as 8-byte loads, bias + activation fused into the store.
// 36 KiB LDS per workgroup => 4 workgroups per CU: while one multiplies, three gather.
#include "common.h"
#include "vecio.h"
#include <limits.h>
#include <type_traits>

namespace mp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FusedArgs {
  const int32_t* rowptr; const int32_t* col; const float* val;
  int32_t N;
  const float* X; int64_t ldx;
  const float* S; int64_t lds; float self_scale;
  const float* Wm; int64_t ldw;
  const float* bias; int32_t act;
  float* P; int64_t ldp;
  float* out; int64_t ldo; int32_t dout;
  int32_t out_vec4;   // out rows allow 16-byte stores
  int32_t mean;   // rows are divided by their entry count (applied to the saved P rows and in the output epilogue)
};

constexpr int kTileRows = 32;

// VAR bits (mp_fused_config, tuning experiments): 1 = non-temporal stores of out; W fragments fetched
// 6 (bit 2), 4 (bit 16), 2 (bit 32) K groups ahead instead of 1
template <int W, bool WEIGHTED, int U, int VAR>
__global__ __launch_bounds__(kBlock, 4) void agg_dense_kernel(FusedArgs a) {
  constexpr bool NT_OUT = VAR & 1;
  constexpr int PF = (VAR & 2) ? 6 : ((VAR & 16) ? 4 : ((VAR & 32) ? 2 : 1));
  constexpr bool SKIP_MFMA = VAR & 4;     // timing diagnostics only (results are wrong): phase A alone
  constexpr bool SKIP_GATHER = VAR & 8;   //                                              phase B alone
  constexpr bool SPLIT = VAR & 128;       //   even workgroups run phase A only, odd ones phase B only
  constexpr bool NO_BMEM = VAR & 256;     //   phase B without its W loads and output stores (registers only)
  constexpr int F = kWave * W;
  constexpr int LDT = F + 4;   // row stride of the tile: 16-byte aligned rows, conflict-free b128 fragment reads
  __shared__ __attribute__((aligned(16))) float T[kTileRows][LDT];
  __shared__ __attribute__((aligned(16))) float carry[kWavesPerBlock - 1][F];
  __shared__ int carry_row[kWavesPerBlock];
  __shared__ float inv_deg[kTileRows];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int R0 = (SPLIT ? blockIdx.x >> 1 : blockIdx.x) * kTileRows;
  const bool role_b = SPLIT && (blockIdx.x & 1);   // wave-uniform
  const int R1 = min(R0 + kTileRows, a.N);

  // ---- init: T = self_scale * S rows (or zeros; rows past N stay zero) ----
  {
    constexpr int VPR = F / 4;                 // float4 per row
    for (int i = tid; i < kTileRows * VPR; i += kBlock) {
      const int m = i / VPR, c = (i % VPR) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a.S != nullptr && R0 + m < R1) {
        v = *reinterpret_cast<const f32x4*>(a.S + (int64_t)(R0 + m) * a.lds + c);
        v *= a.self_scale;
      }
      *reinterpret_cast<f32x4*>(&T[m][c]) = v;
    }
  }

  // ---- phase A: this wave's run of entries ----
  // lane i (<= 32) holds the start of tile row i (rows past the end of the matrix are empty)
  const int rp_v = lane <= kTileRows ? a.rowptr[min(R0 + lane, R1)] : INT_MAX;
  if (a.mean && wave == 0) {   // 1 / (entries of the row): lane i sees the starts of rows i and i + 1
    const int nxt = __shfl_down(rp_v, 1, kWave);
    if (lane < kTileRows) inv_deg[lane] = nxt > rp_v ? 1.0f / (float)(nxt - rp_v) : 0.f;
  }
  const int E0 = bcast_i(rp_v, 0);
  const int E1 = bcast_i(rp_v, kTileRows);
  const int q = (E1 - E0 + kWavesPerBlock - 1) / kWavesPerBlock;
  const int es = min(E0 + wave * q, E1);
  const int ee = min(es + q, E1);
  int first_rl = -1;
  bool cont = false;
  if (es < ee) {
    const unsigned long long started = __ballot(lane >= 1 && lane <= kTileRows && rp_v <= es);
    first_rl = __builtin_amdgcn_readfirstlane((int)__popcll(started));
    cont = bcast_i(rp_v, first_rl) < es;
  }
  if (lane == 0) carry_row[wave] = cont ? first_rl : -1;
  __syncthreads();   // T initialised

  if (es < ee && !SKIP_GATHER && !role_b) {
    const float* __restrict__ xlane = a.X + lane * W;
    int rl = first_rl;
    int rend = bcast_i(rp_v, rl + 1);
    float acc[W];
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k] = 0.f;

    auto flush = [&]() {
      if (cont && rl == first_rl) {
        store_vec<W>(&carry[wave - 1][lane * W], acc);
      } else {
        float t[W];
        load_vec<W>(&T[rl][lane * W], t);
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] += acc[k];
        store_vec<W>(&T[rl][lane * W], t);
      }
#pragma unroll
      for (int k = 0; k < W; ++k) acc[k] = 0.f;
    };

    for (int ec = es; ec < ee; ec += kWave) {
      const int me = min(ec + lane, ee - 1);
      const int cv = a.col[me];
      float wv = 1.f;
      if (WEIGHTED) wv = a.val[me];
      const int n = min(kWave, ee - ec);
      for (int jb = 0; jb < n; jb += U) {
        float v[U][W];
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const int c = bcast_i(cv, jb + j);
          load_vec<W>(xlane + (int64_t)c * a.ldx, v[j]);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const int e = ec + jb + j;
          if (e < ee) {
            while (e >= rend) {
              flush();
              rl += 1;
              rend = bcast_i(rp_v, rl + 1);
            }
            const float w = WEIGHTED ? bcast_f(wv, jb + j) : 1.f;
#pragma unroll
            for (int k = 0; k < W; ++k) acc[k] = fmaf(w, v[j][k], acc[k]);
          }
        }
      }
    }
    flush();
  }
  __syncthreads();

  // ---- carries: a row cut by run boundaries gets its later parts in wave order ----
  if (tid < F) {
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) {
      const int cr = carry_row[w];
      if (cr >= 0) T[cr][tid] += carry[w - 1][tid];
    }
  }
  __syncthreads();

  if (a.P != nullptr) {   // the aggregated rows, kept for the weight gradient
    constexpr int VPR = F / 4;
    for (int i = tid; i < kTileRows * VPR; i += kBlock) {
      const int m = i / VPR, c = (i % VPR) * 4;
      if (R0 + m < R1) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&T[m][c]);
        if (a.mean) v *= inv_deg[m];
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.P + (int64_t)(R0 + m) * a.ldp + c));
      }
    }
  }

  // ---- phase B: [32 x F] tile x W[F x dout] on the matrix cores ----
  // K is walked in groups of 8: hardware k-slot kk (= lane >> 5) of MFMA j takes k = 8 g + 4 kk + j, so a
  // lane's four A values are one 16-byte LDS read and its B values are four rows of W.  The wave's two
  // 32-column accumulator tiles interleave columns (tile t holds columns n0 + 2 n + t): one 8-byte load
  // feeds both tiles and every output row is stored as 256 contiguous bytes per half-wave.
  const int fr = lane & 31, kk = lane >> 5;
  if (SKIP_MFMA || (SPLIT && !role_b)) {
    if (T[fr][kk] == 12345.678f) a.out[0] = 1.f;   // keep phase A alive
    return;
  }
  for (int cb = 0; cb < a.dout; cb += 64 * kWavesPerBlock) {
    const int n0 = cb + wave * 64;
    if (n0 >= a.dout) break;                       // wave-uniform
    const int cpair = n0 + 2 * fr;
    const bool col_ok = cpair < a.dout;
    const float* __restrict__ wp = a.Wm + (int64_t)(4 * kk) * a.ldw + (col_ok ? cpair : a.dout - 2);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // ring of PF + 1 register slots: the fragments of group g + PF are requested before the MFMAs of group g.
    // W comes from L2, but under the gather traffic of the other workgroups an L2 hit takes on the order of a
    // microsecond while one group's MFMAs take 0.2 us, so the distance has to cover several groups.
    f32x2 bq[PF + 1][4];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) bq[p][j] = *reinterpret_cast<const f32x2*>(wp + (int64_t)(8 * p + j) * a.ldw);
#pragma unroll
    for (int g = 0; g < F / 8; ++g) {      // fully unrolled: every slot index is a constant
      const int cur = g % (PF + 1), nxt = (g + PF) % (PF + 1);
      if (g + PF < F / 8 && !NO_BMEM) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bq[nxt][j] = *reinterpret_cast<const f32x2*>(wp + (int64_t)(8 * (g + PF) + j) * a.ldw);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the MFMAs (the scheduler would sink them)
      const f32x4 av = *reinterpret_cast<const f32x4*>(&T[fr][8 * g + 4 * kk]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bq[cur][j][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bq[cur][j][1], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x2 bv = {0.f, 0.f};
    if (a.bias != nullptr && col_ok) bv = *reinterpret_cast<const f32x2*>(a.bias + cpair);
    auto finish = [&](int r, int rl) {
      const float sc = a.mean ? inv_deg[rl] : 1.f;
      f32x2 o = {fmaf(acc0[r], sc, bv[0]), fmaf(acc1[r], sc, bv[1])};
      if (a.act == MP_ACT_RELU) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); }
      return o;
    };
    if constexpr (NO_BMEM) {
      if (acc0[0] + acc1[5] == 12345.678f) a.out[1] = 1.f;
      continue;
    }
    if (a.out_vec4) {
      // 16-byte stores: neighbouring lanes swap one row's pair, the even lane stores row r (its two columns and
      // the neighbour's two), the odd lane row r + 1
      const bool odd = fr & 1;
      const int c4 = n0 + 2 * (fr & ~1);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * kk;       // rows rl, rl + 1
        const f32x2 m0 = finish(r, rl), m1 = finish(r + 1, rl + 1);
        const f32x2 send = odd ? m0 : m1;
        f32x2 recv;
        recv[0] = __shfl_xor(send[0], 1, kWave);
        recv[1] = __shfl_xor(send[1], 1, kWave);
        const f32x4 o4 = odd ? f32x4{recv[0], recv[1], m1[0], m1[1]} : f32x4{m0[0], m0[1], recv[0], recv[1]};
        const int row = R0 + rl + (odd ? 1 : 0);
        if (row < R1) {
          float* dst = a.out + (int64_t)row * a.ldo + c4;
          if (c4 + 3 < a.dout) {
            if constexpr (NT_OUT) __builtin_nontemporal_store(o4, reinterpret_cast<f32x4*>(dst));
            else *reinterpret_cast<f32x4*>(dst) = o4;
          } else if (c4 < a.dout) {
            *reinterpret_cast<f32x2*>(dst) = f32x2{o4[0], o4[1]};
          }
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * kk;
        const int row = R0 + rl;
        const f32x2 o = finish(r, rl);
        if (col_ok && row < R1) *reinterpret_cast<f32x2*>(a.out + (int64_t)row * a.ldo + cpair) = o;
      }
    }
  }
}

static int g_fused_var = 32;
static int g_fused_u = 8;
static int g_fused_no_vec4 = 0;

template <int W, int U, int VAR>
static int launch_fused_v(const FusedArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)ceil_div(a.N, kTileRows) * ((VAR & 128) ? 2 : 1)), block(kBlock);
  if (a.val) hipLaunchKernelGGL((agg_dense_kernel<W, true, U, VAR>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((agg_dense_kernel<W, false, U, VAR>), grid, block, 0, st, a);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

template <int W>
static int launch_fused(const FusedArgs& a, hipStream_t st) {
  if constexpr (W == 4) {   // the tuning variants exist for the 256-wide case only
    if (g_fused_u == 4) return launch_fused_v<4, 4, 32>(a, st);
    if (g_fused_u == 16) return launch_fused_v<4, 16, 32>(a, st);
    switch (g_fused_var) {
      case 1: return launch_fused_v<4, 8, 1>(a, st);
      case 33: return launch_fused_v<4, 8, 33>(a, st);
      case 2: return launch_fused_v<4, 8, 2>(a, st);
      case 4: return launch_fused_v<4, 8, 4>(a, st);
      case 16: return launch_fused_v<4, 8, 16>(a, st);
      case 32: return launch_fused_v<4, 8, 32>(a, st);
      case 36: return launch_fused_v<4, 8, 36>(a, st);
      case 40: return launch_fused_v<4, 8, 40>(a, st);
      case 160: return launch_fused_v<4, 8, 160>(a, st);
      case 416: return launch_fused_v<4, 8, 416>(a, st);
      case 296: return launch_fused_v<4, 8, 296>(a, st);
      case 8: return launch_fused_v<4, 8, 8>(a, st);
      default: break;
    }
  }
  return launch_fused_v<W, 8, 0>(a, st);
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_fused_config(int rows_in_flight, int variant_bits) {
  if ((rows_in_flight != 4 && rows_in_flight != 8 && rows_in_flight != 16) || variant_bits < 0 || variant_bits > 511) return MP_ERR_INVALID_ARG;
  g_fused_u = rows_in_flight;
  g_fused_no_vec4 = (variant_bits & 64) ? 1 : 0;
  g_fused_var = variant_bits & ~64;
  return MP_OK;
}

int mp_agg_dense_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int reduce,
                     const float* X, int64_t ldx, int32_t F, const float* S, int64_t lds, float self_scale, const float* W,
                     int64_t ldw, int32_t d_out, const float* bias, int act, float* P, int64_t ldp, float* out,
                     int64_t ldo, mp_stream_t stream) {
  if (!rowptr || !X || !W || !out || N < 0 || F <= 0 || d_out <= 0) return MP_ERR_INVALID_ARG;
  if (ldx < F || ldw < d_out || ldo < d_out || (S && lds < F) || (P && ldp < F)) return MP_ERR_INVALID_ARG;
  if (act != MP_ACT_NONE && act != MP_ACT_RELU) return MP_ERR_INVALID_ARG;
  if (reduce != MP_SUM && reduce != MP_MEAN) return MP_ERR_INVALID_ARG;
  if (reduce == MP_MEAN && S) return MP_ERR_INVALID_ARG;
  if (F != 64 && F != 128 && F != 256) return MP_ERR_UNSUPPORTED;
  if (d_out % 2) return MP_ERR_UNSUPPORTED;
  if (N >= INT32_MAX - kTileRows) return MP_ERR_UNSUPPORTED;
  const int w = F / kWave;
  auto mis = [](const void* p, int64_t ld, int bytes) { return ((uintptr_t)p % bytes) || ((ld * 4) % bytes); };
  if (mis(X, ldx, 4 * w) || (S && mis(S, lds, 16)) || (P && mis(P, ldp, 16)) || mis(W, ldw, 8) || mis(out, ldo, 8) ||
      (bias && ((uintptr_t)bias % 8)))
    return MP_ERR_ALIGNMENT;
  if (N == 0) return MP_OK;
  if (!col) return MP_ERR_INVALID_ARG;
  FusedArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.N = (int32_t)N;
  a.X = X; a.ldx = ldx; a.S = S; a.lds = lds; a.self_scale = self_scale;
  a.Wm = W; a.ldw = ldw; a.bias = bias; a.act = act; a.P = P; a.ldp = ldp; a.out = out; a.ldo = ldo; a.dout = d_out; a.mean = reduce == MP_MEAN;
  a.out_vec4 = !g_fused_no_vec4 && !mis(out, ldo, 16);
  switch (w) {
    case 4: return launch_fused<4>(a, as_stream(stream));
    case 2: return launch_fused<2>(a, as_stream(stream));
    default: return launch_fused<1>(a, as_stream(stream));
  }
}

}  // extern "C"

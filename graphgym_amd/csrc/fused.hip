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
//
// F = 512 (config C5's width) runs as two K halves over the same 32-row tile: gather columns 0..255 of the
// neighbour rows, multiply by W[0:256], gather columns 256..511, multiply by W[256:512] into the same
// accumulators (the accumulators of all output column blocks stay in registers: d_out <= 512).
//
// The identity branch of the ID layers (out = A (X W + S X W_id), TfgIDLayer.py:510-517, idconv.py:150-177)
// needs no second tile here: A S X W_id = A_id Z with Z = X[id] W_id (n_id rows, a small product) and A_id the
// entries whose source is an identity node.  The main kernel leaves rows that own such an entry un-activated
// (`defer_act`), and id_fixup_kernel adds A_id Z to exactly those rows and applies the activation.
#include "common.h"
#include <map>
#include <mutex>
#include <utility>
#include "vecio.h"
#include <atomic>
#include <limits.h>
#include <type_traits>

namespace mp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct FusedArgs {
  const int32_t* rowptr; const int32_t* col; const float* val;
  int32_t N;
  const float* X; int64_t ldx;
  const float* S; int64_t lds; float self_scale;
  const float* Wm; int64_t ldw;
  const __bf16* Wsp; int64_t ldws;   // BF16X3: W split three ways into bf16, [3][F / 8][dout][8], ldws = F; see mfma_half_bf16x3
  const float* bias; int32_t act;
  const uint8_t* defer_act;   // [N] or NULL: rows with a nonzero flag are stored without the activation
  float* P; int64_t ldp;
  float* out; int64_t ldo; int32_t dout;
  const float* R; int64_t ldr;   // optional residual added before the activation: out = act(acc + bias + R)
  int32_t out_vec4;   // out rows allow 16-byte stores
  int32_t mean;   // rows are divided by their entry count (applied to the saved P rows and in the output epilogue)
  // AGG_ONLY, two-branch form (mp_idgnn_agg_tiles_f32): the identity branch's rows Q — written as zeros here except the
  // rows flagged in defer_act, which mp_id_rows_f32 writes from their few identity entries
  float* Q = nullptr; int64_t ldq = 0;
};

constexpr int kTileRows = 32;
// v_max_f32 as it is: fmaxf() in a loop costs a second v_max (x, x) per operand — the compiler cannot prove a
// loop-carried maximum canonical and quiets it every time (three instructions per term instead of two in the gather
// loop of the max aggregation: 20.4 -> 19.9 ms).  A NaN term never wins, as in spmm.hip's `if (m > a)`.
__device__ __forceinline__ float max_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// -DMP_FUSED_TIMING: s_memtime stamps of the phases of sampled tiles (one tile in 128), read back with mp_debug_read by
// scripts/dbg/fused_phases.py — how profiles/r03_fused_phases.json was made.  Not part of the product build.
#ifdef MP_FUSED_TIMING
__device__ long long g_dbg[1 << 18];
#define DBG_T(slot) do { if (dbg_on && lane == 0) g_dbg[dbg_base + wave * 16 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
// (producer/consumer kernel: workgroups 5, 37, 69, ... stamp their first 1000 items: producer wave 0 in slots 0-3, the
// first consumer wave in slots 4-7)
#define PC_T(slot) do { if ((blockIdx.x & 31) == 5 && (blockIdx.x >> 5) < 8 && it < 1000 && lane == 0) \
  g_dbg[(((blockIdx.x >> 5) * 1000 + it) << 3) + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define DBG_T(slot) do {} while (0)
#define PC_T(slot) do {} while (0)
#endif
#ifndef MP_W_RING
#define MP_W_RING 2    // K groups the W fragments of the 64-row kernel are fetched ahead
#endif
#ifndef MP_FUSED_U
#define MP_FUSED_U 16   // neighbour rows in flight per wave in phase A
#endif

// one K half of phase B: acc += T[32 x FH] * W[k0 : k0 + FH, 64 columns of this wave]
// K is walked in groups of 8: hardware k-slot kk (= lane >> 5) of MFMA j takes k = 8 g + 4 kk + j, so a
// lane's four A values are one 16-byte LDS read and its B values are four rows of W.  The wave's two
// 32-column accumulator tiles interleave columns (tile t holds columns n0 + 2 n + t): one 8-byte load
// feeds both tiles and every output row is stored as 256 contiguous bytes per half-wave.
template <int FH, int PF>
__device__ __forceinline__ void mfma_half(const float (*T)[FH + 4], const float* __restrict__ wp, int64_t ldw,
                                          f32x16& acc0, f32x16& acc1, int fr, int kk) {
  // ring of PF + 1 register slots: the fragments of group g + PF are requested before the MFMAs of group g.
  // W comes from L2, but under the gather traffic of the other workgroups an L2 hit takes on the order of a
  // microsecond while one group's MFMAs take 0.2 us, so the distance has to cover several groups.
  f32x2 bq[PF + 1][4];
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[p][j] = *reinterpret_cast<const f32x2*>(wp + (int64_t)(8 * p + j) * ldw);
#pragma unroll
  for (int g = 0; g < FH / 8; ++g) {      // fully unrolled: every slot index is a constant
    const int cur = g % (PF + 1), nxt = (g + PF) % (PF + 1);
    if (g + PF < FH / 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bq[nxt][j] = *reinterpret_cast<const f32x2*>(wp + (int64_t)(8 * (g + PF) + j) * ldw);
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
}

// The same K half on the bf16 matrix pipe, fp32-accurate: every operand is split three ways, x = x0 + x1 + x2 with
// x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (24 mantissa bits in all), and the product keeps the six
// terms down to 2^-24 of the result: x0 y0 + x0 y1 + x1 y0 + x0 y2 + x1 y1 + x2 y0, each a v_mfma_f32_32x32x16_bf16
// accumulating in fp32 (a bf16 x bf16 product is exact in fp32).  gfx950 runs f32 MFMA at 1/16 of the bf16 rate, so six
// bf16 MFMAs of K = 16 replace eight f32 MFMAs of K = 2 at 3/8 of the cycles: the layer's MFMA cycles, which add almost
// one for one to its gather time, shrink 2.7x.  A is split on the fly from the fp32 LDS tile (VALU work
// that hides behind the MFMAs); W arrives pre-split in the layout [3][F / 8][dout][8] bf16: the 8 k-values a lane feeds to
// one MFMA are 16 contiguous bytes, and neighbouring columns are neighbours in memory, so a half-wave's loads of its two
// interleaved column tiles cover 1 KiB contiguously (a [dout][F] layout, k contiguous per column, put every lane on its
// own cache line: 8x over-fetch from L2 and the kernel ran 1.4x SLOWER than the f32 form).  Fetched one group ahead.
template <int FH>
__device__ __forceinline__ void mfma_half_bf16x3(const float (*T)[FH + 4], const __bf16* __restrict__ w0,
                                                 int64_t gstride, int64_t plane, f32x16& acc0, f32x16& acc1, int fr,
                                                 int kk) {
  // w0: this lane's column pair (tile 0 = first 8 values, tile 1 = the next 8) in split plane 0 at K group 0 of this
  // half; `gstride` = 2 * dout * 8 elements per K = 16 group; `plane` = dout * F elements per split plane.
  const __bf16* __restrict__ w1 = w0 + 8;
  // One register slot per (tile, plane); a plane's registers are refilled for the next K group as soon as its last MFMA
  // of this group has been issued (plane 2 is used once, plane 1 twice, plane 0 three times, in that order), so the
  // loads run one group ahead without a second buffer (a double buffer spills at four waves per SIMD).
  bf16x8 bq[2][3];
  auto fetch = [&](int sp, int g) {
#if defined(MP_ABL_W_L1)      // (ablation builds: every K group reads group 0 / plane 2 re-reads plane 1)
    g = 0;
#elif defined(MP_ABL_W_2P)
    if (sp == 2) { bq[0][2] = bq[0][1]; bq[1][2] = bq[1][1]; return; }
#endif
    bq[0][sp] = *reinterpret_cast<const bf16x8*>(w0 + sp * plane + g * gstride);
    bq[1][sp] = *reinterpret_cast<const bf16x8*>(w1 + sp * plane + g * gstride);
  };
  fetch(2, 0); fetch(1, 0); fetch(0, 0);
#pragma unroll
  for (int g = 0; g < FH / 16; ++g) {
    const bool more = g + 1 < FH / 16;
    const f32x4 alo = *reinterpret_cast<const f32x4*>(&T[fr][16 * g + 8 * kk]);
    const f32x4 ahi = *reinterpret_cast<const f32x4*>(&T[fr][16 * g + 8 * kk + 4]);
    bf16x8 a0, a1, a2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = i < 4 ? alo[i] : ahi[i - 4];
      const __bf16 b0 = (__bf16)x;
      const float r1 = x - (float)b0;
      const __bf16 b1 = (__bf16)r1;
      const float r2 = r1 - (float)b1;
      a0[i] = b0; a1[i] = b1; a2[i] = (__bf16)r2;
    }
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[0][2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[1][2], acc1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(2, g + 1);
    __builtin_amdgcn_sched_barrier(0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[0][1], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[1][1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[0][1], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[1][1], acc1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(1, g + 1);
    __builtin_amdgcn_sched_barrier(0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bq[0][0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bq[1][0], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[0][0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[1][0], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[0][0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[1][0], acc1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(0, g + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// W: floats per lane of one K half (half width FH = 64 W); KH: K halves (F = KH * FH); NCB: output column blocks of
// 256 whose accumulators stay live across the halves (KH == 2 only; KH == 1 walks the blocks one after another);
// PF: W fragments fetched PF K-groups ahead; NT_OUT: non-temporal stores of out (on: -0.3 % in an
// in-process A/B, scripts/dbg/fused_ab.py)
template <int W, bool WEIGHTED, int U, int KH, int NCB, int PF, bool NT_OUT, bool BF16X3>
__global__ __launch_bounds__(kBlock, KH == 2 ? 3 : 4) void agg_dense_kernel(FusedArgs a) {
  constexpr int FH = kWave * W;
  constexpr int LDT = FH + 4;   // row stride of the tile: 16-byte aligned rows, conflict-free b128 fragment reads
  __shared__ __attribute__((aligned(16))) float T[kTileRows][LDT];
  __shared__ __attribute__((aligned(16))) float carry[kWavesPerBlock - 1][FH];
  __shared__ int carry_row[kWavesPerBlock];
  __shared__ float inv_deg[kTileRows];
  __shared__ int defer_l[kTileRows];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // A workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (kFusedMaxGrid workgroups at most: ~5 tiles each at
  // 10^7 rows).  Fewer, longer-lived workgroups: -1.7 % (out only) / -4 % (aggregated rows kept) against one workgroup
  // per tile in an in-process A/B; a smaller grid loses to the hub tiles at the head of the matrix (1024 workgroups:
  // +14 %).  Tiles are independent and a tile's summation order does not depend on who computes it: same bits.
  // (KH == 2 — F = 512 — keeps one tile per workgroup: with the tile loop its 164 registers spill)
  for (int tile = blockIdx.x; (int64_t)tile * kTileRows < a.N; tile += gridDim.x) {
  const int R0 = tile * kTileRows;
  const int R1 = min(R0 + kTileRows, a.N);
#ifdef MP_FUSED_TIMING
  const bool dbg_on = (tile % 128) == 7 && tile / 128 < 2048;
  const int dbg_base = (tile / 128) * 64;
#endif
  DBG_T(0);

  // ---- the wave's run of entries (the same for every K half) ----
  // lane i (<= 32) holds the start of tile row i (rows past the end of the matrix are empty)
  const int rp_v = lane <= kTileRows ? a.rowptr[min(R0 + lane, R1)] : INT_MAX;
  if (wave == 0) {   // 1 / (entries of the row): lane i sees the starts of rows i and i + 1
    const int nxt = __shfl_down(rp_v, 1, kWave);
    if (lane < kTileRows) {
      inv_deg[lane] = (a.mean && nxt > rp_v) ? 1.0f / (float)(nxt - rp_v) : (a.mean ? 0.f : 1.f);
      defer_l[lane] = (a.defer_act != nullptr && R0 + lane < R1) ? (int)a.defer_act[R0 + lane] : 0;
    }
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

  const int fr = lane & 31, kk = lane >> 5;
  const int fr_c = fr, kk_c = kk;
  f32x16 acc[KH == 2 ? NCB : 1][2];
  if constexpr (KH == 2) {
#pragma unroll
    for (int b = 0; b < NCB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[b][0][r] = 0.f; acc[b][1][r] = 0.f; }
  }

  // epilogue of one 64-column block of this wave: bias, 1/deg for mean, activation, store
  auto store_block = [&](const f32x16& acc0, const f32x16& acc1, int n0) {
    // the lane coordinates are re-derived here from opaque copies that depend on an accumulator value: the compiler
    // otherwise computes the 16 - 20 row pointers of this epilogue BEFORE phase B, runs out of registers there and
    // spills them — 41 KB of scratch written and re-read per 32-row tile, 12 GB per launch at 10^7 rows
    // (rocprofv3 WRITE_SIZE 22.5 GB for a 10.2 GB output)
    int fr = fr_c, kk = kk_c;
    asm volatile("" : "+v"(fr), "+v"(kk) : "v"(acc0[0]), "v"(acc1[15]));
    const int cpair = n0 + 2 * fr;
    const bool col_ok = cpair < a.dout;
    f32x2 bv = {0.f, 0.f};
    if (a.bias != nullptr && col_ok) bv = *reinterpret_cast<const f32x2*>(a.bias + cpair);
    auto finish = [&](int r, int rl) {
      const float sc = inv_deg[rl];
      f32x2 o = {fmaf(acc0[r], sc, bv[0]), fmaf(acc1[r], sc, bv[1])};
      if (a.R != nullptr && col_ok && R0 + rl < R1) {
        const f32x2 rv = *reinterpret_cast<const f32x2*>(a.R + (int64_t)(R0 + rl) * a.ldr + cpair);
        o[0] += rv[0]; o[1] += rv[1];
      }
      if (a.act == MP_ACT_RELU && !defer_l[rl]) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); }
      return o;
    };
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
  };

#pragma unroll
  for (int kh = 0; kh < KH; ++kh) {
    const int k0 = kh * FH;   // first feature column of this half
    // ---- init: T = self_scale * S rows (or zeros; rows past N stay zero) ----
    {
      constexpr int VPR = FH / 4;                 // float4 per row
      for (int i = tid; i < kTileRows * VPR; i += kBlock) {
        const int m = i / VPR, c = (i % VPR) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a.S != nullptr && R0 + m < R1) {
          v = *reinterpret_cast<const f32x4*>(a.S + (int64_t)(R0 + m) * a.lds + k0 + c);
          v *= a.self_scale;
        }
        *reinterpret_cast<f32x4*>(&T[m][c]) = v;
      }
    }
    DBG_T(1);
    __syncthreads();   // T initialised (and carry_row / inv_deg / defer_l visible)
    DBG_T(2);

    // ---- phase A: this wave's run of entries, feature columns [k0, k0 + FH) ----
    if (es < ee) {
      const float* __restrict__ xlane = a.X + k0 + lane * W;
      int rl = first_rl;
      int rend = bcast_i(rp_v, rl + 1);
      float accr[W];
#pragma unroll
      for (int k = 0; k < W; ++k) accr[k] = 0.f;

      auto flush = [&]() {
        if (cont && rl == first_rl) {
          store_vec<W>(&carry[wave - 1][lane * W], accr);
        } else {
          float t[W];
          load_vec<W>(&T[rl][lane * W], t);
#pragma unroll
          for (int k = 0; k < W; ++k) t[k] += accr[k];
          store_vec<W>(&T[rl][lane * W], t);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) accr[k] = 0.f;
      };

      for (int ec = es; ec < ee; ec += kWave) {
        const int me = min(ec + lane, ee - 1);
        const int cv = a.col[me] & 0x7fffffff;   // an identity mark (sign bit) is not part of the index
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
              for (int k = 0; k < W; ++k) accr[k] = fmaf(w, v[j][k], accr[k]);
            }
          }
        }
      }
      flush();
    }
    DBG_T(3);
    __syncthreads();
    DBG_T(4);

    // ---- carries: a row cut by run boundaries gets its later parts in wave order ----
    if (tid < FH) {
#pragma unroll
      for (int w = 1; w < kWavesPerBlock; ++w) {
        const int cr = carry_row[w];
        if (cr >= 0) T[cr][tid] += carry[w - 1][tid];
      }
    }
    __syncthreads();

    if (a.P != nullptr) {   // the aggregated rows, kept for the weight gradient
      constexpr int VPR = FH / 4;
      for (int i = tid; i < kTileRows * VPR; i += kBlock) {
        const int m = i / VPR, c = (i % VPR) * 4;
        if (R0 + m < R1) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&T[m][c]);
          v *= inv_deg[m];
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.P + (int64_t)(R0 + m) * a.ldp + k0 + c));
        }
      }
    }

    DBG_T(5);
    // ---- phase B: [32 x FH] tile x W[k0 : k0 + FH, :] on the matrix cores ----
    if constexpr (KH == 1) {
      for (int cb = 0; cb < a.dout; cb += 64 * kWavesPerBlock) {
        const int n0 = cb + wave * 64;
        if (n0 >= a.dout) break;                       // wave-uniform
        const int cpair = n0 + 2 * fr;
        const int ccol = cpair < a.dout ? cpair : a.dout - 2;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        if constexpr (BF16X3) {
          const __bf16* w0 = a.Wsp + ((int64_t)kk * a.dout + ccol) * 8;
          mfma_half_bf16x3<FH>(T, w0, (int64_t)a.dout * 16, (int64_t)a.dout * a.ldws, acc0, acc1, fr, kk);
        } else {
          const float* __restrict__ wp = a.Wm + (int64_t)(4 * kk) * a.ldw + ccol;
          mfma_half<FH, PF>(T, wp, a.ldw, acc0, acc1, fr, kk);
        }
        DBG_T(6);
        store_block(acc0, acc1, n0);
        DBG_T(7);
      }
    } else {
#pragma unroll
      for (int b = 0; b < NCB; ++b) {
        const int n0 = b * 64 * kWavesPerBlock + wave * 64;
        if (n0 < a.dout) {                             // wave-uniform
          const int cpair = n0 + 2 * fr;
          const int ccol = cpair < a.dout ? cpair : a.dout - 2;
          if constexpr (BF16X3) {
            const __bf16* w0 = a.Wsp + ((int64_t)(k0 / 8 + kk) * a.dout + ccol) * 8;
            mfma_half_bf16x3<FH>(T, w0, (int64_t)a.dout * 16, (int64_t)a.dout * a.ldws, acc[b][0], acc[b][1], fr, kk);
          } else {
            const float* __restrict__ wp = a.Wm + (int64_t)(k0 + 4 * kk) * a.ldw + ccol;
            mfma_half<FH, PF>(T, wp, a.ldw, acc[b][0], acc[b][1], fr, kk);
          }
        }
      }
      if (kh + 1 < KH) __syncthreads();   // every wave has read T before the next half re-initialises it
    }
  }
  if constexpr (KH == 2) {
#pragma unroll
    for (int b = 0; b < NCB; ++b) {
      const int n0 = b * 64 * kWavesPerBlock + wave * 64;
      if (n0 < a.dout) store_block(acc[b][0], acc[b][1], n0);
    }
  }
  if constexpr (KH == 2) break;
  __syncthreads();   // every wave is done with T, carry_row, inv_deg before the next tile rewrites them
  }
}

// The same two products for RB 32-row blocks of one tile at once (the producer/consumer kernel below): every W fragment
// is fetched once and used RB times — the fragments, L2 hits, are what the product costs (see that kernel).
template <int FH, int RB, int NB>
__device__ __forceinline__ void mfma_rows_bf16x3(const float (*T)[FH + 4], const __bf16* __restrict__ w0,
                                                 int64_t bstride, int64_t gstride, int64_t plane,
                                                 f32x16 (&acc)[NB][RB][2], int fr, int kk) {
  // NB column blocks (bstride elements apart in W) walk K together: one split of the tile's values and one fetch round
  // trip per K group serve all of them (a wave with two blocks, one after the other, waits twice as often).
  bf16x8 bq[NB][2][3];
  auto fetch = [&](int sp, int g) {
#if defined(MP_ABL_W_L1)      // (ablation builds: every K group reads group 0 / plane 2 re-reads plane 1)
    g = 0;
#elif defined(MP_ABL_W_2P)
    if (sp == 2) {
#pragma unroll
      for (int n = 0; n < NB; ++n) { bq[n][0][2] = bq[n][0][1]; bq[n][1][2] = bq[n][1][1]; }
      return;
    }
#endif
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      bq[n][0][sp] = *reinterpret_cast<const bf16x8*>(w0 + n * bstride + sp * plane + g * gstride);
      bq[n][1][sp] = *reinterpret_cast<const bf16x8*>(w0 + n * bstride + 8 + sp * plane + g * gstride);
    }
  };
  fetch(2, 0); fetch(1, 0); fetch(0, 0);
#pragma unroll
  for (int g = 0; g < FH / 16; ++g) {
    const bool more = g + 1 < FH / 16;
    bf16x8 a0[RB], a1[RB], a2[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const f32x4 alo = *reinterpret_cast<const f32x4*>(&T[32 * b + fr][16 * g + 8 * kk]);
      const f32x4 ahi = *reinterpret_cast<const f32x4*>(&T[32 * b + fr][16 * g + 8 * kk + 4]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? alo[i] : ahi[i - 4];
        const __bf16 b0 = (__bf16)x;
        const float r1 = x - (float)b0;
        const __bf16 b1 = (__bf16)r1;
        const float r2 = r1 - (float)b1;
        a0[b][i] = b0; a1[b][i] = b1; a2[b][i] = (__bf16)r2;
      }
    }
    // (per accumulator the order of the six terms is that of mfma_half_bf16x3: same bits for every tile height)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        acc[n][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[n][0][2], acc[n][b][0], 0, 0, 0);
        acc[n][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[n][1][2], acc[n][b][1], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(2, g + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        acc[n][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[b], bq[n][0][1], acc[n][b][0], 0, 0, 0);
        acc[n][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[b], bq[n][1][1], acc[n][b][1], 0, 0, 0);
        acc[n][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[n][0][1], acc[n][b][0], 0, 0, 0);
        acc[n][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[n][1][1], acc[n][b][1], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(1, g + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        acc[n][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[b], bq[n][0][0], acc[n][b][0], 0, 0, 0);
        acc[n][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[b], bq[n][1][0], acc[n][b][1], 0, 0, 0);
        acc[n][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[b], bq[n][0][0], acc[n][b][0], 0, 0, 0);
        acc[n][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[b], bq[n][1][0], acc[n][b][1], 0, 0, 0);
        acc[n][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[n][0][0], acc[n][b][0], 0, 0, 0);
        acc[n][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[n][1][0], acc[n][b][1], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(0, g + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same with the W fragments in a ring of D + 1 slots, fetched D whole K groups ahead: behind the gathers of the CU's
// other waves a fragment (an L2 hit) takes ~1.5 us to arrive while a group's MFMAs take 0.3 us, and the rolling refill
// above runs at most one group ahead.  For the instantiations with registers to spare (K in one half, one block).
template <int FH, int RB, int D>
__device__ __forceinline__ void mfma_rows_bf16x3_ring(const float (*T)[FH + 4], const __bf16* __restrict__ w0,
                                                      int64_t gstride, int64_t plane, f32x16 (&acc)[1][RB][2], int fr, int kk) {
  constexpr int G = FH / 16;
  bf16x8 bq[D + 1][2][3];
  auto fetch = [&](int slot, int g) {
#if defined(MP_ABL_W_L1)
    g = 0;
#endif
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
      bq[slot][0][sp] = *reinterpret_cast<const bf16x8*>(w0 + sp * plane + g * gstride);
      bq[slot][1][sp] = *reinterpret_cast<const bf16x8*>(w0 + 8 + sp * plane + g * gstride);
    }
  };
#pragma unroll
  for (int g = 0; g < D && g < G; ++g) fetch(g, g);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int cur = g % (D + 1);
    if (g + D < G) fetch((g + D) % (D + 1), g + D);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 a0[RB], a1[RB], a2[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const f32x4 alo = *reinterpret_cast<const f32x4*>(&T[32 * b + fr][16 * g + 8 * kk]);
      const f32x4 ahi = *reinterpret_cast<const f32x4*>(&T[32 * b + fr][16 * g + 8 * kk + 4]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? alo[i] : ahi[i - 4];
        const __bf16 b0 = (__bf16)x;
        const float r1 = x - (float)b0;
        const __bf16 b1 = (__bf16)r1;
        const float r2 = r1 - (float)b1;
        a0[b][i] = b0; a1[b][i] = b1; a2[b][i] = (__bf16)r2;
      }
    }
    // (the order of the six terms per accumulator is that of mfma_half_bf16x3: same bits)
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      acc[0][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[cur][0][2], acc[0][b][0], 0, 0, 0);
      acc[0][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[cur][1][2], acc[0][b][1], 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      acc[0][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[b], bq[cur][0][1], acc[0][b][0], 0, 0, 0);
      acc[0][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[b], bq[cur][1][1], acc[0][b][1], 0, 0, 0);
      acc[0][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[cur][0][1], acc[0][b][0], 0, 0, 0);
      acc[0][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[cur][1][1], acc[0][b][1], 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      acc[0][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[b], bq[cur][0][0], acc[0][b][0], 0, 0, 0);
      acc[0][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[b], bq[cur][1][0], acc[0][b][1], 0, 0, 0);
      acc[0][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[b], bq[cur][0][0], acc[0][b][0], 0, 0, 0);
      acc[0][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[b], bq[cur][1][0], acc[0][b][1], 0, 0, 0);
      acc[0][b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[cur][0][0], acc[0][b][0], 0, 0, 0);
      acc[0][b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[b], bq[cur][1][0], acc[0][b][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int FH, int PF, int RB>
__device__ __forceinline__ void mfma_rows(const float (*T)[FH + 4], const float* __restrict__ wp, int64_t ldw,
                                          f32x16 (&acc)[RB][2], int fr, int kk) {
  f32x2 bq[PF + 1][4];
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[p][j] = *reinterpret_cast<const f32x2*>(wp + (int64_t)(8 * p + j) * ldw);
#pragma unroll
  for (int g = 0; g < FH / 8; ++g) {
    const int cur = g % (PF + 1), nxt = (g + PF) % (PF + 1);
    if (g + PF < FH / 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bq[nxt][j] = *reinterpret_cast<const f32x2*>(wp + (int64_t)(8 * (g + PF) + j) * ldw);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(&T[32 * b + fr][8 * g + 4 * kk]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bq[cur][j][0], acc[b][0], 0, 0, 0);
        acc[b][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bq[cur][j][1], acc[b][1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same layer with the two phases on DIFFERENT waves (round 3; DESIGN.md §4.5 "Round 3").  Measured on the kernel
// above (profiles/r03_fused_phases.json): a workgroup gathers only 60 % of its time — phase B is 24 % (its W fragments
// are L2 hits, but the vector memory path returns in order, so behind the gathers of the CU's other waves each of the 16
// fetch rounds costs ~1 us), the head of the tile and the output store 6 % each.  Here a workgroup is NP + NC waves
// walking tiles it draws from a counter, over TWO tile buffers in LDS:
//   waves 0 .. NP-1 (producers): gather their runs of item j into buffer b, add the carries
//   the NC other waves (consumers): store the aggregated rows of item j - 1, multiply buffer b ^ 1 by W, store the output
// An item is one K half of one tile of TR rows.  The roles run SEPARATE loops over the same item sequence and meet at two
// workgroup barriers per item (runs reduced / carries added); the consumers usually reach them first, so the gathers of
// a workgroup do not stop for a product or a store.  Tiles come from an atomic counter in ascending order: the hub tiles
// at the head of the matrix start first and a workgroup that holds one simply draws fewer tiles; which workgroup computes
// a tile does not enter its arithmetic (same bits every run; against the kernel above only rows cut by a run boundary
// differ, the cut positions depend on TR and NP).
//
// Shapes (dispatch: launch_fused below).  Every tile reads ALL of W from L2 — 6 bytes per weight in the three-way bf16
// split, 393 KB at F = dout = 256, more than a 32-row tile's own gathers (352 KB) — and that traffic is what the
// product costs: with W served from L1 the 32-row form runs at 20.0 ms, below the plain aggregation (ablation builds,
// profiles/r03_fused_ablation.json: 22.3 ms as is, 21.2 ms with one of the three planes not read, 19.7 ms without the
// product, 17.6 ms without product and store).  TR = 64 uses every W fragment for two 32-row MFMA blocks: half the
// traffic; its two buffers take 133 KB of LDS — one workgroup per CU, up to 256 registers per wave.  NP = 4 producers
// are the optimum there (2: 24.6 ms, 4: 20.97, 6: 21.8, 8: 22.0).  NC = 4 consumers (64 output columns each per block of
// 256) for dout <= 256; NC = 8 for wider outputs of one K half; F = 512 keeps NC = 4 with both column blocks of a wave
// walking K together (eight consumers would need 12 waves of <= 168 registers and spill).  HAS_S: the self-term form.
// AGG_ONLY: no product — the consumers store the aggregated rows (a.P) and nothing else: the aggregation itself on
// this kernel's structure (mp_agg_rows_tiles_f32 below).
template <int W, bool WEIGHTED, int U, int KH, int NCB, int PF, bool NT_OUT, bool BF16X3, int TR, int NP, int NC, bool HAS_S,
          bool AGG_ONLY = false, bool MAXR = false>
__global__ __launch_bounds__((NP + NC) * kWave, TR == 64 ? (NP + NC + 3) / 4 : 4)
void agg_dense_pc_kernel(FusedArgs a, unsigned int* __restrict__ tile_ctr, int32_t n_tiles) {
  constexpr int FH = kWave * W;
  constexpr int LDT = FH + 4;
  constexpr int kPcGather = NP;
  constexpr int kTileRows = TR;
  constexpr int kPcThreads = (NP + NC) * kWave;
  constexpr int kPcCons = NC;
  constexpr int RB = TR / 32;       // 32-row MFMA blocks per tile
  static_assert(TR == 32 || TR == 64, "a tile's row starts live in one wave");
  // MAXR: the rows' maximum of w_ij x[j] instead of their sum (AGG_ONLY, no self term; a row without entries is 0 as in
  // spmm.hip's finish_row).  A maximum does not depend on the order of its terms: same bits as the plan-based kernel.
  static_assert(!MAXR || (AGG_ONLY && !HAS_S), "the maximum is an aggregation-only form");
  constexpr float kInit = MAXR ? -INFINITY : 0.f;
  __shared__ __attribute__((aligned(16))) float T[2][kTileRows][LDT];
  __shared__ __attribute__((aligned(16))) float carry[kPcGather - 1][FH];
  __shared__ int carry_row[kPcGather];
  __shared__ float inv_deg[2][kTileRows];
  __shared__ float deg_l[2][kTileRows];       // mean: the row's entry count (the aggregated rows are DIVIDED by it: exact for equal terms)
  __shared__ int defer_l[2][kTileRows];
  __shared__ int next_tile_s, next2_tile_s;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < kPcGather;
  const int cw = wave - kPcGather;            // consumer wave: owns output columns [64 cw, 64 cw + 64) of every block
  const int fr = lane & 31, kk = lane >> 5;
  const int fr_c = fr, kk_c = kk;

  // items are drawn TWO ahead: `cur` is gathered, `nxt` is known (its row starts are requested while `cur` is gathered)
  if (tid == 0) {
    next_tile_s = (int)atomicAdd(tile_ctr, 1u);
    if (KH == 1) next2_tile_s = (int)atomicAdd(tile_ctr, 1u);
  }
  __syncthreads();
  int cur_tile = next_tile_s < n_tiles ? next_tile_s : -1;
  int cur_kh = 0;
  int nxt_tile = KH == 1 ? (next2_tile_s < n_tiles ? next2_tile_s : -1) : cur_tile;
  int nxt_kh = KH == 1 ? 0 : 1;
  int prev_tile = -1, prev_kh = 0;            // the item the consumers work on
  int buf = 0;
  constexpr bool has_s = HAS_S;               // the self term (GIN's (1 + eps) x) initialises the tile
  constexpr int RPW = TR / NP;                // ... each producer wave RPW of its rows, requested one item ahead

  // The two roles run SEPARATE loops over the same item sequence (the workgroup barriers b0 / b1 / b2 pair up by
  // count: s_barrier counts arriving waves, wherever they are in the code — role_barrier(), common.h: the hardware
  // barrier with explicit fences, not __syncthreads(), whose contract is one call site for the whole block), so the
  // registers of the gathers in flight and of the MFMA accumulators are never live in the same code.  Per item both
  // loops execute exactly b1 and b2 (+ b0 once with a self term); tests/test_emitted_barriers.py checks the emitted code.
  if (producer) {
    // producer state: the run of the item being gathered (rp_v: the tile's 33 row starts, lane i = row i; [es, ee): this
    // wave's entries; cvF / wvF: its first 64 indices / values; vb: its first U rows, requested BEFORE the barriers that
    // close the previous item when `pre` is set) and the same for the next item (suffix 1), prepared while this one runs
    typedef const int __attribute__((address_space(4))) cint_t;
    cint_t* __restrict__ rps = (cint_t*)(uintptr_t)a.rowptr;
    int rp_v = INT_MAX, rp_e = 0, es = 0, ee = 0, first_rl = -1, cvF = 0;   // (rp_e: the tile's last row start + its count)
    float wvF = 1.f;
    bool cont = false, pre = false;
    int rp1 = INT_MAX, rp_e1 = 0, es1 = 0, ee1 = 0, first_rl1 = -1, cv1 = 0;
    float wv1 = 1.f;
    bool cont1 = false;
    float vb[U][W];
    // row starts of a tile through the scalar cache (a uniform address; not queued behind the CU's gathers)
    // (lane i = start of tile row i, i < TR; `end` = start of row TR = the end of the tile's entries; rows past the end
    // of the matrix are empty)
    auto row_starts = [&](int tile, int& end) {
      const int NR0 = tile * kTileRows;
      int r = INT_MAX;
      if (NR0 + kTileRows <= a.N) {
#pragma unroll
        for (int i = 0; i < kTileRows; ++i) { const int vi = rps[NR0 + i]; r = lane == i ? vi : r; }
        end = rps[NR0 + kTileRows];
      } else {
        const int NR1 = a.N;
#pragma unroll
        for (int i = 0; i < kTileRows; ++i) { const int vi = rps[min(NR0 + i, NR1)]; r = lane == i ? vi : r; }
        end = rps[NR1];
      }
      return r;
    };
    auto rp_at = [&](int rp, int end, int r) { return r >= kTileRows ? end : bcast_i(rp, r); };   // r: wave-uniform
    // this wave's quarter of a tile's entries: [s, e), the tile row its first entry lies in, and whether that row began
    // in an earlier wave's run
    auto run_of = [&](int rp, int E1, int& s_, int& e_, int& frl, bool& ct) {
      const int E0 = bcast_i(rp, 0);
      const int q = (E1 - E0 + kPcGather - 1) / kPcGather;
      s_ = min(E0 + wave * q, E1);
      e_ = min(s_ + q, E1);
      frl = -1;
      ct = false;
      if (s_ < e_) {
        const unsigned long long started = __ballot(lane >= 1 && lane < kTileRows && rp <= s_);   // (row TR starts at E1 > s_)
        frl = __builtin_amdgcn_readfirstlane((int)__popcll(started));
        ct = bcast_i(rp, frl) < s_;
      }
    };
    if (cur_tile >= 0) {
      rp_v = row_starts(cur_tile, rp_e);
      run_of(rp_v, rp_e, es, ee, first_rl, cont);
      if (es < ee) {
        cvF = a.col[min(es + lane, ee - 1)];
        if (WEIGHTED) wvF = a.val[min(es + lane, ee - 1)];
      }
    }
    // Self term: wave w holds rows RPW w .. RPW w + RPW - 1 of the NEXT item's tile of S in registers from the top of
    // the current item (requested a whole item ahead: no latency left) and writes them, scaled, into the other buffer
    // between the barriers that close the current item — after b1 the consumers are done with that buffer, and b2
    // stands between these stores and the runs that add into them.  (Round 2 / the first form of this kernel loaded
    // the rows at the top of their own item and paid a third barrier per item.)
    float sv[has_s ? RPW : 1][W];
    auto own_rows_load = [&](int tile, int kh) {
      const int NR0 = tile * kTileRows + wave * RPW;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int row = min(NR0 + i, a.N - 1);
        load_vec<W>(a.S + (int64_t)row * a.lds + kh * FH + lane * W, sv[i]);
      }
    };
    auto own_rows_store = [&](int tile, int b) {
      const int NR0 = tile * kTileRows + wave * RPW;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        float v[W];
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = NR0 + i < a.N ? sv[i][k] * a.self_scale : 0.f;
        store_vec<W>(&T[b][wave * RPW + i][lane * W], v);
      }
    };
    if constexpr (has_s) {
      if (cur_tile >= 0) {
        own_rows_load(cur_tile, cur_kh);
        own_rows_store(cur_tile, 0);
        role_barrier();   // b0, once: the first tile initialised (the consumers pass it too)
      }
    }
    int it = 0;
    while (cur_tile >= 0 || prev_tile >= 0) {
      if (wave == 0) PC_T(0);
      const int R0 = cur_tile * kTileRows;
      const int R1 = min(R0 + kTileRows, a.N);
      const int k0 = cur_kh * FH;
      // ================= producers: set up (cur_tile, cur_kh) in buffer `buf` =================
      if (cur_tile >= 0) {
        if (cur_kh == 0) {
          if (wave == 0) {
            int nxt = __shfl_down(rp_v, 1, kWave);
            if (lane == kTileRows - 1) nxt = rp_e;
            if (lane < kTileRows) {
              inv_deg[buf][lane] = (a.mean && nxt > rp_v) ? 1.0f / (float)(nxt - rp_v) : (a.mean ? 0.f : 1.f);
              deg_l[buf][lane] = (a.mean && nxt > rp_v) ? (float)(nxt - rp_v) : 1.f;
              defer_l[buf][lane] = (a.defer_act != nullptr && R0 + lane < R1) ? (int)a.defer_act[R0 + lane] : 0;
            }
          }
        } else if (wave == 0 && lane < kTileRows) {      // second half: the tile's scales move to this buffer
          inv_deg[buf][lane] = inv_deg[buf ^ 1][lane];
          deg_l[buf][lane] = deg_l[buf ^ 1][lane];
          defer_l[buf][lane] = defer_l[buf ^ 1][lane];
        }
        if (lane == 0) carry_row[wave] = cont ? first_rl : -1;
      }
      if (nxt_tile >= 0) {   // the next item's run and its first index batch: on their way while this item runs
        if (nxt_kh == 0) {
          rp1 = row_starts(nxt_tile, rp_e1);
          run_of(rp1, rp_e1, es1, ee1, first_rl1, cont1);
        } else {
          rp1 = rp_v; rp_e1 = rp_e; es1 = es; ee1 = ee; first_rl1 = first_rl; cont1 = cont;
        }
        if (es1 < ee1) {
          cv1 = a.col[min(es1 + lane, ee1 - 1)];
          if (WEIGHTED) wv1 = a.val[min(es1 + lane, ee1 - 1)];
        }
        if constexpr (has_s) own_rows_load(nxt_tile, nxt_kh);   // written into the other buffer once the consumers leave it (b1)
      }

      {
        // ================= producers: phase A of (cur_tile, cur_kh) =================
        if (cur_tile >= 0 && !has_s && wave == 0 && bcast_i(rp_v, 0) == rp_e) {
          float z[W];                                      // a tile without a single entry
#pragma unroll
          for (int k = 0; k < W; ++k) z[k] = 0.f;
          for (int r = 0; r < kTileRows; ++r) store_vec<W>(&T[buf][r][lane * W], z);
        }
        if (cur_tile >= 0 && es < ee) {
          const float* __restrict__ xlane = a.X + k0 + lane * W;
          int rl = first_rl;
          int rend = rp_at(rp_v, rp_e, rl + 1);
          // Row sums in three levels — the burst's 16 terms (`part`), up to 64 bursts (`accr`), the rest (`tot`) — so that
          // a wave's share of a hub row (thousands of entries) is not one sequential fp32 chain: with all-ones input the
          // 10^6-node graph's first row was off by 1.4e-5 of its own value.  A row inside one burst sums as before.
          float accr[W], part[W], tot[W];
          int nfold = 0, cnt = 0;   // (cnt: entries since the last flush — MAXR: a row the run passes over without one is 0)
#pragma unroll
          for (int k = 0; k < W; ++k) { accr[k] = kInit; part[k] = kInit; tot[k] = kInit; }
          // Without a self term nothing initialises the tile: the wave in whose run a row STARTS stores the row (also a
          // row without entries it passes over), later parts of a cut row go to the carries, and the rows no run passes
          // over — empty rows in front of a run's first entry, and behind the tile's last entry — are zeroed by that run.
          auto flush = [&]() {
#pragma unroll
            for (int k = 0; k < W; ++k) {
              if constexpr (MAXR) accr[k] = cnt > 0 ? fmaxf(tot[k], fmaxf(accr[k], part[k])) : 0.f;
              else accr[k] = tot[k] + (accr[k] + part[k]);
              part[k] = kInit; tot[k] = kInit;
            }
            nfold = 0;
            cnt = 0;
            if (cont && rl == first_rl) {
              store_vec<W>(&carry[wave - 1][lane * W], accr);
            } else if (has_s) {
              float t[W];
              load_vec<W>(&T[buf][rl][lane * W], t);
#pragma unroll
              for (int k = 0; k < W; ++k) t[k] += accr[k];
              store_vec<W>(&T[buf][rl][lane * W], t);
            } else {
              store_vec<W>(&T[buf][rl][lane * W], accr);
            }
#pragma unroll
            for (int k = 0; k < W; ++k) accr[k] = kInit;
          };
          if (!has_s) {   // the empty rows between the previous run's last entry (or the tile's head) and this run's first
            int lo = 0;
            if (es > bcast_i(rp_v, 0)) {
              const unsigned long long before = __ballot(lane >= 1 && lane < kTileRows && rp_v <= es - 1);
              lo = __builtin_amdgcn_readfirstlane((int)__popcll(before)) + 1;
            }
            float z[W];
#pragma unroll
            for (int k = 0; k < W; ++k) z[k] = 0.f;
            for (int r = lo; r < first_rl; ++r) store_vec<W>(&T[buf][r][lane * W], z);
          }
          auto issue_burst = [&](int cvb, int jb, const float* __restrict__ xl) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
              // F = 512 (two K halves): the gathered rows are read once and stream past 1.5 MB of split W that every tile
              // re-reads from L2 — non-temporal loads keep them from pushing W out (round 4, in-process A/B on the same
              // buffers: 52.1 -> 51.0 ms; at F = 256, where W is 0.4 MB, no difference: left as plain loads)
              if constexpr (!AGG_ONLY && KH == 2) load_vec_nt<W>(xl + (int64_t)bcast_i(cvb, jb + j) * a.ldx, vb[j]);
              else load_vec<W>(xl + (int64_t)bcast_i(cvb, jb + j) * a.ldx, vb[j]);
            }
          };
          auto consume_burst = [&](int ec, int jb, float wvb) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
              const int e = ec + jb + j;
              if (e < ee) {
                while (e >= rend) {
                  flush();
                  rl += 1;
                  rend = rp_at(rp_v, rp_e, rl + 1);
                }
                const float w = WEIGHTED ? bcast_f(wvb, jb + j) : 1.f;
                cnt += 1;
#pragma unroll
                for (int k = 0; k < W; ++k) part[k] = MAXR ? max_raw(part[k], w * vb[j][k]) : fmaf(w, vb[j][k], part[k]);
              }
            }
            if constexpr (!MAXR) {
#pragma unroll
              for (int k = 0; k < W; ++k) { accr[k] += part[k]; part[k] = 0.f; }
              if (++nfold == 64) {
#pragma unroll
                for (int k = 0; k < W; ++k) { tot[k] += accr[k]; accr[k] = 0.f; }
                nfold = 0;
              }
            }
          };
          int cvb = cvF & 0x7fffffff;   // an identity mark (sign bit) is not part of the index
          float wvb = wvF;
          int jb0 = 0;
          if (pre) {   // the first U rows were requested before the barriers that closed the previous item
            consume_burst(es, 0, wvb);
            jb0 = U;
          }
          for (int ec = es; ec < ee; ec += kWave, jb0 = 0) {
            if (ec != es) {
              const int me = min(ec + lane, ee - 1);
              cvb = a.col[me] & 0x7fffffff;
              if (WEIGHTED) wvb = a.val[me];
            }
            const int n = min(kWave, ee - ec);
            for (int jb = jb0; jb < n; jb += U) {
              issue_burst(cvb, jb, xlane);
              consume_burst(ec, jb, wvb);
            }
          }
          flush();
          if (!has_s && ee == rp_e) {   // this run closes the tile: the empty rows behind it
            float z[W];
#pragma unroll
            for (int k = 0; k < W; ++k) z[k] = 0.f;
            for (int r = rl + 1; r < kTileRows; ++r) store_vec<W>(&T[buf][r][lane * W], z);
          }
        }
        pre = false;
        if (nxt_tile >= 0 && es1 < ee1) {   // the next item's first U rows travel through the barriers below
          const float* __restrict__ xl1 = a.X + nxt_kh * FH + lane * W;
          const int c1 = cv1 & 0x7fffffff;
#pragma unroll
          for (int j = 0; j < U; ++j) {
            if constexpr (!AGG_ONLY && KH == 2) load_vec_nt<W>(xl1 + (int64_t)bcast_i(c1, j) * a.ldx, vb[j]);
            else load_vec<W>(xl1 + (int64_t)bcast_i(c1, j) * a.ldx, vb[j]);
          }
          pre = true;
        }
      }
      if (wave == 0) PC_T(1);
      role_barrier();   // b1: every producer run is reduced into `buf` (and the consumers are done with buf ^ 1)
      if (wave == 0) PC_T(2);

      if constexpr (has_s) {
        if (nxt_tile >= 0) own_rows_store(nxt_tile, buf ^ 1);
      }
      // ---- carries: a row cut by run boundaries gets its later parts in wave order ----
      if (cur_tile >= 0 && tid < FH) {
#pragma unroll
        for (int w = 1; w < kPcGather; ++w) {
          const int cr = carry_row[w];
          if (cr >= 0) T[buf][cr][tid] = MAXR ? fmaxf(T[buf][cr][tid], carry[w - 1][tid]) : T[buf][cr][tid] + carry[w - 1][tid];
        }
      }
      role_barrier();   // b2: buffer `buf` complete; next_tile_s published by the consumers
      if (wave == 0) PC_T(3);
      ++it;

      rp_v = rp1; rp_e = rp_e1; es = es1; ee = ee1; first_rl = first_rl1; cont = cont1; cvF = cv1; wvF = wv1;
      prev_tile = cur_tile; prev_kh = cur_kh;
      cur_tile = nxt_tile; cur_kh = nxt_kh;
      if (nxt_tile >= 0) {
        if (nxt_kh + 1 < KH) {
          nxt_kh += 1;
        } else {
          nxt_kh = 0;
          nxt_tile = next_tile_s < n_tiles ? next_tile_s : -1;
        }
      }
      buf ^= 1;
    }
  } else {
    // consumer accumulators (KH == 2: live across the two halves of a tile)
    f32x16 acc[KH == 2 ? NCB : 1][RB][2];
    int it = 0;   // (items seen; used by the timing build only)

    // one item of consumer work: (prev_tile, K half PKH) from buffer buf ^ 1.  (Walking the two halves of a tile in
    // straight-line code, the half a compile-time constant, was tried: the allocator does worse, 1.3-3.8 KB of scratch.)
    auto work = [&](const int PKH) {
      const int pb = buf ^ 1;
      const int PR0 = prev_tile * kTileRows;
      const int PR1 = min(PR0 + kTileRows, a.N);
      const int pk0 = PKH * FH;
      const int ctid = tid - kPcGather * kWave;
      if (a.P != nullptr) {   // the aggregated rows, kept for the weight gradient
        constexpr int VPR = FH / 4;
        for (int i = ctid; i < kTileRows * VPR; i += kPcCons * kWave) {
          const int m = i / VPR, c = (i % VPR) * 4;
          if (PR0 + m < PR1) {
            f32x4 v = *reinterpret_cast<const f32x4*>(&T[pb][m][c]);
            if (a.mean) v /= deg_l[pb][m];     // (not * 1 / count: a mean of equal values must return the value)
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.P + (int64_t)(PR0 + m) * a.ldp + pk0 + c));
          }
        }
      }
      if constexpr (AGG_ONLY) {
        if (a.Q != nullptr) {   // the second branch: zero rows, except those next to an identity node
          constexpr int VPR = FH / 4;
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          for (int i = ctid; i < kTileRows * VPR; i += kPcCons * kWave) {
            const int m = i / VPR, c = (i % VPR) * 4;
            if (PR0 + m < PR1 && !defer_l[pb][m])
              __builtin_nontemporal_store(z4, reinterpret_cast<f32x4*>(a.Q + (int64_t)(PR0 + m) * a.ldq + pk0 + c));
          }
        }
        return;   // (the aggregated rows above are the output)
      } else {
      auto store_block = [&](const f32x16& acc0, const f32x16& acc1, int n0, int rb) {   // rows [32 rb, 32 rb + 32) of the tile
        int fr = fr_c, kk = kk_c;   // (re-derived behind the accumulators: see the kernel above)
        asm volatile("" : "+v"(fr), "+v"(kk) : "v"(acc0[0]), "v"(acc1[15]));
        const int cpair = n0 + 2 * fr;
        const bool col_ok = cpair < a.dout;
        f32x2 bv = {0.f, 0.f};
        if (a.bias != nullptr && col_ok) bv = *reinterpret_cast<const f32x2*>(a.bias + cpair);
        auto finish = [&](int r, int rl) {
          const float sc = inv_deg[pb][rl];
          f32x2 o = {fmaf(acc0[r], sc, bv[0]), fmaf(acc1[r], sc, bv[1])};
          if (a.R != nullptr && col_ok && PR0 + rl < PR1) {
            const f32x2 rv = *reinterpret_cast<const f32x2*>(a.R + (int64_t)(PR0 + rl) * a.ldr + cpair);
            o[0] += rv[0]; o[1] += rv[1];
          }
          if (a.act == MP_ACT_RELU && !defer_l[pb][rl]) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); }
          return o;
        };
        if (a.out_vec4) {
          const bool odd = fr & 1;
          const int c4 = n0 + 2 * (fr & ~1);
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int rl = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kk;       // rows rl, rl + 1
            const f32x2 m0 = finish(r, rl), m1 = finish(r + 1, rl + 1);
            const f32x2 send = odd ? m0 : m1;
            f32x2 recv;
            recv[0] = __shfl_xor(send[0], 1, kWave);
            recv[1] = __shfl_xor(send[1], 1, kWave);
            const f32x4 o4 = odd ? f32x4{recv[0], recv[1], m1[0], m1[1]} : f32x4{m0[0], m0[1], recv[0], recv[1]};
            const int row = PR0 + rl + (odd ? 1 : 0);
            if (row < PR1) {
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
            const int rl = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int row = PR0 + rl;
            const f32x2 o = finish(r, rl);
            if (col_ok && row < PR1) *reinterpret_cast<f32x2*>(a.out + (int64_t)row * a.ldo + cpair) = o;
          }
        }
      };
      if constexpr (KH == 1) {
        for (int cb = 0; cb < a.dout; cb += 64 * kPcCons) {
          const int n0 = cb + cw * 64;
          if (n0 >= a.dout) break;                       // wave-uniform
          const int cpair = n0 + 2 * fr;
          const int ccol = cpair < a.dout ? cpair : a.dout - 2;
#pragma unroll
          for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][b][0][r] = 0.f; acc[0][b][1][r] = 0.f; }
#ifndef MP_PC_NO_MFMA
          if constexpr (BF16X3) {
            const __bf16* w0 = a.Wsp + ((int64_t)kk * a.dout + ccol) * 8;
            // (64-row tiles, 4 consumers: 169 of 256 registers — room for fragments two K groups ahead: 21.09 -> 20.93 ms,
            // aggregated rows kept 23.03 -> 22.65 ms; three ahead: the same)
            if constexpr (TR == 64 && NC == 4 && NP == 4) mfma_rows_bf16x3_ring<FH, RB, MP_W_RING>(T[pb], w0, (int64_t)a.dout * 16, (int64_t)a.dout * a.ldws, acc, fr, kk);
            else mfma_rows_bf16x3<FH, RB, 1>(T[pb], w0, 0, (int64_t)a.dout * 16, (int64_t)a.dout * a.ldws, acc, fr, kk);
          } else {
            const float* __restrict__ wp = a.Wm + (int64_t)(4 * kk) * a.ldw + ccol;
            mfma_rows<FH, PF, RB>(T[pb], wp, a.ldw, acc[0], fr, kk);
          }
#else
          acc[0][0][0][0] = T[pb][fr][kk + n0];   // (ablation build: no transform)
#endif
#pragma unroll
          for (int b = 0; b < RB; ++b) {
#ifndef MP_PC_NO_STORE
            store_block(acc[0][b][0], acc[0][b][1], n0, b);
#else
            if (acc[0][b][0][0] == 1.2345e-30f) store_block(acc[0][b][0], acc[0][b][1], n0, b);   // (ablation build: no output)
#endif
          }
        }
      } else {
        if (PKH == 0) {
#pragma unroll
          for (int b = 0; b < NCB; ++b)
#pragma unroll
            for (int q = 0; q < RB; ++q)
#pragma unroll
              for (int r = 0; r < 16; ++r) { acc[b][q][0][r] = 0.f; acc[b][q][1][r] = 0.f; }
        }
#ifdef MP_PC_NO_MFMA
        if (false) {
#else
        if constexpr (BF16X3 && NCB > 1) {
#endif
          // the wave's column blocks walk K together (the host sends only dout = 64 NC NCB here: every block of every
          // wave is inside dout; a per-block path beside this one costs the allocator 146 registers of scratch)
          const int ccol = cw * 64 + 2 * fr;
          const __bf16* w0 = a.Wsp + ((int64_t)(pk0 / 8 + kk) * a.dout + ccol) * 8;
          mfma_rows_bf16x3<FH, RB, NCB>(T[pb], w0, (int64_t)64 * kPcCons * 8, (int64_t)a.dout * 16,
                                        (int64_t)a.dout * a.ldws, acc, fr, kk);
#ifdef MP_PC_NO_MFMA
        } else if (a.N < 0) {
#else
        } else {
#endif
#pragma unroll
          for (int b = 0; b < NCB; ++b) {
            const int n0 = b * 64 * kPcCons + cw * 64;
            if (n0 < a.dout) {                             // wave-uniform
              const int cpair = n0 + 2 * fr;
              const int ccol = cpair < a.dout ? cpair : a.dout - 2;
              if constexpr (BF16X3) {
                const __bf16* w0 = a.Wsp + ((int64_t)(pk0 / 8 + kk) * a.dout + ccol) * 8;
                f32x16 one[1][RB][2];   // (a copy in registers: a cast of acc[b] would put all of acc in scratch)
#pragma unroll
                for (int q = 0; q < RB; ++q) { one[0][q][0] = acc[b][q][0]; one[0][q][1] = acc[b][q][1]; }
                mfma_rows_bf16x3<FH, RB, 1>(T[pb], w0, 0, (int64_t)a.dout * 16, (int64_t)a.dout * a.ldws, one, fr, kk);
#pragma unroll
                for (int q = 0; q < RB; ++q) { acc[b][q][0] = one[0][q][0]; acc[b][q][1] = one[0][q][1]; }
              } else {
                const float* __restrict__ wp = a.Wm + (int64_t)(pk0 + 4 * kk) * a.ldw + ccol;
                mfma_rows<FH, PF, RB>(T[pb], wp, a.ldw, acc[b], fr, kk);
              }
            }
          }
        }
        if (cw == 0) PC_T(5);
        if (PKH == KH - 1) {
#pragma unroll
          for (int b = 0; b < NCB; ++b) {
            const int n0 = b * 64 * kPcCons + cw * 64;
            if (n0 < a.dout) {
#pragma unroll
              for (int q = 0; q < RB; ++q) {
#ifdef MP_PC_NO_STORE
                if (acc[b][q][0][0] == 1.2345e-30f)
#endif
                store_block(acc[b][q][0], acc[b][q][1], n0, q);
              }
            }
          }
        }
      }
      }   // !AGG_ONLY
    };
    auto sync_advance = [&]() {
      if (cw == 0) PC_T(6);
      role_barrier();   // b1: the consumers are done with buf ^ 1
      if (cw == 0) PC_T(7);
      // the item after `nxt` opens a new tile when `nxt` is a last half: it is drawn here and published by b2
      const bool draw = nxt_tile >= 0 && nxt_kh == KH - 1;
      if (tid == kPcThreads - 1 && draw) next_tile_s = (int)atomicAdd(tile_ctr, 1u);
      role_barrier();   // b2
      prev_tile = cur_tile; prev_kh = cur_kh;
      cur_tile = nxt_tile; cur_kh = nxt_kh;
      if (nxt_tile >= 0) {
        if (nxt_kh + 1 < KH) {
          nxt_kh += 1;
        } else {
          nxt_kh = 0;
          nxt_tile = next_tile_s < n_tiles ? next_tile_s : -1;
        }
      }
      buf ^= 1;
      ++it;
    };
    if (cur_tile >= 0) {
      if (has_s) role_barrier();   // b0, once (the producers initialise the first tile's buffer)
      sync_advance();
      while (prev_tile >= 0) {      // (the producers run the same number of items)
        if (cw == 0) PC_T(4);
        work(prev_kh);
        sync_advance();
      }
    }
  }
}

constexpr int kPcSlots = 4096;
constexpr int kMaxDevPc = 16;
__device__ unsigned int g_pc_tile_ctr[kPcSlots];

constexpr int kFusedMaxGrid = 65536;

template <int W, int KH, int NCB, int PF>
static int launch_fused_tiles(const FusedArgs& a, hipStream_t st) {   // one workgroup walks tiles b, b + grid, ... (round 2 form)
  const int64_t n_tiles = ceil_div(a.N, kTileRows);
  const dim3 grid((unsigned)(KH == 2 || n_tiles < kFusedMaxGrid ? n_tiles : kFusedMaxGrid)), block(kBlock);
  if (a.Wsp != nullptr) {
    if (a.val) hipLaunchKernelGGL((agg_dense_kernel<W, true, MP_FUSED_U, KH, NCB, PF, true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((agg_dense_kernel<W, false, MP_FUSED_U, KH, NCB, PF, true, true>), grid, block, 0, st, a);
  } else {
    if (a.val) hipLaunchKernelGGL((agg_dense_kernel<W, true, MP_FUSED_U, KH, NCB, PF, true, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((agg_dense_kernel<W, false, MP_FUSED_U, KH, NCB, PF, true, false>), grid, block, 0, st, a);
  }
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// The tile counter of one launch: 4 bytes of a device array, zeroed on the launch's stream right before it.  Launches
// on ONE stream run in order, so a stream owns one slot for its lifetime (keyed by (device, stream)); two streams never
// share a slot, whatever the number of launches in flight (rounds 2-3 handed out 64 slots round-robin: an eager launch
// could memset the counter of a launch on another stream — ADVICE r3).  A launch recorded into a HIP graph bakes its
// slot into the graph, and graphs captured on one stream may later replay concurrently on different streams: every
// captured launch gets a slot of its own that is never handed out again.  4096 slots per device; when they run out
// (thousands of captured launches) the call fails with MP_ERR_UNSUPPORTED rather than share one.
static int pc_counter(unsigned int** ctr, hipStream_t st) {
  static std::mutex mu;
  static unsigned int* base[kMaxDevPc] = {nullptr};
  static int next_slot[kMaxDevPc] = {0};
  static std::map<std::pair<int, hipStream_t>, int> slot_of;
  int dev = 0;
  MP_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevPc) return MP_ERR_UNSUPPORTED;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (st != nullptr) MP_HIP(hipStreamIsCapturing(st, &cap));
  int slot = -1;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!base[dev]) {
      void* p = nullptr;
      MP_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_pc_tile_ctr)));
      base[dev] = reinterpret_cast<unsigned int*>(p);
    }
    if (cap == hipStreamCaptureStatusActive) {
      if (next_slot[dev] < kPcSlots) slot = next_slot[dev]++;
    } else {
      auto key = std::make_pair(dev, st);
      auto it = slot_of.find(key);
      if (it != slot_of.end()) {
        slot = it->second;
      } else if (next_slot[dev] < kPcSlots) {
        slot = next_slot[dev]++;
        slot_of[key] = slot;
      }
    }
  }
  if (slot < 0) return MP_ERR_UNSUPPORTED;
  *ctr = base[dev] + slot;
  MP_HIP(hipMemsetAsync(*ctr, 0, sizeof(unsigned int), st));
  return MP_OK;
}

template <int W, int KH, int NCB, int PF, int TR, int NP, int NC, bool HAS_S>
static int launch_fused_pc_s(const FusedArgs& a, hipStream_t st) {
  const int64_t n_tiles = ceil_div(a.N, TR);
  unsigned int* ctr = nullptr;
  const int rc = pc_counter(&ctr, st);
  if (rc != MP_OK) return rc;
  const int64_t resident = (TR == 64 ? 1 : 2) * kNumCU;   // workgroups the chip holds at once (LDS: 133 KB / 67 KB each)
  const dim3 grid((unsigned)(n_tiles < resident ? n_tiles : resident)), block((NP + NC) * kWave);
  const int32_t nt = (int32_t)n_tiles;
  if (a.Wsp != nullptr) {
    if (a.val) hipLaunchKernelGGL((agg_dense_pc_kernel<W, true, MP_FUSED_U, KH, NCB, PF, true, true, TR, NP, NC, HAS_S>), grid, block, 0, st, a, ctr, nt);
    else hipLaunchKernelGGL((agg_dense_pc_kernel<W, false, MP_FUSED_U, KH, NCB, PF, true, true, TR, NP, NC, HAS_S>), grid, block, 0, st, a, ctr, nt);
  } else {
    if (a.val) hipLaunchKernelGGL((agg_dense_pc_kernel<W, true, MP_FUSED_U, KH, NCB, PF, true, false, TR, NP, NC, HAS_S>), grid, block, 0, st, a, ctr, nt);
    else hipLaunchKernelGGL((agg_dense_pc_kernel<W, false, MP_FUSED_U, KH, NCB, PF, true, false, TR, NP, NC, HAS_S>), grid, block, 0, st, a, ctr, nt);
  }
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// the aggregation alone on the producer/consumer structure: out = reduce_j w_ij x[j] (+ self_scale * S), sum / mean
template <int W, int KH, int TR, int NP, bool MAXR = false>
static int launch_agg_tiles(const FusedArgs& a, hipStream_t st) {
  const int64_t n_tiles = ceil_div(a.N, TR);
  unsigned int* ctr = nullptr;
  const int rc = pc_counter(&ctr, st);
  if (rc != MP_OK) return rc;
  const int64_t resident = (TR == 64 ? 1 : 2) * kNumCU;
  const dim3 grid((unsigned)(n_tiles < resident ? n_tiles : resident)), block((NP + 4) * kWave);
  const int32_t nt = (int32_t)n_tiles;
#define MP_AGG_TILES(WV, SV) \
  hipLaunchKernelGGL((agg_dense_pc_kernel<W, WV, MP_FUSED_U, KH, 1, 1, true, false, TR, NP, 4, SV, true>), grid, block, 0, st, a, ctr, nt)
  if constexpr (MAXR) {
    if (a.val) hipLaunchKernelGGL((agg_dense_pc_kernel<W, true, MP_FUSED_U, KH, 1, 1, true, false, TR, NP, 4, false, true, true>), grid, block, 0, st, a, ctr, nt);
    else hipLaunchKernelGGL((agg_dense_pc_kernel<W, false, MP_FUSED_U, KH, 1, 1, true, false, TR, NP, 4, false, true, true>), grid, block, 0, st, a, ctr, nt);
  } else {
    if (a.val) { if (a.S) MP_AGG_TILES(true, true); else MP_AGG_TILES(true, false); }
    else { if (a.S) MP_AGG_TILES(false, true); else MP_AGG_TILES(false, false); }
  }
#undef MP_AGG_TILES
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// SELF_OK: the self-term form of this shape is built (it holds TR / NP rows of S in registers: not every shape has them)
template <int W, int KH, int NCB, int PF, int TR, int NP, int NC, bool SELF_OK = false>
static int launch_fused_pc(const FusedArgs& a, hipStream_t st) {
  if (a.S == nullptr) return launch_fused_pc_s<W, KH, NCB, PF, TR, NP, NC, false>(a, st);
  if constexpr (SELF_OK) return launch_fused_pc_s<W, KH, NCB, PF, TR, NP, NC, true>(a, st);
  else return launch_fused_tiles<W, KH, NCB, PF>(a, st);
}

static int fused_variant() {   // MP_FUSED_VARIANT (studies: scripts/dbg/fused_variants.py), read per launch
  const char* e = getenv("MP_FUSED_VARIANT");
  return e ? atoi(e) : 0;
}

// Dispatch.  F >= 256: tiles of 64 rows, 4 gathering + 4 multiplying waves, one workgroup per CU.  In-process A/B at
// 10^7 rows, 1.1 x 10^8 entries (profiles/r03_fused_variants.json), out only / aggregated rows kept:
//   F = 256: one-role kernel 23.7 / 24.6 ms, 32 rows x 4 producers 22.5 / 24.3, 64 x 2 24.6 / 25.9, 64 x 4 20.97 / 22.98,
//            64 x 6 21.8 / 23.9, 64 x 8 22.0 / 24.6 (plain aggregation alone: 20.7);   F = 512: 57.4 / 56.3 (32 x 4) / 53.2.
// Narrower layers keep 32-row tiles: their buffers are small enough for two workgroups per CU either way.
template <int W, int KH, int NCB, int PF>
static int launch_fused(const FusedArgs& a, hipStream_t st) {
  const int v = fused_variant();
  if (v == 9) return launch_fused_tiles<W, KH, NCB, PF>(a, st);
  if constexpr (W == 4) {
    if constexpr (KH == 1) {
      if (v == 1) return launch_fused_pc<W, KH, NCB, PF, 32, 4, 4, true>(a, st);
      if (v == 2) return launch_fused_pc<W, KH, NCB, PF, 64, 8, 4>(a, st);
      if (v == 4) return launch_fused_pc<W, KH, NCB, PF, 64, 2, 4>(a, st);
      if (v == 5) return launch_fused_pc<W, KH, NCB, PF, 64, 6, 4>(a, st);
    }
    if constexpr (KH == 1) {
      if (v == 3) return launch_fused_pc<W, KH, 1, PF, 64, 4, 4, true>(a, st);
      if (a.dout > 256 && a.S != nullptr) return launch_fused_tiles<W, KH, NCB, PF>(a, st);   // (no self-term form with 8 consumers)
      // small operators: 32-row tiles (twice the workgroups: 2 x 10^5 rows 0.57 vs 0.71 ms; 10^6 rows 2.13 vs 2.09)
      if (a.N < (1 << 19) && a.dout <= 256) return launch_fused_pc<W, KH, NCB, PF, 32, 4, 4, true>(a, st);
      if (a.dout > 256) return launch_fused_pc<W, KH, 1, PF, 64, 4, 8>(a, st);   // one column block per consumer wave
      return launch_fused_pc<W, KH, 1, PF, 64, 4, 4, true>(a, st);
    } else {
      // F = 512 -> 512: the accumulators live across the K halves.  Eight multiplying waves with one 64-column block each
      // (152 registers; 168 + 20 B of scratch with the self term) against four with two blocks walking K together
      // (rounds 2-3: the one-block form spilled then): 50.8 vs 51.6 ms, with the self term 52.1 vs 52.7 — round 4, same
      // process and buffers, same bits (MP_FUSED_VARIANT=7: the four-wave form)
      if (a.dout == 512 && v != 7) return launch_fused_pc<W, KH, 1, PF, 64, 4, 8, true>(a, st);
      if (a.dout == 512) return launch_fused_pc<W, KH, 2, PF, 64, 4, 4, true>(a, st);
      if (a.dout > 256) return launch_fused_tiles<W, KH, NCB, PF>(a, st);   // (a ragged second block: the one-role kernel)
      return launch_fused_pc<W, KH, 1, PF, 64, 4, 4, true>(a, st);
    }
  } else {
    return launch_fused_pc<W, KH, NCB, PF, 32, 4, 4, true>(a, st);
  }
}

// out[rows[k], :] = act(out[rows[k], :] + sum_{e in [crp[k], crp[k+1])} val[e] * Z[slot[e], :]) — one wave per
// listed row (a few entries each), columns in 16-byte pieces where the alignment allows
// SET: out[rows[k], :] = the sum alone (the row's previous contents are not read)
template <int VW, bool SET = false>
__global__ __launch_bounds__(kBlock) void id_fixup_kernel(const int32_t* __restrict__ rows,
                                                          const int32_t* __restrict__ crp,
                                                          const int32_t* __restrict__ slot,
                                                          const float* __restrict__ val, int32_t n_rows,
                                                          const float* __restrict__ Z, int64_t ldz, float* out,
                                                          int64_t ldo, int32_t d, int32_t act) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int k = blockIdx.x * kWavesPerBlock + wave; k < n_rows; k += gridDim.x * kWavesPerBlock) {
    const int r = rows[k];
    const int e0 = crp[k], e1 = crp[k + 1];
    for (int c0 = lane * VW; c0 < d; c0 += kWave * VW) {
      float acc[VW];
      if constexpr (SET) {
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = 0.f;
      } else {
        load_vec<VW>(out + (int64_t)r * ldo + c0, acc);
      }
      for (int e = e0; e < e1; ++e) {
        float z[VW];
        load_vec<VW>(Z + (int64_t)slot[e] * ldz + c0, z);
        const float w = val ? val[e] : 1.f;
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = fmaf(w, z[i], acc[i]);
      }
      if (act == MP_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = fmaxf(acc[i], 0.f);
      }
      store_vec<VW>(out + (int64_t)r * ldo + c0, acc);
    }
  }
}

static int launch_id_rows(const int32_t* rows, const int32_t* crp, const int32_t* slot, const float* val, int64_t n_rows,
                          const float* Z, int64_t ldz, float* out, int64_t ldo, int32_t d, int max_blocks,
                          hipStream_t st) {
  int blocks = (int)ceil_div(n_rows, kWavesPerBlock);
  if (blocks > max_blocks) blocks = max_blocks;
  const bool v4 = d % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)Z % 16) == 0 && ((uintptr_t)out % 16) == 0;
  if (v4)
    hipLaunchKernelGGL((id_fixup_kernel<4, true>), dim3(blocks), dim3(kBlock), 0, st, rows, crp, slot, val,
                       (int32_t)n_rows, Z, ldz, out, ldo, d, (int32_t)MP_ACT_NONE);
  else
    hipLaunchKernelGGL((id_fixup_kernel<1, true>), dim3(blocks), dim3(kBlock), 0, st, rows, crp, slot, val,
                       (int32_t)n_rows, Z, ldz, out, ldo, d, (int32_t)MP_ACT_NONE);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp

using namespace mp;

extern "C" {

#ifdef MP_FUSED_TIMING
int mp_debug_read(void* dst, size_t bytes) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mp::g_dbg), bytes) == hipSuccess ? 0 : 4;
}
#endif

static int agg_dense_common(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int reduce,
                            const float* X, int64_t ldx, int32_t F, const float* S, int64_t lds, float self_scale,
                            const float* W, int64_t ldw, int32_t d_out, const float* bias, int act,
                            const uint8_t* defer_act, float* P, int64_t ldp, float* out, int64_t ldo,
                            const void* W_split, const float* R, int64_t ldr, mp_stream_t stream) {
  if (!rowptr || !X || !W || !out || N < 0 || F <= 0 || d_out <= 0) return MP_ERR_INVALID_ARG;
  if (W_split && ((uintptr_t)W_split % 16)) return MP_ERR_ALIGNMENT;
  if (ldx < F || ldw < d_out || ldo < d_out || (S && lds < F) || (P && ldp < F) || (R && ldr < d_out))
    return MP_ERR_INVALID_ARG;
  if (act != MP_ACT_NONE && act != MP_ACT_RELU) return MP_ERR_INVALID_ARG;
  if (reduce != MP_SUM && reduce != MP_MEAN) return MP_ERR_INVALID_ARG;
  if (reduce == MP_MEAN && S) return MP_ERR_INVALID_ARG;
  if (F != 64 && F != 128 && F != 256 && F != 512) return MP_ERR_UNSUPPORTED;
  if (d_out % 2) return MP_ERR_UNSUPPORTED;
  if (F == 512 && d_out > 512) return MP_ERR_UNSUPPORTED;   // the accumulators of every column block stay in registers
  if (N >= INT32_MAX - kTileRows) return MP_ERR_UNSUPPORTED;
  const int w = F == 512 ? 4 : F / kWave;
  auto mis = [](const void* p, int64_t ld, int bytes) { return ((uintptr_t)p % bytes) || ((ld * 4) % bytes); };
  if (mis(X, ldx, 4 * w) || (S && mis(S, lds, 16)) || (P && mis(P, ldp, 16)) || mis(W, ldw, 8) || mis(out, ldo, 8) ||
      (bias && ((uintptr_t)bias % 8)) || (R && mis(R, ldr, 8)))
    return MP_ERR_ALIGNMENT;
  if (N == 0) return MP_OK;
  if (!col) return MP_ERR_INVALID_ARG;
  FusedArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.N = (int32_t)N;
  a.X = X; a.ldx = ldx; a.S = S; a.lds = lds; a.self_scale = self_scale;
  a.Wm = W; a.ldw = ldw; a.bias = bias; a.act = act; a.defer_act = defer_act;
  a.Wsp = reinterpret_cast<const __bf16*>(W_split); a.ldws = F;
  a.P = P; a.ldp = ldp; a.out = out; a.ldo = ldo; a.dout = d_out; a.mean = reduce == MP_MEAN;
  a.R = R; a.ldr = ldr;
  a.out_vec4 = !mis(out, ldo, 16);
  hipStream_t st = as_stream(stream);
  if (F == 512) return launch_fused<4, 2, 2, 2>(a, st);   // (the one-block instantiation spills: the compiler's choice)
  switch (w) {
    case 4: return launch_fused<4, 1, 1, 2>(a, st);
    case 2: return launch_fused<2, 1, 1, 1>(a, st);
    default: return launch_fused<1, 1, 1, 1>(a, st);
  }
}

int mp_agg_rows_tiles_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int reduce,
                          const float* X, int64_t ldx, int32_t F, const float* S, int64_t lds, float self_scale,
                          float* out, int64_t ldo, mp_stream_t stream) {
  if (!rowptr || !X || !out || N < 0 || F <= 0) return MP_ERR_INVALID_ARG;
  if (ldx < F || ldo < F || (S && lds < F)) return MP_ERR_INVALID_ARG;
  if (reduce != MP_SUM && reduce != MP_MEAN && reduce != MP_MAX) return MP_ERR_INVALID_ARG;
  if ((reduce == MP_MEAN || reduce == MP_MAX) && S) return MP_ERR_INVALID_ARG;
  if (F != 128 && F != 256 && F != 512) return MP_ERR_UNSUPPORTED;
  if (N >= INT32_MAX - 64) return MP_ERR_UNSUPPORTED;
  auto mis = [](const void* p, int64_t ld, int bytes) { return ((uintptr_t)p % bytes) || ((ld * 4) % bytes); };
  if (mis(X, ldx, 16) || (S && mis(S, lds, 16)) || mis(out, ldo, 16)) return MP_ERR_ALIGNMENT;
  if (N == 0) return MP_OK;
  if (!col) return MP_ERR_INVALID_ARG;
  FusedArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.N = (int32_t)N;
  a.X = X; a.ldx = ldx; a.S = S; a.lds = lds; a.self_scale = self_scale;
  a.Wm = nullptr; a.ldw = 0; a.bias = nullptr; a.act = MP_ACT_NONE; a.defer_act = nullptr;
  a.Wsp = nullptr; a.ldws = F;
  a.P = out; a.ldp = ldo; a.out = out; a.ldo = ldo; a.dout = F; a.mean = reduce == MP_MEAN;
  a.R = nullptr; a.ldr = 0;
  a.out_vec4 = true;
  hipStream_t st = as_stream(stream);
  if (reduce == MP_MAX) {   // (values only: the argmax the backward pass needs stays with mp_spmm_f32)
    if (F == 512) return launch_agg_tiles<4, 2, 64, 4, true>(a, st);
    if (F == 128) return launch_agg_tiles<2, 1, 32, 4, true>(a, st);
    return launch_agg_tiles<4, 1, 64, 4, true>(a, st);
  }
  if (F == 512) return launch_agg_tiles<4, 2, 64, 4>(a, st);
  if (F == 128) return launch_agg_tiles<2, 1, 32, 4>(a, st);   // (512-byte rows: 32-row tiles, two workgroups per CU; 64-row tiles: 13.0 vs 9.85 ms)
  return launch_agg_tiles<4, 1, 64, 4>(a, st);
}

int mp_idgnn_agg_tiles_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, const float* X,
                           int64_t ldx, int32_t F, const uint8_t* id_rows, const int32_t* rows, const int32_t* crp,
                           const int32_t* slot, const float* val_id, int64_t n_rows, const float* Z, int64_t ldz, float* P,
                           int64_t ldp, float* Q, int64_t ldq, mp_stream_t stream) {
  if (!rowptr || !X || !P || !Q || N < 0 || F <= 0 || n_rows < 0) return MP_ERR_INVALID_ARG;
  if (ldx < F || ldp < F || ldq < F) return MP_ERR_INVALID_ARG;
  if (n_rows > 0 && (!id_rows || !rows || !crp || !slot || !Z || ldz < F)) return MP_ERR_INVALID_ARG;
  if (F != 128 && F != 256 && F != 512) return MP_ERR_UNSUPPORTED;
  if (N >= INT32_MAX - 64) return MP_ERR_UNSUPPORTED;
  auto mis = [](const void* p, int64_t ld, int bytes) { return ((uintptr_t)p % bytes) || ((ld * 4) % bytes); };
  if (mis(X, ldx, 16) || mis(P, ldp, 16) || mis(Q, ldq, 16)) return MP_ERR_ALIGNMENT;
  if (N == 0) return MP_OK;
  if (!col) return MP_ERR_INVALID_ARG;
  FusedArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.N = (int32_t)N;
  a.X = X; a.ldx = ldx; a.S = nullptr; a.lds = 0; a.self_scale = 0.f;
  a.Wm = nullptr; a.ldw = 0; a.bias = nullptr; a.act = MP_ACT_NONE; a.defer_act = n_rows > 0 ? id_rows : nullptr;
  a.Wsp = nullptr; a.ldws = F;
  a.P = P; a.ldp = ldp; a.out = P; a.ldo = ldp; a.dout = F; a.mean = 0;
  a.R = nullptr; a.ldr = 0;
  a.out_vec4 = true;
  a.Q = Q; a.ldq = ldq;
  hipStream_t st = as_stream(stream);
  auto tiles = [&]() {
    if (F == 512) return launch_agg_tiles<4, 2, 64, 4>(a, st);
    if (F == 128) return launch_agg_tiles<2, 1, 32, 4>(a, st);
    return launch_agg_tiles<4, 1, 64, 4>(a, st);
  };
  const int rc = tiles();
  if (rc != MP_OK || n_rows == 0) return rc;
  // (run BESIDE the tile kernel on a second stream — the rows are disjoint — it gains nothing: 21.74-21.77 ms against
  // 21.77 in line, same box; the zero rows of Q, 10 GB of stores, are what the second branch costs)
  return launch_id_rows(rows, crp, slot, val_id, n_rows, Z, ldz, Q, ldq, F, kNumCU * 16, st);
}

int mp_agg_dense_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int reduce,
                     const float* X, int64_t ldx, int32_t F, const float* S, int64_t lds, float self_scale, const float* W,
                     int64_t ldw, int32_t d_out, const float* bias, int act, const uint8_t* defer_act, float* P,
                     int64_t ldp, float* out, int64_t ldo, const void* W_split, mp_stream_t stream) {
  return agg_dense_common(rowptr, col, val, N, reduce, X, ldx, F, S, lds, self_scale, W, ldw, d_out, bias, act, defer_act,
                          P, ldp, out, ldo, W_split, nullptr, 0, stream);
}

int mp_agg_dense_add_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t N, int reduce,
                         const float* X, int64_t ldx, int32_t F, const float* S, int64_t lds, float self_scale,
                         const float* W, int64_t ldw, int32_t d_out, const float* bias, int act,
                         const uint8_t* defer_act, float* P, int64_t ldp, float* out, int64_t ldo, const void* W_split,
                         const float* R, int64_t ldr, mp_stream_t stream) {
  if (!R) return MP_ERR_INVALID_ARG;
  return agg_dense_common(rowptr, col, val, N, reduce, X, ldx, F, S, lds, self_scale, W, ldw, d_out, bias, act, defer_act,
                          P, ldp, out, ldo, W_split, R, ldr, stream);
}

int mp_id_fixup_f32(const int32_t* rows, const int32_t* crp, const int32_t* slot, const float* val, int64_t n_rows,
                    const float* Z, int64_t ldz, float* out, int64_t ldo, int32_t d, int act, mp_stream_t stream) {
  if (n_rows < 0 || d <= 0 || (n_rows > 0 && (!rows || !crp || !slot || !Z || !out))) return MP_ERR_INVALID_ARG;
  if (act != MP_ACT_NONE && act != MP_ACT_RELU) return MP_ERR_INVALID_ARG;
  if (ldz < d || ldo < d || n_rows >= INT32_MAX) return MP_ERR_INVALID_ARG;
  if (n_rows == 0) return MP_OK;
  int blocks = (int)ceil_div(n_rows, kWavesPerBlock);
  if (blocks > kNumCU * 16) blocks = kNumCU * 16;
  const bool v4 = d % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)Z % 16) == 0 && ((uintptr_t)out % 16) == 0;
  hipStream_t st = as_stream(stream);
  if (v4)
    hipLaunchKernelGGL(id_fixup_kernel<4>, dim3(blocks), dim3(kBlock), 0, st, rows, crp, slot, val, (int32_t)n_rows, Z,
                       ldz, out, ldo, d, act);
  else
    hipLaunchKernelGGL(id_fixup_kernel<1>, dim3(blocks), dim3(kBlock), 0, st, rows, crp, slot, val, (int32_t)n_rows, Z,
                       ldz, out, ldo, d, act);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_id_rows_f32(const int32_t* rows, const int32_t* crp, const int32_t* slot, const float* val, int64_t n_rows,
                   const float* Z, int64_t ldz, float* out, int64_t ldo, int32_t d, mp_stream_t stream) {
  if (n_rows < 0 || d <= 0 || (n_rows > 0 && (!rows || !crp || !slot || !Z || !out))) return MP_ERR_INVALID_ARG;
  if (ldz < d || ldo < d || n_rows >= INT32_MAX) return MP_ERR_INVALID_ARG;
  if (n_rows == 0) return MP_OK;
  return launch_id_rows(rows, crp, slot, val, n_rows, Z, ldz, out, ldo, d, kNumCU * 16, as_stream(stream));
}

}  // extern "C"

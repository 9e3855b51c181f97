// The dense feature transform that follows an aggregation (K10/K11/K15), fused:
//     out = act( P @ W  [+ Q @ W_id]  + bias )
// i.e. the post-aggregation form of gcn_id (TfgIDLayer.py:510-523) / GCNIDConvLayer (idconv.py:152-184):
// with P = A_hat X and Q = A_hat S X from the two-branch aggregation, this one kernel replaces two
// GEMMs, an add, a bias add and an activation (five passes over [N, d]) of the library path.
//
// fp32 in, fp32 out, the product on the bf16 matrix pipe with three-way split operands (bf16x3.h: fp32-accurate).
// The two products are one GEMM over the concatenated K axis [P | Q] x [W ; W_id].
//   block 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 = 2 x 2 MFMA tiles, K step 16,
//   LDS double-buffered bf16 images (operands split once, by the thread that stages them), global -> register
//   prefetch of tile t+1 issued before the MFMAs of tile t, written to LDS after.  48 KiB LDS, 3 blocks per CU.
// This is the general kernel (any F, d, leading dimension, the dual product).  The hot shape — one product, d = 64 /
// 128 / 256, F % 32 == 0 — has its own streaming kernel in dense_x3.hip.
#include "common.h"
#include "vecio.h"
#include "bf16x3.h"
#include <limits.h>

namespace mp {

constexpr int BM = 128, BK = 16;

// TN = 32-column MFMA tiles per wave along N: block tile 128 x (64 * TN).  TN = 2 (128 columns, 3 waves per
// SIMD) is the default; TN = 4 (256 columns: every A row read once at d <= 256) needs 297 registers, runs
// one wave per SIMD and measured 25 % slower — not built (profiles/r01_dense.log has the measurement).
// VEC = operands allow 16-byte loads (F % 8 == 0, d % 4 == 0, aligned rows); otherwise the loaders fall back to
// guarded scalar loads (any F, d, leading dimension: e.g. Cora's F = 1433) and everything else is unchanged.
// The split happens ONCE per element, in the thread that stages it into LDS (splitting in the MFMA lanes instead repeats
// it per consuming wave and is VALU-bound here: 11.5 ms instead of 13.6 for the weight gradient, not the 2.7x the MFMA
// count promises); bf16x3.h has the split and the six-term product.
// A [16 k-rows][128 columns] bf16 tile with 256-byte rows whose 16-byte chunks are XOR-swizzled so that both the
// row-wise 16-byte stores of the loaders and the transposed reads below are free of bank conflicts
// (cdna_hip_programming.md T10, image (b)).  One plane per split: 4 KiB.
constexpr int kPlaneBytes = 16 * 256;
__device__ __forceinline__ int swz_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
// the MFMA operand of a k-strided tile: lane l of the wave gets, for column col0 + (l & 31), the 8 k-rows 8 (l >> 5) ..
// + 7 — two ds_read_b64_tr_b16 (hardware transpose read: per 16-lane group a 4-row x 16-column block, lane 4q + p
// supplying the address of row q, columns 4p .. 4p + 3, lane i receiving column i).  EXEC must be all ones.
__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* plane, int col0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int chunk = ((col0 + 16 * (g & 1)) >> 3) + (p >> 1);
  const int r0 = 8 * (g >> 1) + q;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4*)(plane + swz_off(r0, chunk) + 8 * (p & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4*)(plane + swz_off(r0 + 4, chunk) + 8 * (p & 1)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
template <bool DUAL, int TN, bool VEC>
__global__ __launch_bounds__(kBlock, TN == 2 ? 3 : 1) void dense_fused_kernel(const float* __restrict__ P, int64_t ldp,
                                                             const float* __restrict__ W,
                                                             const float* __restrict__ Q, int64_t ldq,
                                                             const float* __restrict__ Wid,
                                                             const float* __restrict__ bias, int act,
                                                             float* __restrict__ out, int64_t ldo, int64_t M,
                                                             int32_t F, int32_t d) {
  constexpr int BN = 64 * TN;
  constexpr int NB4 = BN / 64;           // float4 loads of B per thread per tile
  static_assert(TN == 2, "the bf16 LDS images below are laid out for a 128-column block tile");
  // bf16x3 images (three split planes each): A [128 rows][16 k] k-contiguous (a lane's operand is one 16-byte read);
  // B [16 k][128 columns] row-major, chunk-swizzled, consumed through transposed reads
  __shared__ __attribute__((aligned(16))) unsigned char Aimg[2][3][BM * 32];
  __shared__ __attribute__((aligned(16))) unsigned char Bimg[2][3][kPlaneBytes];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // column blocks of one row block are neighbours in dispatch order: the second reader of an A tile
  // finds it in the Infinity Cache instead of HBM
  const int ncb = (d + BN - 1) / BN;   // (BN declared below is a compile-time constant of this instantiation)
  const int64_t m0 = (int64_t)(blockIdx.x / ncb) * BM;
  const int n0 = (int)(blockIdx.x % ncb) * BN;

  // loader roles
  const int a_row = tid >> 1;            // 0..127
  const int a_k = (tid & 1) * 8;         // 0 or 8
  const int b_k = tid >> 4;              // 0..15
  const int b_n = (tid & 15) * (4 * NB4);  // first of this thread's 4*NB4 consecutive columns
  const int64_t g_row = m0 + a_row;
  const bool row_ok = g_row < M;
  const int KT = DUAL ? 2 * F : F;

  f32x4 ra[2], rb[NB4];
  auto fetch = [&](int kt) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const int k = kt + a_k;
    const int kb = kt + b_k;
    if constexpr (VEC) {
      // A: 8 consecutive k of one row, from P or (past F) from Q; F % 8 == 0 keeps a fetch inside one operand
      const bool second = DUAL && k >= F;
      const float* src = second ? Q + g_row * ldq + (k - F) : P + g_row * ldp + k;
      const bool ok = row_ok && k < KT;
      ra[0] = ok ? *reinterpret_cast<const f32x4*>(src) : z;
      ra[1] = ok ? *reinterpret_cast<const f32x4*>(src + 4) : z;
      // B: consecutive n of one k row of W or W_id
      const bool second_b = DUAL && kb >= F;
      const float* wsrc = (second_b ? Wid + (int64_t)(kb - F) * d : W + (int64_t)kb * d) + n0 + b_n;
#pragma unroll
      for (int q = 0; q < NB4; ++q)
        rb[q] = (kb < KT && n0 + b_n + 4 * q < d) ? *reinterpret_cast<const f32x4*>(wsrc + 4 * q) : z;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ki = k + i;
        float v = 0.f;
        if (row_ok && ki < KT) v = (DUAL && ki >= F) ? Q[g_row * ldq + (ki - F)] : P[g_row * ldp + ki];
        ra[i >> 2][i & 3] = v;
      }
      const bool second_b = DUAL && kb >= F;
      const float* wrow = second_b ? Wid + (int64_t)(kb - F) * d : W + (int64_t)kb * d;
#pragma unroll
      for (int q = 0; q < NB4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int nn = n0 + b_n + 4 * q + i;
          rb[q][i] = (kb < KT && nn < d) ? wrow[nn] : 0.f;
        }
    }
  };
  auto stash = [&](int buf) {
    const float av[8] = {ra[0][0], ra[0][1], ra[0][2], ra[0][3], ra[1][0], ra[1][1], ra[1][2], ra[1][3]};
    bf16x8 sa[3];
    split3_bf16(av, sa[0], sa[1], sa[2]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<bf16x8*>(&Aimg[buf][pl][a_row * 32 + (a_k >> 3) * 16]) = sa[pl];
    const float bv[8] = {rb[0][0], rb[0][1], rb[0][2], rb[0][3], rb[1][0], rb[1][1], rb[1][2], rb[1][3]};
    bf16x8 sb[3];
    split3_bf16(bv, sb[0], sb[1], sb[2]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<bf16x8*>(&Bimg[buf][pl][swz_off(b_k, b_n >> 3)]) = sb[pl];
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch(0);
  stash(0);
  __syncthreads();
  const int ntiles = (KT + BK - 1) / BK;
  const int fr = lane & 31, fk = lane >> 5;
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) fetch((t + 1) * BK);          // in flight while the MFMAs below run
    {
      // the K = 16 tile as one step on the bf16 matrix pipe: operands were split when they were staged
      bf16x8 as[2][3], bs3[TN][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          as[i][pl] = *reinterpret_cast<const bf16x8*>(&Aimg[buf][pl][(wm * 64 + i * 32 + fr) * 32 + fk * 16]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bs3[j][pl] = tr_read8(Bimg[buf][pl], wn * (32 * TN) + j * 32, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mfma6(acc[i][j], as[i], bs3[j]);
    }
    if (t + 1 < ntiles) stash(buf ^ 1);               // the other buffer: nobody reads it this iteration
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 f32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * (32 * TN) + j * 32 + fr;
    const bool col_ok = col < d;
    const float bv = (bias != nullptr && col_ok) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        float v = acc[i][j][r] + bv;
        if (act == MP_ACT_RELU) v = fmaxf(v, 0.f);
        if (col_ok && row < M) out[row * ldo + col] = v;
      }
    }
  }
}


// Weight gradient of the transform: dW[F, d] = P^T g = sum_m P[m, :]^T g[m, :]  (the reduction runs over
// the node axis).  Both MFMA operands are plain row tiles with k = node index — A[i = f][k = m] = P[m][f],
// B[k = m][j] = g[m][j] — so the LDS tiles are straight copies of global rows and every fragment read is
// 32 consecutive words.  Split-K over nodes: a block owns `chunk` rows and one 128 x 128 output tile and
// writes its partial to a slab; a second kernel adds the slabs in chunk order (bitwise reproducible, unlike
// float atomics).
// RELU: the backward of a ReLU epilogue rides along — g is masked by [Y > 0] (Y = the forward output) on its way to
// LDS, and the blocks of the first f-tile also write the masked gradient GM (what the input-gradient launch reads):
// the separate masking pass (30 GB at 10^7 x 256) disappears into a pass that is bound by its MFMA work.
template <bool VEC, bool RELU>
__global__ __launch_bounds__(kBlock, 3) void dense_wgrad_kernel(const float* __restrict__ P, int64_t ldp,
                                                                const float* G, int64_t ldg,
                                                                const float* __restrict__ Y, int64_t ldy,
                                                                float* GM, int64_t ldgm,
                                                                int64_t M, int32_t F, int32_t d, int64_t chunk,
                                                                float* __restrict__ slabs,
                                                                float* __restrict__ bias_slabs) {
  constexpr int BT = 128;                 // output tile 128 (f) x 128 (d)
  // bf16x3 images: [16 node rows][128 columns] per split plane, chunk-swizzled; both MFMA operands are k-strided
  // (k = node row) and come out of them through transposed reads
  __shared__ __attribute__((aligned(16))) unsigned char Pimg[2][3][kPlaneBytes];
  __shared__ __attribute__((aligned(16))) unsigned char Gimg[2][3][kPlaneBytes];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tf = (F + BT - 1) / BT, td = (d + BT - 1) / BT;
  const int tile = blockIdx.x % (tf * td);
  const int64_t c = blockIdx.x / (tf * td);
  const int f0 = (tile / td) * BT, d0 = (tile % td) * BT;
  const int64_t mb = c * chunk;
  const int64_t me = mb + chunk < M ? mb + chunk : M;

  const int l_row = tid >> 4;              // 0..15: node row inside the tile
  const int l_col = (tid & 15) * 8;        // 8 consecutive columns
  f32x4 rp[2], rg[2];
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int64_t m0) {
    const int64_t m = m0 + l_row;
    const bool ok = m < me;
    const float* ps = P + m * ldp + f0 + l_col;
    const float* gs = G + m * ldg + d0 + l_col;
    if constexpr (VEC) {
      rp[0] = (ok && f0 + l_col < F) ? *reinterpret_cast<const f32x4*>(ps) : z;
      rp[1] = (ok && f0 + l_col + 4 < F) ? *reinterpret_cast<const f32x4*>(ps + 4) : z;
      rg[0] = (ok && d0 + l_col < d) ? *reinterpret_cast<const f32x4*>(gs) : z;
      rg[1] = (ok && d0 + l_col + 4 < d) ? *reinterpret_cast<const f32x4*>(gs + 4) : z;
      if constexpr (RELU) {
        const float* ys = Y + m * ldy + d0 + l_col;
        const f32x4 y0 = (ok && d0 + l_col < d) ? *reinterpret_cast<const f32x4*>(ys) : z;
        const f32x4 y1 = (ok && d0 + l_col + 4 < d) ? *reinterpret_cast<const f32x4*>(ys + 4) : z;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rg[0][i] = y0[i] > 0.f ? rg[0][i] : 0.f;
          rg[1][i] = y1[i] > 0.f ? rg[1][i] : 0.f;
        }
        if (GM != nullptr && f0 == 0 && ok) {
          float* go = GM + m * ldgm + d0 + l_col;
          if (d0 + l_col < d) *reinterpret_cast<f32x4*>(go) = rg[0];
          if (d0 + l_col + 4 < d) *reinterpret_cast<f32x4*>(go + 4) = rg[1];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        rp[i >> 2][i & 3] = (ok && f0 + l_col + i < F) ? ps[i] : 0.f;
        float gv = (ok && d0 + l_col + i < d) ? gs[i] : 0.f;
        if constexpr (RELU) {
          const bool in = ok && d0 + l_col + i < d;
          const float yv = in ? Y[m * ldy + d0 + l_col + i] : 0.f;
          gv = yv > 0.f ? gv : 0.f;
          if (GM != nullptr && f0 == 0 && in) GM[m * ldgm + d0 + l_col + i] = gv;
        }
        rg[i >> 2][i & 3] = gv;
      }
    }
  };
  auto stash = [&](int buf) {
    const float pv[8] = {rp[0][0], rp[0][1], rp[0][2], rp[0][3], rp[1][0], rp[1][1], rp[1][2], rp[1][3]};
    const float gv[8] = {rg[0][0], rg[0][1], rg[0][2], rg[0][3], rg[1][0], rg[1][1], rg[1][2], rg[1][3]};
    bf16x8 sp[3], sg[3];
    split3_bf16(pv, sp[0], sp[1], sp[2]);
    split3_bf16(gv, sg[0], sg[1], sg[2]);
    const int off = swz_off(l_row, l_col >> 3);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      *reinterpret_cast<bf16x8*>(&Pimg[buf][pl][off]) = sp[pl];
      *reinterpret_cast<bf16x8*>(&Gimg[buf][pl][off]) = sg[pl];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // the bias gradient (column sums of g) rides along: the blocks of the first f-tile see every element of their
  // g columns once, in registers, on its way to LDS
  const bool do_bias = bias_slabs != nullptr && f0 == 0;
  float bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bs[i] = 0.f;
  auto tally = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) { bs[i] += rg[0][i]; bs[4 + i] += rg[1][i]; }
  };

  const int fr = lane & 31, fk = lane >> 5;
  const int64_t ntiles = (me - mb + BK - 1) / BK;
  if (ntiles > 0) {
    fetch(mb);
    if (do_bias) tally();
    stash(0);
  }
  __syncthreads();
  for (int64_t t = 0; t < ntiles; ++t) {
    const int buf = (int)(t & 1);
    if (t + 1 < ntiles) {
      fetch(mb + (t + 1) * BK);
      if (do_bias) tally();
    }
    {
      // the 16 node rows of the tile as ONE K = 16 step on the bf16 matrix pipe (operands split when staged)
      bf16x8 as[2][3], bs3[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          as[i][pl] = tr_read8(Pimg[buf][pl], wm * 64 + i * 32, lane);
          bs3[i][pl] = tr_read8(Gimg[buf][pl], wn * 64 + i * 32, lane);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma6(acc[i][j], as[i], bs3[j]);
    }
    if (t + 1 < ntiles) stash(buf ^ 1);
    __syncthreads();
  }
  float* slab = slabs + c * (int64_t)F * d;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = d0 + wn * 64 + j * 32 + fr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = f0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (row < F && col < d) slab[(int64_t)row * d + col] = acc[i][j][r];
      }
    }
  }
  if (do_bias) {   // 16 threads hold partial sums of the same 8 columns: add them in row order through LDS
    float (*red)[BT] = reinterpret_cast<float (*)[BT]>(&Pimg[0][0][0]);      // [16][128] floats = 8 KiB of the 24 KiB image
    *reinterpret_cast<f32x4*>(&red[l_row][l_col]) = f32x4{bs[0], bs[1], bs[2], bs[3]};
    *reinterpret_cast<f32x4*>(&red[l_row][l_col + 4]) = f32x4{bs[4], bs[5], bs[6], bs[7]};
    __syncthreads();
    if (tid < BT && d0 + tid < d) {
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < BK; ++r) sacc += red[r][tid];
      bias_slabs[c * (int64_t)d + d0 + tid] = sacc;
    }
  }
}

// The same product with a 256 (f) x 128 (d) output tile per block (F > 128): wave w owns f rows 64 w .. 64 w + 63 and
// all four 32-column d tiles (128 accumulator registers), so a staged element of g feeds eight MFMA tiles instead of
// four and g, y and the masked-gradient output pass through a block ONCE per d tile instead of once per (f, d) tile
// pair.  The kernel is bound by the issue of its vector memory instructions (section on dense_x3.hip in DESIGN.md):
// per MFMA this form issues 40 % fewer with the ReLU mask, 25 % fewer without.  72 KiB LDS: two blocks per CU.
// ABL: ablation bits for scripts/dbg/wgrad_ablate.hip only (1 no global loads after the first tile, 2 no split / LDS
// stores, 4 no MFMAs, 8 no LDS fragment reads — timing experiments with wrong results); the library builds ABL = 0.
template <bool RELU, int ABL = 0>
__global__ __launch_bounds__(kBlock, 2) void dense_wgrad_wide_kernel(const float* __restrict__ P, int64_t ldp,
                                                                     const float* G, int64_t ldg,
                                                                     const float* __restrict__ Y, int64_t ldy,
                                                                     float* GM, int64_t ldgm,
                                                                     int64_t M, int32_t F, int32_t d, int64_t chunk,
                                                                     float* __restrict__ slabs,
                                                                     float* __restrict__ bias_slabs) {
  constexpr int BF = 256, BD = 128;
  __shared__ __attribute__((aligned(16))) unsigned char Pimg[2][3][2][kPlaneBytes];   // two 128-column halves
  __shared__ __attribute__((aligned(16))) unsigned char Gimg[2][3][kPlaneBytes];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tf = (F + BF - 1) / BF, td = (d + BD - 1) / BD;
  const int tile = blockIdx.x % (tf * td);
  const int64_t c = blockIdx.x / (tf * td);
  const int f0 = (tile / td) * BF, d0 = (tile % td) * BD;
  const int64_t mb = c * chunk;
  const int64_t me = mb + chunk < M ? mb + chunk : M;

  const int l_row = tid >> 4;              // 0..15: node row inside the tile
  const int l_col = (tid & 15) * 8;        // 8 consecutive columns (of each 128-column half for P)
  f32x4 rp[2][2], rg[2];
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int64_t m0) {
    const int64_t m = m0 + l_row;
    const bool ok = m < me;
    const float* gs = G + m * ldg + d0 + l_col;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int fc = f0 + 128 * h + l_col;
      const float* ps = P + m * ldp + fc;
      rp[h][0] = (ok && fc < F) ? *reinterpret_cast<const f32x4*>(ps) : z;
      rp[h][1] = (ok && fc + 4 < F) ? *reinterpret_cast<const f32x4*>(ps + 4) : z;
    }
    rg[0] = (ok && d0 + l_col < d) ? *reinterpret_cast<const f32x4*>(gs) : z;
    rg[1] = (ok && d0 + l_col + 4 < d) ? *reinterpret_cast<const f32x4*>(gs + 4) : z;
    if constexpr (RELU) {
      const float* ys = Y + m * ldy + d0 + l_col;
      const f32x4 y0 = (ok && d0 + l_col < d) ? *reinterpret_cast<const f32x4*>(ys) : z;
      const f32x4 y1 = (ok && d0 + l_col + 4 < d) ? *reinterpret_cast<const f32x4*>(ys + 4) : z;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rg[0][i] = y0[i] > 0.f ? rg[0][i] : 0.f;
        rg[1][i] = y1[i] > 0.f ? rg[1][i] : 0.f;
      }
      if (GM != nullptr && f0 == 0 && ok) {
        float* go = GM + m * ldgm + d0 + l_col;
        if (d0 + l_col < d) *reinterpret_cast<f32x4*>(go) = rg[0];
        if (d0 + l_col + 4 < d) *reinterpret_cast<f32x4*>(go + 4) = rg[1];
      }
    }
  };
  auto stash = [&](int buf) {
    const int off = swz_off(l_row, l_col >> 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float pv[8] = {rp[h][0][0], rp[h][0][1], rp[h][0][2], rp[h][0][3], rp[h][1][0], rp[h][1][1], rp[h][1][2],
                           rp[h][1][3]};
      bf16x8 sp[3];
      split3_bf16(pv, sp[0], sp[1], sp[2]);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(&Pimg[buf][pl][h][off]) = sp[pl];
    }
    const float gv[8] = {rg[0][0], rg[0][1], rg[0][2], rg[0][3], rg[1][0], rg[1][1], rg[1][2], rg[1][3]};
    bf16x8 sg[3];
    split3_bf16(gv, sg[0], sg[1], sg[2]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(&Gimg[buf][pl][off]) = sg[pl];
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool do_bias = bias_slabs != nullptr && f0 == 0;
  float bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bs[i] = 0.f;
  auto tally = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) { bs[i] += rg[0][i]; bs[4 + i] += rg[1][i]; }
  };

  const int fr = lane & 31, fk = lane >> 5;
  const int64_t ntiles = (me - mb + BK - 1) / BK;
  if (ntiles > 0) {
    fetch(mb);
    if (do_bias) tally();
    stash(0);
  }
  __syncthreads();
  for (int64_t t = 0; t < ntiles; ++t) {
    const int buf = (int)(t & 1);
    if (t + 1 < ntiles && !(ABL & 1)) {
      fetch(mb + (t + 1) * BK);
      if (do_bias) tally();
    }
    {
      bf16x8 as[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          if constexpr (ABL & 8) as[i][pl] = __builtin_bit_cast(bf16x8, rp[i][0] + rg[0]);
          else as[i][pl] = tr_read8(Pimg[buf][pl][wave >> 1], (wave & 1) * 64 + i * 32, lane);
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8 b3[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          if constexpr (ABL & 8) b3[pl] = __builtin_bit_cast(bf16x8, rg[j & 1] + rp[0][1]);
          else b3[pl] = tr_read8(Gimg[buf][pl], j * 32, lane);
        }
        if constexpr (ABL & 4) {
          asm volatile("" ::"v"(b3[0]), "v"(b3[1]), "v"(b3[2]), "v"(as[0][0]), "v"(as[1][2]));
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) mfma6(acc[i][j], as[i], b3);
        }
      }
    }
    if (t + 1 < ntiles && !(ABL & 2)) stash(buf ^ 1);
    __syncthreads();
  }
  float* slab = slabs + c * (int64_t)F * d;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = d0 + j * 32 + fr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = f0 + wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (row < F && col < d) slab[(int64_t)row * d + col] = acc[i][j][r];
      }
    }
  }
  if (do_bias) {   // 16 threads hold partial sums of the same 8 columns: add them in row order through LDS
    float (*red)[BD] = reinterpret_cast<float (*)[BD]>(&Pimg[0][0][0][0]);
    *reinterpret_cast<f32x4*>(&red[l_row][l_col]) = f32x4{bs[0], bs[1], bs[2], bs[3]};
    *reinterpret_cast<f32x4*>(&red[l_row][l_col + 4]) = f32x4{bs[4], bs[5], bs[6], bs[7]};
    __syncthreads();
    if (tid < BD && d0 + tid < d) {
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < BK; ++r) sacc += red[r][tid];
      bias_slabs[c * (int64_t)d + d0 + tid] = sacc;
    }
  }
}

// The weight gradient at the hot shape (128 < F, d <= 256) with loading and multiplying on DIFFERENT waves (round 3).
// The kernels above stage, split and multiply in the same waves: their MFMAs alone take 3.9 ms at 10^7 x 256 x 256 and
// their loads + split + LDS stores alone 4.7 ms, but a wave that waits at a vector memory instruction issues no MFMAs
// and the two add up to 10.5 ms.  Here a workgroup is 12 waves: waves 0-7 own the WHOLE 256 x 256
// gradient (wave w: f rows 64 (w >> 1) .., d columns 128 (w & 1) .., 128 accumulator registers) and do nothing but
// read fragments and issue MFMAs; waves 8-11 load the next 16-node tile of P and g (and y for the ReLU mask), split
// it three ways and store the bf16 images — 48 KiB per stage, two stages — one workgroup barrier per tile.  P, g and y
// are read exactly ONCE (the 256 x 128 tile read P once per d tile: FETCH 30.7 GB for 20.5 GB of operands).  One
// workgroup per CU and one contiguous range of nodes per workgroup: at most kNumCU slabs, all of the same length.
#ifndef MP_WPC_ABL
#define MP_WPC_ABL 0   // ablation bits (timing studies, wrong results): 1 loaders idle, 2 no MFMAs, 16 with 1: loads only
#endif
constexpr int WPC_MFMA_WAVES = 8, WPC_LOAD_WAVES = 4;
constexpr int WPC_THREADS = 64 * (WPC_MFMA_WAVES + WPC_LOAD_WAVES);

// DT: 128-column halves of d (2: d <= 256, wave w owns f rows 64 (w >> 1) .. and d half w & 1; 1: d <= 128, wave w owns f
// rows 32 w .. and all of d — 64 accumulator registers)
template <bool RELU, int DT>
__global__ __launch_bounds__(WPC_THREADS, 3) void dense_wgrad_pc_kernel(const float* __restrict__ P, int64_t ldp,
                                                                        const float* G, int64_t ldg,
                                                                        const float* __restrict__ Y, int64_t ldy,
                                                                        float* GM, int64_t ldgm, int64_t M, int32_t F,
                                                                        int32_t d, int64_t chunk,
                                                                        float* __restrict__ slabs,
                                                                        float* __restrict__ bias_slabs) {
  __shared__ __attribute__((aligned(16))) unsigned char Pimg[2][3][2][kPlaneBytes];   // [stage][plane][128-column half]
  __shared__ __attribute__((aligned(16))) unsigned char Gimg[2][3][DT][kPlaneBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wider gradients: output tiles of 256 (f) x 128 DT (d), tile index fastest (P is then read once per d tile, g once
  // per f tile: F = d = 512 moves each operand twice)
  const int tf = (F + 255) / 256, td = (d + 128 * DT - 1) / (128 * DT);
  const int tile = (int)(blockIdx.x % (unsigned)(tf * td));
  const int64_t c = blockIdx.x / (unsigned)(tf * td);
  const int f0 = (tile / td) * 256, d0 = (tile % td) * 128 * DT;
  const int64_t mb = c * chunk;
  const int64_t me = mb + chunk < M ? mb + chunk : M;
  const int64_t ntiles = (me - mb + BK - 1) / BK;      // >= 1: the host launches ceil(M / chunk) workgroups

  // Both roles run `nsteps` steps, a multiple of NSET: the tiles past the end of the range are all zero (rows >= me are
  // zeroed at the split), so the loader loop is branch-free and unrolled over its register sets.
  constexpr int NSET = RELU ? 2 : 3;         // tiles in flight per loader thread (16 / 24 registers each)
  const int64_t nsteps = (ntiles + NSET - 1) / NSET * NSET;
  if (wave >= WPC_MFMA_WAVES) {
    // ------------------------------------------------ loaders ------------------------------------------------
    const int lt = tid - 64 * WPC_MFMA_WAVES;
    const int l_row = lt >> 4;               // 0..15: node row inside the tile
    const int l_col = (lt & 15) * 8;         // 8 consecutive columns of each 128-column half
    const bool do_bias = bias_slabs != nullptr && f0 == 0;
    float bs[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 8; ++i) bs[h][i] = 0.f;
    // (F % 8 == 0 and d % 8 == 0 — the host checks — so a thread's 8 columns are inside or outside as a whole; loads
    // outside read column 0 / the last row of the range and are zeroed at the split: no branch around a load)
    bool pin[2], gin[2];
    int pcol[2], gcol[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int fc = 128 * h + l_col;
      pin[h] = f0 + fc < F; gin[h] = d0 + fc < d;
      pcol[h] = pin[h] ? f0 + fc : 0; gcol[h] = gin[h] ? d0 + fc : 0;
    }
    f32x4 rp[NSET][2][2], rg[NSET][2][2], ry[RELU ? NSET : 1][2][2];
    auto fetch = [&](int s, int64_t tile) {            // s: compile-time after unrolling
      int64_t m = mb + tile * BK + l_row;
      m = m < me ? m : me - 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float* ps = P + m * ldp + pcol[h];
        rp[s][h][0] = *reinterpret_cast<const f32x4*>(ps);
        rp[s][h][1] = *reinterpret_cast<const f32x4*>(ps + 4);
        if (h < DT) {                                  // (compile-time after unrolling)
          const float* gs = G + m * ldg + gcol[h];
          rg[s][h][0] = *reinterpret_cast<const f32x4*>(gs);
          rg[s][h][1] = *reinterpret_cast<const f32x4*>(gs + 4);
          if constexpr (RELU) {
            const float* ys = Y + m * ldy + gcol[h];
            ry[s][h][0] = *reinterpret_cast<const f32x4*>(ys);
            ry[s][h][1] = *reinterpret_cast<const f32x4*>(ys + 4);
          }
        }
      }
    };
    auto stash = [&](int s, int64_t tile, int buf) {
      const int off = swz_off(l_row, l_col >> 3);
      const int64_t m = mb + tile * BK + l_row;
      const bool ok = m < me;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) pv[i] = (ok && pin[h]) ? rp[s][h][i >> 2][i & 3] : 0.f;
        bf16x8 sp[3];
        split3_bf16(pv, sp[0], sp[1], sp[2]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(&Pimg[buf][pl][h][off]) = sp[pl];
        if (h < DT) {
          float gv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            gv[i] = (ok && gin[h]) ? rg[s][h][i >> 2][i & 3] : 0.f;
            if constexpr (RELU) gv[i] = ry[s][h][i >> 2][i & 3] > 0.f ? gv[i] : 0.f;
          }
          if constexpr (RELU) {
            if (GM != nullptr && f0 == 0 && ok && gin[h]) {
              float* go = GM + m * ldgm + gcol[h];
              *reinterpret_cast<f32x4*>(go) = f32x4{gv[0], gv[1], gv[2], gv[3]};
              *reinterpret_cast<f32x4*>(go + 4) = f32x4{gv[4], gv[5], gv[6], gv[7]};
            }
          }
          bf16x8 sg[3];
          split3_bf16(gv, sg[0], sg[1], sg[2]);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(&Gimg[buf][pl][h][off]) = sg[pl];
          if (do_bias) {
#pragma unroll
            for (int i = 0; i < 8; ++i) bs[h][i] += gv[i];
          }
        }
      }
    };
    // tile k lives in register set k % NSET from NSET steps before its step; during step t (the MFMA waves read stage
    // t & 1) tile t + 1 is split into stage (t + 1) & 1 and tile t + 1 + NSET is requested into the set it leaves
#pragma unroll
    for (int k = 0; k < NSET; ++k) fetch(k, k);
    stash(0, 0, 0);
    fetch(0, NSET);
    role_barrier();                                    // stage 0 holds tile 0
    for (int64_t base = 0; base < nsteps; base += NSET) {
#pragma unroll
      for (int u = 0; u < NSET; ++u) {
        const int64_t t = base + u;
        const int sidx = (u + 1) % NSET;
#if !(MP_WPC_ABL & 1)
        stash(sidx, t + 1, (int)((t + 1) & 1));
        fetch(sidx, t + 1 + NSET);
#elif (MP_WPC_ABL & 16)
        fetch(sidx, t + 1 + NSET);                       // (loads only: no split, no LDS stores)
#endif
        role_barrier();                                // step t done
      }
    }
    if (do_bias) {   // 16 threads hold partial sums of the same 8 columns: add them in row order through LDS
      float (*red)[256] = reinterpret_cast<float (*)[256]>(&Pimg[0][0][0][0]);      // [16][256] floats = 16 KiB
#pragma unroll
      for (int h = 0; h < DT; ++h) {
        *reinterpret_cast<f32x4*>(&red[l_row][128 * h + l_col]) = f32x4{bs[h][0], bs[h][1], bs[h][2], bs[h][3]};
        *reinterpret_cast<f32x4*>(&red[l_row][128 * h + l_col + 4]) = f32x4{bs[h][4], bs[h][5], bs[h][6], bs[h][7]};
      }
    }
    role_barrier();                                    // (the MFMA waves pass it on their way to the slab stores)
    if (do_bias && lt < 128 * DT && d0 + lt < d) {
      const float (*red)[256] = reinterpret_cast<const float (*)[256]>(&Pimg[0][0][0][0]);
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < BK; ++r) sacc += red[r][lt];
      bias_slabs[c * (int64_t)d + d0 + lt] = sacc;
    }
  } else {
    // ------------------------------------------------ MFMA waves ------------------------------------------------
    constexpr int NI = DT;                  // 32-row f blocks per wave
    const int wf = DT == 2 ? wave >> 1 : wave;                 // f block of 64 (DT == 2) / 32 (DT == 1) rows
    const int wd = DT == 2 ? wave & 1 : 0;
    const int f_lo = DT == 2 ? wf * 64 : wf * 32;              // first f row of this wave
    const int fr = lane & 31, fk = lane >> 5;
    f32x16 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    role_barrier();                                    // stage 0 holds tile 0
    for (int64_t t = 0; t < nsteps; ++t) {
      const int buf = (int)(t & 1);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        bf16x8 as[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) as[pl] = tr_read8(Pimg[buf][pl][f_lo >> 7], (f_lo & 127) + i * 32, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x8 b3[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) b3[pl] = tr_read8(Gimg[buf][pl][wd], (MP_WPC_ABL & 4) ? 0 : j * 32, lane);
#if !(MP_WPC_ABL & 2)
          mfma6(acc[i][j], as, b3);
#else
          asm volatile("" ::"v"(b3[0]), "v"(b3[1]), "v"(b3[2]), "v"(as[0]), "v"(as[1]), "v"(as[2]));
#endif
          // (one fragment set at a time: hoisting the reads of all four d tiles costs 121 registers of scratch; the
          // SIMD's other MFMA wave covers the LDS latency)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      role_barrier();                                  // step t done: the loaders may overwrite this stage
    }
    role_barrier();                                    // (the loaders' bias reduction)
    float* slab = slabs + c * (int64_t)F * d;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = d0 + wd * 128 + j * 32 + fr;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = f0 + f_lo + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
          if (row < F && col < d) slab[(int64_t)row * d + col] = acc[i][j][r];
        }
      }
    }
  }
}

// Narrow inputs (F <= 8: the reference's synthetic datasets carry node_feature = [1.], so the first layer of every
// model has F = 1): dW[f, :] = sum_m P[m, f] g[m, :] is a handful of weighted column sums — HBM-bound, no use for
// 128 x 128 MFMA tiles (17.9 ms at 10^7 x 1 x 256 through them).  A thread owns one output column over a chunk of
// rows; partial slabs are added in chunk order like the wide kernel's.  Same fused ReLU mask / masked-gradient output /
// bias gradient.
template <bool RELU>
__global__ __launch_bounds__(kBlock) void narrow_wgrad_kernel(const float* __restrict__ P, int64_t ldp,
                                                              const float* G, int64_t ldg,
                                                              const float* __restrict__ Y, int64_t ldy, float* GM,
                                                              int64_t ldgm, int64_t M, int32_t F, int32_t d,
                                                              int64_t chunk, float* __restrict__ slabs,
                                                              float* __restrict__ bias_slabs) {
  const int c = blockIdx.y * kBlock + threadIdx.x;
  const int64_t ch = blockIdx.x;
  const int64_t mb = ch * chunk;
  const int64_t me = mb + chunk < M ? mb + chunk : M;
  if (c >= d) return;
  float acc[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) acc[f] = 0.f;
  float bs = 0.f;
  constexpr int UR = 4;
  for (int64_t m0 = mb; m0 < me; m0 += UR) {
    float gv[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t m = m0 + u;
      float v = 0.f;
      if (m < me) {
        v = G[m * ldg + c];
        if constexpr (RELU) {
          v = Y[m * ldy + c] > 0.f ? v : 0.f;
          if (GM != nullptr) GM[m * ldgm + c] = v;
        }
      }
      gv[u] = v;
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t m = m0 + u;
      if (m < me) {
        bs += gv[u];
        const float* pr = P + m * ldp;          // the same address for every lane: one broadcast load
#pragma unroll
        for (int f = 0; f < 8; ++f)
          if (f < F) acc[f] = fmaf(pr[f], gv[u], acc[f]);
      }
    }
  }
  float* slab = slabs + ch * (int64_t)F * d;
  for (int f = 0; f < F; ++f) slab[(int64_t)f * d + c] = acc[f];
  if (bias_slabs != nullptr) bias_slabs[ch * (int64_t)d + c] = bs;
}

// The same with 16-byte accesses (d % 4 == 0, d <= 1024, aligned rows): a thread owns four adjacent columns, the block's
// 256 / (d / 4) row lanes take rows round-robin and are added in lane order at the end (fixed order: reproducible).
// 128 bytes in flight per thread and operand instead of 16: 11.6 -> ~5 ms at 10^7 x 1 x 256 with the mask.
// FT: rows of dW the kernel carries (1 for F == 1, the datasets' node_feature = [1.]; 8 for F <= 8).  All loads of an
// iteration — UR rows of g, of y and of P — are issued before the first use.
template <bool RELU, int FT>
__global__ __launch_bounds__(kBlock) void narrow_wgrad_vec_kernel(const float* __restrict__ P, int64_t ldp,
                                                                  const float* G, int64_t ldg,
                                                                  const float* __restrict__ Y, int64_t ldy, float* GM,
                                                                  int64_t ldgm, int64_t M, int32_t F, int32_t d,
                                                                  int64_t chunk, float* __restrict__ slabs,
                                                                  float* __restrict__ bias_slabs) {
  __shared__ __attribute__((aligned(16))) float red[kBlock * 4 * (FT + 1)];   // [row lane][f or bias][d]
  const int ncg = d >> 2;                       // column groups of four
  const int nrl = kBlock / ncg;                 // row lanes
  const int cg = threadIdx.x % ncg, rl = threadIdx.x / ncg;
  const int64_t ch = blockIdx.x;
  const int64_t mb = ch * chunk;
  const int64_t me = mb + chunk < M ? mb + chunk : M;
  f32x4 acc[FT];
#pragma unroll
  for (int f = 0; f < FT; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bs = {0.f, 0.f, 0.f, 0.f};
  constexpr int UR = FT == 1 ? 8 : 4;
  for (int64_t m0 = mb + rl; m0 < me; m0 += (int64_t)UR * nrl) {
    f32x4 gv[UR], yv[UR];
    float pv[UR][FT];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      int64_t m = m0 + (int64_t)u * nrl;
      m = m < me ? m : me - 1;                  // a clamped row is loaded and dropped below: no branches around the loads
      gv[u] = *reinterpret_cast<const f32x4*>(G + m * ldg + 4 * cg);
      if constexpr (RELU) yv[u] = *reinterpret_cast<const f32x4*>(Y + m * ldy + 4 * cg);
      const float* pr = P + m * ldp;            // one address per row lane: a broadcast load
#pragma unroll
      for (int f = 0; f < FT; ++f) pv[u][f] = pr[f < F ? f : F - 1];
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t m = m0 + (int64_t)u * nrl;
      if (m < me) {
        f32x4 v = gv[u];
        if constexpr (RELU) {
          v[0] = yv[u][0] > 0.f ? v[0] : 0.f; v[1] = yv[u][1] > 0.f ? v[1] : 0.f;
          v[2] = yv[u][2] > 0.f ? v[2] : 0.f; v[3] = yv[u][3] > 0.f ? v[3] : 0.f;
          if (GM != nullptr) *reinterpret_cast<f32x4*>(GM + m * ldgm + 4 * cg) = v;
        }
        bs += v;
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f] += pv[u][f] * v;   // rows f >= F accumulate a copy of row F - 1, never stored
      }
    }
  }
#pragma unroll
  for (int f = 0; f < FT; ++f) *reinterpret_cast<f32x4*>(&red[((rl * (FT + 1) + f) * ncg + cg) * 4]) = acc[f];
  *reinterpret_cast<f32x4*>(&red[((rl * (FT + 1) + FT) * ncg + cg) * 4]) = bs;
  __syncthreads();
  if (rl == 0) {
    float* slab = slabs + ch * (int64_t)F * d;
    for (int f = 0; f <= FT; ++f) {
      if (f < F || (f == FT && bias_slabs != nullptr)) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < nrl; ++r) t += *reinterpret_cast<const f32x4*>(&red[((r * (FT + 1) + f) * ncg + cg) * 4]);
        if (f < FT) *reinterpret_cast<f32x4*>(slab + (int64_t)f * d + 4 * cg) = t;
        else *reinterpret_cast<f32x4*>(bias_slabs + ch * (int64_t)d + 4 * cg) = t;
      }
    }
  }
}

// Narrow OUTPUTS (d <= 16: the classifier head, 256 -> 7 / 10 classes): dW[f, :] = sum_m P[m, f] g[m, :] is a stream
// over P with d FMAs per element — HBM-bound (10 GB at 10^7 x 256), and the library's split-K GEMM for it runs at
// 1.4 TB/s (7.2 ms).  A thread owns four adjacent f columns and all d outputs (4 x 16 accumulators); the block's row
// lanes take rows round-robin and are added in lane order at the end; g rows are broadcast loads.
__global__ __launch_bounds__(kBlock) void narrow_out_wgrad_kernel(const float* __restrict__ P, int64_t ldp,
                                                                  const float* __restrict__ G, int64_t ldg, int64_t M,
                                                                  int32_t F, int32_t d, int64_t chunk,
                                                                  float* __restrict__ slabs,
                                                                  float* __restrict__ bias_slabs) {
  __shared__ float red[64][4][17];              // [column group][f][d (+ pad)]; bias partials in red[0][0][..] afterwards
  const int ncg = F >> 2;                       // column groups handled per pass (<= 64: F <= 256 per pass)
  const int64_t ch = blockIdx.x;
  const int64_t mb = ch * chunk;
  const int64_t me = mb + chunk < M ? mb + chunk : M;
  float* slab = slabs + ch * (int64_t)F * d;
  for (int cg0 = 0; cg0 < ncg; cg0 += 64) {     // F > 256: further passes over the rows
    const int ng = ncg - cg0 < 64 ? ncg - cg0 : 64;
    const int nrl = kBlock / 64;                // 4 row lanes
    const int cgi = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const bool on = cgi < ng;
    const int f0 = 4 * (cg0 + cgi);
    float acc[4][16];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[f][j] = 0.f;
    float bs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bs[j] = 0.f;
    constexpr int UR = 4;
    for (int64_t m0 = mb + rl; m0 < me; m0 += (int64_t)UR * nrl) {
      f32x4 pv[UR];
      float gv[UR][16];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        int64_t m = m0 + (int64_t)u * nrl;
        m = m < me ? m : me - 1;                // a clamped row is loaded and dropped below
        pv[u] = on ? *reinterpret_cast<const f32x4*>(P + m * ldp + f0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) gv[u][j] = G[m * ldg + (j < d ? j : d - 1)];   // one address per row lane
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        if (m0 + (int64_t)u * nrl < me) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            bs[j] += gv[u][j];
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f][j] = fmaf(pv[u][f], gv[u][j], acc[f][j]);
          }
        }
      }
    }
    // the four row lanes add up in lane order
    for (int r = 0; r < nrl; ++r) {
      if (rl == r && on) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int j = 0; j < 16; ++j) red[cgi][f][j] = (r == 0 ? 0.f : red[cgi][f][j]) + acc[f][j];
      }
      __syncthreads();
    }
    if (rl == 0 && on) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
        for (int j = 0; j < d; ++j) slab[(int64_t)(f0 + f) * d + j] = red[cgi][f][j];
    }
    __syncthreads();
    if (bias_slabs != nullptr && cg0 == 0) {    // column sums of g: every thread of a row lane holds the same partial
      for (int r = 0; r < nrl; ++r) {
        if (rl == r && cgi == 0) {
#pragma unroll
          for (int j = 0; j < 16; ++j) red[0][0][j] = (r == 0 ? 0.f : red[0][0][j]) + bs[j];
        }
        __syncthreads();
      }
      if (threadIdx.x < d) bias_slabs[ch * (int64_t)d + threadIdx.x] = red[0][0][threadIdx.x];
      __syncthreads();
    }
  }
}

// out[i] = sum over slabs of slabs[c][i], in a fixed order (bitwise reproducible): a workgroup owns 16 consecutive
// elements, its 16 slab lanes each add every 16th slab through four independent running sums, and the 64 partial sums
// of an element are combined lane by lane.  (One thread per element walking all slabs in a dependent chain took 1.8 ms
// for the 4 883 slabs of a first-layer weight gradient at 10^7 rows.)
__global__ __launch_bounds__(kBlock) void slab_reduce_kernel(const float* __restrict__ slabs, int64_t n_slab,
                                                             int64_t elems, float* __restrict__ out) {
  __shared__ float part[16][16];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  for (int64_t i0 = (int64_t)blockIdx.x * 16; i0 < elems; i0 += (int64_t)gridDim.x * 16) {
    const int64_t i = i0 + e;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < elems) {
      int64_t c = sl;
      for (; c + 48 < n_slab; c += 64) {
        a0 += slabs[c * elems + i];
        a1 += slabs[(c + 16) * elems + i];
        a2 += slabs[(c + 32) * elems + i];
        a3 += slabs[(c + 48) * elems + i];
      }
      for (; c < n_slab; c += 16) a0 += slabs[c * elems + i];
    }
    part[sl][e] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && i < elems) {
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += part[r][e];
      out[i] = acc;
    }
    __syncthreads();
  }
}

static int64_t wgrad_chunk(int64_t M, int32_t F, int32_t d) {
  // narrow inputs: a slab is F * d floats, so many small chunks cost nothing and balance the chip
  if (F <= 8) return 2048;
  if (d <= 16) return 4096;                     // narrow outputs: likewise
  // enough chunks to fill the chip with 128 x 128 tiles, few enough that the slab pass stays small
  int64_t chunk = 4096;
  while (chunk < 65536 && ceil_div(M, chunk) > 1024) chunk *= 2;
  return chunk;
}

static bool al16(const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; }

// the producer/consumer weight gradient: one contiguous node range per workgroup, one workgroup per CU (never more
// chunks than wgrad_chunk gives: pc_chunk >= 4096 and >= M / kNumCU)
static int64_t wgrad_pc_chunk(int64_t M) {
  int64_t c = ceil_div(M, (int64_t)kNumCU);
  c = ceil_div(c, (int64_t)BK) * BK;
  return c < 4096 ? 4096 : c;
}
static bool wgrad_no_pc() {   // MP_WGRAD_PC=0: the 256 x 128 tile kernel (A/B studies), read per call
  const char* e = getenv("MP_WGRAD_PC");
  return e && e[0] == '0';
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_dense_fused_f32(const float* P, int64_t ldp, const float* W, const float* Q, int64_t ldq,
                       const float* Wid, const float* bias, int act, float* out, int64_t ldo, int64_t M,
                       int32_t F, int32_t d, mp_stream_t stream) {
  if (M < 0 || F <= 0 || d <= 0 || !P || !W || !out || ldp < F || ldo < d) return MP_ERR_INVALID_ARG;
  if ((Q == nullptr) != (Wid == nullptr) || (Q && ldq < F)) return MP_ERR_INVALID_ARG;
  if (act != MP_ACT_NONE && act != MP_ACT_RELU) return MP_ERR_INVALID_ARG;
  if (M == 0) return MP_OK;
  // 16-byte vector loads need operand widths in multiples of 8 (F) / 4 (d) and aligned rows; else scalar loaders
  const bool vec = !(F % 8 || d % 4 || ldp % 4 || (Q && ldq % 4)) && al16(P) && al16(W) && al16(Q) && al16(Wid);
  const int tn = 2;   // 128-column block tile (a 256-column tile needs 297 registers, one wave per SIMD: 25 % slower, not built)
  const int bn = 64 * tn;
  const int64_t nblocks = ceil_div(d, bn) * ceil_div(M, BM);
  if (nblocks >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  dim3 grid((unsigned)nblocks);
  hipStream_t st = as_stream(stream);
#define MP_DENSE(DUALV, TNV, VECV)                                                                        \
  hipLaunchKernelGGL((dense_fused_kernel<DUALV, TNV, VECV>), grid, dim3(kBlock), 0, st, P, ldp, W, Q, ldq, Wid, \
                     bias, act, out, ldo, M, F, d)
  if (!vec) { if (Q) MP_DENSE(true, 2, false); else MP_DENSE(false, 2, false); }
  else if (Q) MP_DENSE(true, 2, true);
  else MP_DENSE(false, 2, true);
#undef MP_DENSE
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_dense_wgrad_ws_bytes(int64_t M, int32_t F, int32_t d, size_t* bytes_host) {
  if (!bytes_host || M < 0 || F <= 0 || d <= 0) return MP_ERR_INVALID_ARG;
  *bytes_host = (size_t)ceil_div(M > 0 ? M : 1, wgrad_chunk(M, F, d)) * ((size_t)F * d + d) * 4;   // + the bias partials
  return MP_OK;
}

static int wgrad_common(const float* P, int64_t ldp, const float* G, int64_t ldg, const float* Y, int64_t ldy,
                        float* GM, int64_t ldgm, int64_t M, int32_t F, int32_t d, float* dW, float* dbias, void* ws,
                        size_t ws_bytes, hipStream_t st) {
  if (M < 0 || F <= 0 || d <= 0 || !dW || (M > 0 && (!P || !G)) || ldp < F || ldg < d) return MP_ERR_INVALID_ARG;
  if (Y && ldy < d) return MP_ERR_INVALID_ARG;
  if (GM && (!Y || ldgm < d)) return MP_ERR_INVALID_ARG;
  if (M == 0) {
    MP_HIP(hipMemsetAsync(dW, 0, (size_t)F * d * 4, st));
    if (dbias) MP_HIP(hipMemsetAsync(dbias, 0, (size_t)d * 4, st));
    return MP_OK;
  }
  const bool vec = !(F % 4 || d % 4 || ldp % 4 || ldg % 4 || (Y && ldy % 4) || (GM && ldgm % 4)) && al16(P) &&
                   al16(G) && al16(Y) && al16(GM);
  const int64_t chunk = wgrad_chunk(M, F, d);
  const int64_t n_chunk = ceil_div(M, chunk);
  const size_t need = (size_t)n_chunk * ((size_t)F * d + d) * 4;
  if (!ws || ws_bytes < need) return MP_ERR_WORKSPACE;
  float* bias_slabs = dbias ? (float*)ws + (size_t)n_chunk * F * d : nullptr;
  const int64_t tiles = ceil_div(F, 128) * ceil_div(d, 128);
  if (tiles * n_chunk >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  if (F <= 8) {   // weighted column sums: one thread per output column over a chunk of rows
    const dim3 ngrid((unsigned)n_chunk, (unsigned)ceil_div(d, kBlock));
    const bool nvec = d % 4 == 0 && d <= 4 * kBlock && kBlock % (d / 4) == 0 && ldg % 4 == 0 && al16(G) &&
                      (!Y || (ldy % 4 == 0 && al16(Y))) && (!GM || (ldgm % 4 == 0 && al16(GM))) &&
                      al16(ws) && (!bias_slabs || al16(bias_slabs));
    if (nvec) {
#define MP_NARROW(RELUV, FTV)                                                                                       \
  hipLaunchKernelGGL((narrow_wgrad_vec_kernel<RELUV, FTV>), dim3((unsigned)n_chunk), dim3(kBlock), 0, st, P, ldp, G, \
                     ldg, Y, ldy, GM, ldgm, M, F, d, chunk, (float*)ws, bias_slabs)
      if (Y) { if (F == 1) MP_NARROW(true, 1); else MP_NARROW(true, 8); }
      else { if (F == 1) MP_NARROW(false, 1); else MP_NARROW(false, 8); }
#undef MP_NARROW
    } else if (Y)
      hipLaunchKernelGGL(narrow_wgrad_kernel<true>, ngrid, dim3(kBlock), 0, st, P, ldp, G, ldg, Y, ldy, GM, ldgm, M, F,
                         d, chunk, (float*)ws, bias_slabs);
    else
      hipLaunchKernelGGL(narrow_wgrad_kernel<false>, ngrid, dim3(kBlock), 0, st, P, ldp, G, ldg, Y, ldy, GM, ldgm, M,
                         F, d, chunk, (float*)ws, bias_slabs);
    MP_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(flat_grid((int64_t)F * d * 16)), dim3(kBlock), 0, st, (const float*)ws,
                       n_chunk, (int64_t)F * d, dW);
    MP_LAUNCH_CHECK();
    if (dbias) {
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(flat_grid((int64_t)d * 16)), dim3(kBlock), 0, st,
                         (const float*)bias_slabs, n_chunk, (int64_t)d, dbias);
      MP_LAUNCH_CHECK();
    }
    return MP_OK;
  }
  if (!Y && d <= 16 && F % 4 == 0 && ldp % 4 == 0 && al16(P)) {   // the classifier head's shape
    hipLaunchKernelGGL(narrow_out_wgrad_kernel, dim3((unsigned)n_chunk), dim3(kBlock), 0, st, P, ldp, G, ldg, M, F, d,
                       chunk, (float*)ws, bias_slabs);
    MP_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(flat_grid((int64_t)F * d * 16)), dim3(kBlock), 0, st, (const float*)ws,
                       n_chunk, (int64_t)F * d, dW);
    MP_LAUNCH_CHECK();
    if (dbias) {
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(flat_grid((int64_t)d * 16)), dim3(kBlock), 0, st,
                         (const float*)bias_slabs, n_chunk, (int64_t)d, dbias);
      MP_LAUNCH_CHECK();
    }
    return MP_OK;
  }
  if (vec && F > 128 && d >= 64 && F % 8 == 0 && d % 8 == 0 && !wgrad_no_pc()) {   // loaders + MFMA waves, the whole gradient per workgroup
    const int64_t pc_chunk = wgrad_pc_chunk(M);
    const int64_t n_pc = ceil_div(M, pc_chunk);            // <= n_chunk: the workspace of mp_dense_wgrad_ws_bytes holds it
    float* pc_bias = dbias ? (float*)ws + (size_t)n_pc * F * d : nullptr;
    const int64_t pc_tiles = ceil_div(F, 256) * ceil_div(d, d > 128 ? 256 : 128);
    if (n_pc * pc_tiles >= INT32_MAX) return MP_ERR_UNSUPPORTED;
#define MP_WPC(RELUV, DTV)                                                                                          \
  hipLaunchKernelGGL((dense_wgrad_pc_kernel<RELUV, DTV>), dim3((unsigned)(n_pc * pc_tiles)), dim3(WPC_THREADS), 0, st, P, ldp, G, \
                     ldg, Y, ldy, GM, ldgm, M, F, d, pc_chunk, (float*)ws, pc_bias)
    if (d > 128) { if (Y) MP_WPC(true, 2); else MP_WPC(false, 2); }
    else { if (Y) MP_WPC(true, 1); else MP_WPC(false, 1); }
#undef MP_WPC
    MP_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(flat_grid((int64_t)F * d * 16)), dim3(kBlock), 0, st, (const float*)ws,
                       n_pc, (int64_t)F * d, dW);
    MP_LAUNCH_CHECK();
    if (dbias) {
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(flat_grid((int64_t)d * 16)), dim3(kBlock), 0, st,
                         (const float*)pc_bias, n_pc, (int64_t)d, dbias);
      MP_LAUNCH_CHECK();
    }
    return MP_OK;
  }
  if (vec && F > 128) {   // the 256 x 128 tile
    const int64_t wtiles = ceil_div(F, 256) * ceil_div(d, 128);
    if (wtiles * n_chunk >= INT32_MAX) return MP_ERR_UNSUPPORTED;
    const dim3 wgrid((unsigned)(wtiles * n_chunk));
    if (Y)
      hipLaunchKernelGGL(dense_wgrad_wide_kernel<true>, wgrid, dim3(kBlock), 0, st, P, ldp, G, ldg, Y, ldy, GM, ldgm, M,
                         F, d, chunk, (float*)ws, bias_slabs);
    else
      hipLaunchKernelGGL(dense_wgrad_wide_kernel<false>, wgrid, dim3(kBlock), 0, st, P, ldp, G, ldg, Y, ldy, GM, ldgm, M,
                         F, d, chunk, (float*)ws, bias_slabs);
    MP_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(flat_grid((int64_t)F * d * 16)), dim3(kBlock), 0, st, (const float*)ws,
                       n_chunk, (int64_t)F * d, dW);
    MP_LAUNCH_CHECK();
    if (dbias) {
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(flat_grid((int64_t)d * 16)), dim3(kBlock), 0, st,
                         (const float*)bias_slabs, n_chunk, (int64_t)d, dbias);
      MP_LAUNCH_CHECK();
    }
    return MP_OK;
  }
  const dim3 grid((unsigned)(tiles * n_chunk));
#define MP_WGRAD(VECV, RELUV)                                                                                     \
  hipLaunchKernelGGL((dense_wgrad_kernel<VECV, RELUV>), grid, dim3(kBlock), 0, st, P, ldp, G, ldg, Y, ldy, GM,     \
                     ldgm, M, F, d, chunk, (float*)ws, bias_slabs)
  if (Y) { if (vec) MP_WGRAD(true, true); else MP_WGRAD(false, true); }
  else { if (vec) MP_WGRAD(true, false); else MP_WGRAD(false, false); }
#undef MP_WGRAD
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(flat_grid((int64_t)F * d * 16)), dim3(kBlock), 0, st, (const float*)ws,
                     n_chunk, (int64_t)F * d, dW);
  MP_LAUNCH_CHECK();
  if (dbias) {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(flat_grid((int64_t)d * 16)), dim3(kBlock), 0, st,
                       (const float*)bias_slabs, n_chunk, (int64_t)d, dbias);
    MP_LAUNCH_CHECK();
  }
  return MP_OK;
}

int mp_dense_wgrad_f32(const float* P, int64_t ldp, const float* G, int64_t ldg, int64_t M, int32_t F,
                       int32_t d, float* dW, float* dbias, void* ws, size_t ws_bytes, mp_stream_t stream) {
  return wgrad_common(P, ldp, G, ldg, nullptr, 0, nullptr, 0, M, F, d, dW, dbias, ws, ws_bytes, as_stream(stream));
}

int mp_dense_wgrad_relu_f32(const float* P, int64_t ldp, const float* G, int64_t ldg, const float* Y, int64_t ldy,
                            float* GM, int64_t ldgm, int64_t M, int32_t F, int32_t d, float* dW, float* dbias,
                            void* ws, size_t ws_bytes, mp_stream_t stream) {
  if (!Y) return MP_ERR_INVALID_ARG;
  return wgrad_common(P, ldp, G, ldg, Y, ldy, GM, ldgm, M, F, d, dW, dbias, ws, ws_bytes, as_stream(stream));
}

}  // extern "C"

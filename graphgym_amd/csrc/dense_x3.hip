// The dense feature transform at its hot shape — out[M, d] = act(P[M, F] W[F, d] + bias), d = 64 / 128 / 256,
// F % 32 == 0 — as a streaming kernel: the Linear / kernel product of every layer on the path
// (x @ kernel of gcn_id, TfgIDLayer.py:510-523; GeneralLayer's Linear, layer.py:136-147; the GIN MLPs,
// idconv.py:371-399) and, with W^T, the input gradient of the same product.
//
// P is read once and out written once (20 GB at 10^7 x 256 x 256); the product runs on the bf16 matrix pipe with
// three-way split operands (bf16x3.h):
//   * persistent workgroups of 8 waves walk 256-row blocks; wave w owns rows 32 w .. 32 w + 31 and ALL d columns
//     (accumulators: d / 32 tiles of 32 x 32 = 128 registers at d = 256), so every row of P is split into its three bf16
//     planes exactly once, in the lanes that feed it to the MFMAs (the split of step n + 1 rides under the MFMAs of n);
//   * W arrives pre-split (mp_split_w_bf16x3: [3][F / 8][d][8] bf16, the operand order of the MFMA) and both operands
//     are staged by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass): P as raw fp32 in 32-k
//     stages of 8-row x 128-byte pieces — full cache lines; the chunk swizzle goes on the SOURCE address and on the
//     fragment read, the LDS image stays lane-linear — W in 16-k slices shared by the eight waves;
//   * the P stage two ahead stays in flight across the one barrier per 16-k step (raw s_barrier + counted vmcnt), and a
//     row block's stores drain during the first step of the next one;
//   * the W fragment is the A operand of the MFMA, so a lane holds four consecutive output columns of ITS row; tiles
//     pass through a per-wave LDS image and leave as whole 128-byte lines.
// LDS at d = 256: 2 x 32 KiB (P) + 2 x 24 KiB (W) + 8 x 4.5 KiB (output tiles) + bias = 149 KiB: one workgroup (two
// waves per SIMD) per CU.
//
// Measured at 10^7 x 256 x 256 (scripts/dbg/x3_ablate.hip, profiles/r02_dense_x3_ablation.txt): 7.0 ms against 9.7 ms
// for the general kernel of gemm.hip and 9.9 ms for the library's fp32 GEMM.  The MFMAs alone take 4.0 ms and the
// memory side alone 3.9 ms, but the two overlap only partly: a vector memory instruction holds its wave at issue while
// the CU's address path is busy (~50 cycles per 1 KiB piece or store with every wave issuing, ~2800 cycles of the
// ~3100 an MFMA-bound step has), and a stalled wave issues no MFMAs.  Tried and measured no better: spreading the
// memory instructions over the MFMA groups (8.2 ms), skewing the waves' row-block phases (7.8), two 4-wave workgroups
// per CU (7.9), the two waves of a SIMD taking their memory phase at opposite ends of the step (7.1; with the W pieces
// balanced over both halves through a 3-slot ring the accumulators spill).
#include "common.h"
#include "bf16x3.h"

namespace mp {

constexpr int X3_WAVES = 8;
constexpr int X3_THREADS = 64 * X3_WAVES;
constexpr int X3_BM = 32 * X3_WAVES;
constexpr int X3_ASTAGE = X3_BM * 128;       // BM rows x 32 k fp32
constexpr int X3_OROW = 144;                 // one 32-column row of an output tile in LDS: 128 B + 16 B pad
constexpr int X3_OSTAGE = 32 * X3_OROW;      // a wave's 32 x 32 output tile on its way to full-line stores

typedef __attribute__((address_space(3))) void x3_lds_void;
typedef __attribute__((address_space(1))) const void x3_glb_void;
typedef float x3_f32x4 __attribute__((ext_vector_type(4)));

// one LDS-DMA piece: lane l's 16 bytes at gsrc land at lds_base + 16 l (lds_base wave-uniform)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_base) {
  __builtin_amdgcn_global_load_lds((x3_glb_void*)gsrc, (x3_lds_void*)lds_base, 16, 0, 0);
}

// ABL: ablation bits for scripts/dbg/x3_ablate.hip only (1 no P loads after the prologue, 2 no W loads, 4 no split,
// 8 no stores, 16 no barrier, 64 report the shader cycles of workgroup 0 in out[0..1], 128 no MFMAs — timing experiments with wrong results); the library builds ABL = 0.
template <int NCB, int ABL = 0>
__global__ __launch_bounds__(X3_THREADS, 1) void dense_x3_kernel(const float* __restrict__ P, int64_t ldp,
                                                                 const unsigned char* __restrict__ Ws,
                                                                 const float* __restrict__ bias, int act,
                                                                 float* __restrict__ out, int64_t ldo, int64_t M,
                                                                 int32_t F) {
  constexpr int d = 32 * NCB;
  constexpr int BSTAGE = 96 * d;             // [3 planes][2 k-chunks of 8][d columns][8 bf16]
  constexpr int BPIECES = 3 * NCB;           // 1 KiB pieces of one W stage
  constexpr int QP = NCB / 2;                // pieces per (plane, k-chunk)
  static_assert(NCB >= 2 && NCB % 2 == 0, "a (plane, k-chunk) slice of W must be whole 1 KiB pieces");
  // ONE shared array: a second __shared__ object beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * X3_ASTAGE + 2 * BSTAGE + X3_WAVES * X3_OSTAGE + 4 * d];
  unsigned char* const Abuf = lds;
  unsigned char* const Bbuf = lds + 2 * X3_ASTAGE;

  const int tid = threadIdx.x, lane = tid & 63, lane_c = lane;
  const uint64_t x3_c0 = (ABL & 64) ? __builtin_readcyclecounter() : 0;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA bases and piece loops stay wave-uniform
  const int S = F >> 4;                      // 16-k steps per row block (even: F % 32 == 0)
  const int T = F >> 5;                      // 32-k stages of P per row block (>= 2)
  const int64_t nrb = (M + X3_BM - 1) / X3_BM;
  if ((int64_t)blockIdx.x >= nrb) return;
  const int nloc = (int)((nrb - blockIdx.x + gridDim.x - 1) / gridDim.x);   // row blocks of this workgroup
  const int nsteps = nloc * S;

  // One piece of P stage `stage` (k = 32 stage .. + 31) of row-block iteration `it`: rows 8 i .. 8 i + 7 of this wave's
  // 32, 128 B each.  LDS position (row, chunk j) holds global chunk j ^ ((row >> 1) & 7): 16 consecutive rows then read
  // one logical chunk from 16 different 16-byte bank groups
  // (addresses are a wave-uniform 64-bit base plus a 32-bit lane offset: the loop keeps few live address registers)
  const uint32_t ldp_b = (uint32_t)ldp * 4u, ldo_b = (uint32_t)ldo * 4u;
  auto issue_A = [&](int it, int stage, int buf, int i) {
    int lane = lane_c;                       // opaque copy (see flush_tile)
    asm volatile("" : "+v"(lane));
    const int p_row = lane >> 3;
    const int64_t r0 = ((int64_t)blockIdx.x + (int64_t)it * gridDim.x) * X3_BM;      // first row of the row block
    const int64_t left = M - r0;                                                    // rows of it that exist (>= 1)
    const int row = 32 * wave + 8 * i + p_row;
    const int gch = (lane & 7) ^ ((row >> 1) & 7);
    const int srow = row < left ? row : (int)left - 1;   // rows past the end read a valid row; their results are not stored
    const unsigned char* base = reinterpret_cast<const unsigned char*>(P) + r0 * (int64_t)ldp_b;
    glds16(base + ((uint32_t)srow * ldp_b + (uint32_t)(128 * stage + 16 * gch)),
           Abuf + buf * X3_ASTAGE + (32 * wave + 8 * i) * 128);
  };
  // W slice s (k = 16 s .. + 15): piece p = (plane, k-chunk h, quarter q) is 1 KiB contiguous on both sides; wave w
  // moves pieces w, w + 8, ...
  constexpr int NBJ = (BPIECES + X3_WAVES - 1) / X3_WAVES;
  auto issue_B = [&](int s, int buf) {
    int lane = lane_c;                       // opaque copy (see flush_tile)
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      const int p = wave + X3_WAVES * j;
      if (p < BPIECES) {
        const int ph = p / QP, q = p % QP;
        const size_t src = (size_t)(ph >> 1) * ((size_t)F * d * 2) + (size_t)(2 * s + (ph & 1)) * (d * 16) + q * 1024;
        glds16(Ws + src + lane * 16, Bbuf + buf * BSTAGE + p * 1024);
      }
    }
  };

  // this lane's fragment of step n (8 k-values of its row) from the P buffer of stage slot n >> 1, split three ways
  auto read_split = [&](int n, bf16x8 (&p3)[3]) {
    int lane = lane_c;                       // opaque copy (see flush_tile)
    asm volatile("" : "+v"(lane));
    const int f_row = 32 * wave + (lane & 31), f_h = lane >> 5, f_sw = (f_row >> 1) & 7;
    const unsigned char* A = Abuf + ((n >> 1) & 1) * X3_ASTAGE + f_row * 128;
    const int g0 = 4 * (n & 1) + 2 * f_h;
    const x3_f32x4 x0 = *reinterpret_cast<const x3_f32x4*>(A + 16 * (g0 ^ f_sw));
    const x3_f32x4 x1 = *reinterpret_cast<const x3_f32x4*>(A + 16 * ((g0 + 1) ^ f_sw));
    const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    if constexpr (ABL & 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) p3[0][i] = p3[1][i] = p3[2][i] = (__bf16)xv[i];
    } else {
      split3_bf16(xv, p3[0], p3[1], p3[2]);
    }
  };
  // Epilogue of one 32 x 32 tile of row-block iteration `it`.  C/D layout with the W fragment as the A operand:
  // j = lane & 31 is the row of P, i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) the output column inside the tile, so
  // register quad q = r >> 2 is columns 8 q + 4 (lane >> 5) .. + 3 of one row: 16-byte stores
  f32x16 acc[NCB];
  // The tile goes through a per-wave LDS image so that every store instruction writes eight whole 128-byte lines
  // (lane l: row 8 i + l / 8, columns 4 (l % 8) .. + 3); stored straight from the registers an instruction would
  // touch 32 lines with 32 bytes each.
  unsigned char* const Obuf = lds + 2 * X3_ASTAGE + 2 * BSTAGE + wave * X3_OSTAGE;
  // bias in LDS: a global load between the stores would make the compiler wait for it — and, vmcnt retiring in order,
  // for every store before it
  float* const bias_l = reinterpret_cast<float*>(lds + 2 * X3_ASTAGE + 2 * BSTAGE + X3_WAVES * X3_OSTAGE);
  if (tid < d) bias_l[tid] = bias != nullptr ? bias[tid] : 0.f;
  auto flush_tile = [&](int it, int cb) {
    if ((ABL & 8) && act != 77) return;
    int lane = lane_c;                       // opaque copy: keeps the address arithmetic below out of the loop-invariant
    asm volatile("" : "+v"(lane));           // set, whose ~100 hoisted registers would spill the accumulators
    const int f_h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const x3_f32x4 b4 = *reinterpret_cast<const x3_f32x4*>(bias_l + cb * 32 + 8 * q + 4 * f_h);
      x3_f32x4 v = {acc[cb][4 * q] + b4[0], acc[cb][4 * q + 1] + b4[1], acc[cb][4 * q + 2] + b4[2],
                    acc[cb][4 * q + 3] + b4[3]};
      if (act == MP_ACT_RELU) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      *reinterpret_cast<x3_f32x4*>(Obuf + (lane & 31) * X3_OROW + (8 * q + 4 * f_h) * 4) = v;
    }
    const int64_t r0 = ((int64_t)blockIdx.x + (int64_t)it * gridDim.x) * X3_BM + 32 * wave;   // the wave's first row
    const int64_t left = M - r0;
    const int nrow = left >= 32 ? 32 : (left > 0 ? (int)left : 0);
    unsigned char* const base = reinterpret_cast<unsigned char*>(out) + r0 * (int64_t)ldo_b + cb * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const x3_f32x4 v = *reinterpret_cast<const x3_f32x4*>(Obuf + (8 * i + (lane >> 3)) * X3_OROW + 16 * (lane & 7));
      if (8 * i + (lane >> 3) < nrow)
        *reinterpret_cast<x3_f32x4*>(base + ((uint32_t)(8 * i + (lane >> 3)) * ldo_b + (uint32_t)(16 * (lane & 7)))) = v;
    }
  };

  // prologue: W slice 0, the first two P stages (T >= 2), this lane's first fragment
  issue_B(0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_A(0, 0, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_A(0, 1, 1, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 p3[3];
  read_split(0, p3);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
  // the next P stage to request is stage slot 2 (slot m = stage m % T of row-block iteration m / T)
  int pf_stage = T == 2 ? 0 : 2, pf_it = T == 2 ? 1 : 0;

  int s = 0;                                 // k-step of step n: n % S
  int cur_it = 0;                            // row-block iteration being accumulated
  uint64_t c_comp = 0, c_wait = 0, c_bar = 0;   // (ABL & 64) per-phase shader cycles of one wave
  for (int n = 0; n < nsteps; ++n) {
    const uint64_t ca = (ABL & 64) ? __builtin_readcyclecounter() : 0;
    const int s1 = s + 1 == S ? 0 : s + 1;
    // the fragment of step n + 1 (its stage landed by the end of step n - 1), split while the MFMAs run
    bf16x8 p3n[3];
    const bool pre = n + 1 < nsteps;
    if (pre) read_split(n + 1, p3n);

    // W slice of step n + 1: its buffer was last read in step n - 1, which every wave has left.  Whatever follows it
    // in this step may stay in flight past the wait below (s_waitcnt vmcnt(N) waits for all but the N youngest vector
    // memory operations — loads, stores and LDS-DMA count together, in issue order: MI355X_MICROARCH.md, cycle constants):
    int allow = 0;
    if (n + 1 < nsteps && !(ABL & 2)) issue_B(s1, (n + 1) & 1);
    if (s == 0 && n > 0) {
      // seam: the previous row block's tiles, then restart the accumulators; the stores drain during this step and
      // the next (a wave whose 32 rows are not all inside M issues fewer than 4 NCB stores: it waits for everything)
      const bool full = ((int64_t)blockIdx.x + (int64_t)cur_it * gridDim.x) * X3_BM + 32 * wave + 32 <= M;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        flush_tile(cur_it, cb);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
      }
      ++cur_it;
      allow = (full && !(ABL & 8)) ? 1 : 0;
    } else if ((n & 1) && pf_it < nloc && !(ABL & 1)) {
      // the P stage two ahead: its buffer held the stage whose last fragment THIS wave read in step n - 1; waited for
      // at the end of the next, even, step
#pragma unroll
      for (int i = 0; i < 4; ++i) issue_A(pf_it, pf_stage, (n >> 1) & 1, i);
      pf_stage = pf_stage + 1 == T ? 0 : pf_stage + 1;
      if (pf_stage == 0) ++pf_it;
      allow = 2;
    }

    {
      // D[i][j] = sum_k W[k][i] P[j][k]: the W fragment is the A operand (see flush_tile)
      int lane = lane_c;                     // opaque copy (see flush_tile)
      asm volatile("" : "+v"(lane));
      const unsigned char* B = Bbuf + (n & 1) * BSTAGE + (lane >> 5) * (d * 16) + (lane & 31) * 16;
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        bf16x8 w3[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          w3[pl] = *reinterpret_cast<const bf16x8*>(B + pl * (2 * d * 16) + c * 512);
        if constexpr (ABL & 128) {           // no MFMAs: keep the operands alive, nothing else
          asm volatile("" ::"v"(w3[0]), "v"(w3[1]), "v"(w3[2]), "v"(p3[0]), "v"(p3[1]), "v"(p3[2]));
        } else {
          mfma6(acc[c], w3, p3);
        }
      }
    }
    if (pre) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) p3[pl] = p3n[pl];
    }

    const uint64_t cc = (ABL & 64) ? __builtin_readcyclecounter() : 0;
    if (allow == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (allow == 1) {
      if constexpr (NCB == 8) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if constexpr (NCB == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t cd = (ABL & 64) ? __builtin_readcyclecounter() : 0;
    if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
    if constexpr ((ABL & 64) != 0) {
      const uint64_t ce = __builtin_readcyclecounter();
      c_comp += cc - ca; c_wait += cd - cc; c_bar += ce - cd;
    }
    s = s1;
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) flush_tile(cur_it, cb);
  if constexpr ((ABL & 64) != 0) {           // shader cycles of workgroup 0 / wave 0, for scripts/dbg/x3_ablate.hip
    if (blockIdx.x == 0 && tid == 0) {
      uint64_t* o64 = reinterpret_cast<uint64_t*>(out);
      o64[0] = __builtin_readcyclecounter() - x3_c0;
      o64[1] = 0; o64[2] = c_comp; o64[3] = c_wait; o64[4] = c_bar; o64[5] = (uint64_t)nsteps;
    }
  }
}

// The same transform with loading and multiplying on DIFFERENT waves (round 3, d = 256).  In the kernel above every wave
// issues its share of the LDS-DMA pieces, and a vector memory instruction holds its wave at issue (~50 cycles per 1 KiB
// piece with every wave issuing) while that wave issues no MFMAs: the MFMAs alone take 4.0 ms, the memory side alone
// 3.9 ms, together 7.0-7.7.  Here a workgroup is 12 waves: waves 0-7 keep the row tiling (wave w: rows 32 w .. + 31, all
// d columns), read fragments, split, multiply and — once per row block — store their tiles; waves 8-11 do nothing but
// request the W slice of the next step and, every other step, the P stage after the current one, wait for them with
// counted vmcnt, and meet the others at the one barrier per 16-k step.  12 waves mean 168 registers: the MFMA waves
// keep ONE fragment of P (the split of step n + 1 is not computed under the MFMAs of step n; the SIMD's other MFMA
// wave covers the gap).  7.68 -> 7.14 ms at 10^7 x 256 x 256, 14.2 -> 13.0 ms at F = 512 (in-process order A/B,
// scripts/bench_dense_x3.py with MP_X3_PC=0 / 1).  Tried on top and dropped: W slices requested TWO steps ahead through a
// three-slot ring, paid for with half-tile output images (158 KB of LDS): 7.65 vs 7.92 ms for the one-role kernel on
// its box — no better; a W slice landing within its step was not the limit.
#ifndef MP_X3PC_ABL
#define MP_X3PC_ABL 0   // ablation bits (timing studies, wrong results): 1 loaders idle after the prologue, 2 no MFMAs, 4 no output
#endif
constexpr int X3PC_LOADERS = 4;
constexpr int X3PC_THREADS = 64 * (X3_WAVES + X3PC_LOADERS);

__global__ __launch_bounds__(X3PC_THREADS, 3) void dense_x3_pc_kernel(const float* __restrict__ P, int64_t ldp,
                                                                      const unsigned char* __restrict__ Ws,
                                                                      const float* __restrict__ bias, int act,
                                                                      float* __restrict__ out, int64_t ldo, int64_t M,
                                                                      int32_t F) {
  constexpr int NCB = 8;
  constexpr int d = 32 * NCB;
  constexpr int BSTAGE = 96 * d;             // [3 planes][2 k-chunks of 8][d columns][8 bf16]
  constexpr int BPIECES = 3 * NCB;           // 1 KiB pieces of one W stage
  constexpr int QP = NCB / 2;                // pieces per (plane, k-chunk)
  // ONE shared array (see dense_x3_kernel)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * X3_ASTAGE + 2 * BSTAGE + X3_WAVES * X3_OSTAGE + 4 * d];
  unsigned char* const Abuf = lds;
  unsigned char* const Bbuf = lds + 2 * X3_ASTAGE;

  const int tid = threadIdx.x, lane_c = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = F >> 4;                      // 16-k steps per row block (even: F % 32 == 0)
  const int T = F >> 5;                      // 32-k stages of P per row block (>= 2)
  const int64_t nrb = (M + X3_BM - 1) / X3_BM;
  if ((int64_t)blockIdx.x >= nrb) return;
  const int nloc = (int)((nrb - blockIdx.x + gridDim.x - 1) / gridDim.x);   // row blocks of this workgroup
  const int nsteps = nloc * S;
  const int nstages = nloc * T;
  const uint32_t ldp_b = (uint32_t)ldp * 4u, ldo_b = (uint32_t)ldo * 4u;

  if (wave >= X3_WAVES) {
    // ------------------------------------------------ loaders ------------------------------------------------
    const int lw = wave - X3_WAVES;
    // piece i (8 rows x 128 B) of the 32 rows of MFMA wave rw, P stage `stage` of row-block iteration `it` (layout and
    // swizzle: dense_x3_kernel's issue_A)
    auto issue_A = [&](int it, int stage, int buf, int rw, int i) {
      int lane = lane_c;
      asm volatile("" : "+v"(lane));
      const int p_row = lane >> 3;
      const int64_t r0 = ((int64_t)blockIdx.x + (int64_t)it * gridDim.x) * X3_BM;
      const int64_t left = M - r0;
      const int row = 32 * rw + 8 * i + p_row;
      const int gch = (lane & 7) ^ ((row >> 1) & 7);
      const int srow = row < left ? row : (int)left - 1;
      const unsigned char* base = reinterpret_cast<const unsigned char*>(P) + r0 * (int64_t)ldp_b;
      glds16(base + ((uint32_t)srow * ldp_b + (uint32_t)(128 * stage + 16 * gch)),
             Abuf + buf * X3_ASTAGE + (32 * rw + 8 * i) * 128);
    };
    auto issue_stage = [&](int m) {          // global stage m = (row-block iteration m / T, stage m % T) -> slot m & 1
      const int it = m / T, st = m - it * T;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_A(it, st, m & 1, 2 * lw + r, i);
    };
    auto issue_B = [&](int s, int buf) {     // W slice s (k = 16 s .. + 15): loader lw moves pieces lw, lw + 4, ...
      int lane = lane_c;
      asm volatile("" : "+v"(lane));
#pragma unroll
      for (int j = 0; j < BPIECES / X3PC_LOADERS; ++j) {
        const int p = lw + X3PC_LOADERS * j;
        const int ph = p / QP, q = p % QP;
        const size_t src = (size_t)(ph >> 1) * ((size_t)F * d * 2) + (size_t)(2 * s + (ph & 1)) * (d * 16) + q * 1024;
        glds16(Ws + src + lane * 16, Bbuf + buf * BSTAGE + p * 1024);
      }
    };
    issue_B(0, 0);
    issue_stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int s = 0;
    for (int n = 0; n < nsteps; ++n) {
      const int s1 = s + 1 == S ? 0 : s + 1;
      // W slice of step n + 1: its buffer was last read in step n - 1, which every wave has left.  P stage (n >> 1) + 1
      // at even n: its slot held the stage whose second half was read in step n - 1; it is first read in step n + 2.
      const bool more_w = n + 1 < nsteps && !(MP_X3PC_ABL & 1);
      const bool more_p = !(n & 1) && (n >> 1) + 1 < nstages && !(MP_X3PC_ABL & 1);
      if (more_w) issue_B(s1, (n + 1) & 1);
      if (more_p) {
        issue_stage((n >> 1) + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // the W slice has landed; the 8 pieces of P stay in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      s = s1;
    }
    return;
  }

  // ------------------------------------------------ MFMA waves ------------------------------------------------
  auto read_split = [&](int n, bf16x8 (&p3)[3]) {
    int lane = lane_c;
    asm volatile("" : "+v"(lane));
    const int f_row = 32 * wave + (lane & 31), f_h = lane >> 5, f_sw = (f_row >> 1) & 7;
    const unsigned char* A = Abuf + ((n >> 1) & 1) * X3_ASTAGE + f_row * 128;
    const int g0 = 4 * (n & 1) + 2 * f_h;
    const x3_f32x4 x0 = *reinterpret_cast<const x3_f32x4*>(A + 16 * (g0 ^ f_sw));
    const x3_f32x4 x1 = *reinterpret_cast<const x3_f32x4*>(A + 16 * ((g0 + 1) ^ f_sw));
    const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    split3_bf16(xv, p3[0], p3[1], p3[2]);
  };
  f32x16 acc[NCB];
  unsigned char* const Obuf = lds + 2 * X3_ASTAGE + 2 * BSTAGE + wave * X3_OSTAGE;
  float* const bias_l = reinterpret_cast<float*>(lds + 2 * X3_ASTAGE + 2 * BSTAGE + X3_WAVES * X3_OSTAGE);
  if (tid < d) bias_l[tid] = bias != nullptr ? bias[tid] : 0.f;
  auto flush_tile = [&](int it, int cb) {    // (dense_x3_kernel's epilogue: a 32 x 32 tile through the wave's LDS image)
    if ((MP_X3PC_ABL & 4) && act != 77) return;
    int lane = lane_c;
    asm volatile("" : "+v"(lane));
    const int f_h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const x3_f32x4 b4 = *reinterpret_cast<const x3_f32x4*>(bias_l + cb * 32 + 8 * q + 4 * f_h);
      x3_f32x4 v = {acc[cb][4 * q] + b4[0], acc[cb][4 * q + 1] + b4[1], acc[cb][4 * q + 2] + b4[2],
                    acc[cb][4 * q + 3] + b4[3]};
      if (act == MP_ACT_RELU) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      *reinterpret_cast<x3_f32x4*>(Obuf + (lane & 31) * X3_OROW + (8 * q + 4 * f_h) * 4) = v;
    }
    const int64_t r0 = ((int64_t)blockIdx.x + (int64_t)it * gridDim.x) * X3_BM + 32 * wave;
    const int64_t left = M - r0;
    const int nrow = left >= 32 ? 32 : (left > 0 ? (int)left : 0);
    unsigned char* const base = reinterpret_cast<unsigned char*>(out) + r0 * (int64_t)ldo_b + cb * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const x3_f32x4 v = *reinterpret_cast<const x3_f32x4*>(Obuf + (8 * i + (lane >> 3)) * X3_OROW + 16 * (lane & 7));
      if (8 * i + (lane >> 3) < nrow)
        *reinterpret_cast<x3_f32x4*>(base + ((uint32_t)(8 * i + (lane >> 3)) * ldo_b + (uint32_t)(16 * (lane & 7)))) = v;
    }
  };
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // bias_l is written
  __builtin_amdgcn_s_barrier();              // W slice 0 and P stage 0 have landed
  int s = 0, cur_it = 0;
  for (int n = 0; n < nsteps; ++n) {
    if (s == 0 && n > 0) {                   // seam: the previous row block's tiles, then restart the accumulators
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        flush_tile(cur_it, cb);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
      }
      ++cur_it;
    }
    bf16x8 p3[3];
    read_split(n, p3);
    {
      int lane = lane_c;
      asm volatile("" : "+v"(lane));
      const unsigned char* B = Bbuf + (n & 1) * BSTAGE + (lane >> 5) * (d * 16) + (lane & 31) * 16;
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        bf16x8 w3[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          w3[pl] = *reinterpret_cast<const bf16x8*>(B + pl * (2 * d * 16) + c * 512);
#if MP_X3PC_ABL & 2
        asm volatile("" ::"v"(w3[0]), "v"(w3[1]), "v"(w3[2]), "v"(p3[0]), "v"(p3[1]), "v"(p3[2]));
#else
        mfma6(acc[c], w3, p3);
#endif
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this step's LDS reads are done before the loaders overwrite
    __builtin_amdgcn_s_barrier();
    s = s + 1 == S ? 0 : s + 1;
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) flush_tile(cur_it, cb);
}

// W [K rows, n columns] (trans == 0: B[k][c] = W[k][c], K = F, n = d) or its transpose (trans != 0: B[k][c] = W[c][k],
// W [n rows, K columns]) -> [3][K / 8][n][8] bf16, plane s = bf16(B - sum of the planes before it)
__global__ __launch_bounds__(kBlock) void split_w_kernel(const float* __restrict__ W, int64_t ldw, int32_t K,
                                                         int32_t n, int trans, unsigned char* __restrict__ out) {
  const int64_t total = (int64_t)(K / 8) * n;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int kc = (int)(i / n), c = (int)(i % n);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = trans ? W[(int64_t)c * ldw + 8 * kc + j] : W[(int64_t)(8 * kc + j) * ldw + c];
    bf16x8 s3[3];
    split3_bf16(x, s3[0], s3[1], s3[2]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<bf16x8*>(out + ((size_t)pl * (K / 8) * n + (size_t)i) * 16) = s3[pl];
  }
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_split_w_bf16x3(const float* W, int64_t ldw, int32_t K, int32_t n, int32_t trans, void* W_split,
                      mp_stream_t stream) {
  if (!W || !W_split || K < 8 || n < 1 || K % 8) return MP_ERR_INVALID_ARG;
  if (ldw < (trans ? K : n)) return MP_ERR_INVALID_ARG;
  if ((uintptr_t)W_split % 16) return MP_ERR_ALIGNMENT;
  hipLaunchKernelGGL(split_w_kernel, dim3(flat_grid((int64_t)(K / 8) * n)), dim3(kBlock), 0, as_stream(stream), W, ldw,
                     K, n, (int)trans, reinterpret_cast<unsigned char*>(W_split));
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_dense_x3_f32(const float* P, int64_t ldp, const void* W_split, const float* bias, int32_t act, float* out,
                    int64_t ldo, int64_t M, int32_t F, int32_t d, mp_stream_t stream) {
  if (!P || !W_split || !out || M < 0 || F < 1 || d < 1 || ldp < F || ldo < d) return MP_ERR_INVALID_ARG;
  if (act != MP_ACT_NONE && act != MP_ACT_RELU) return MP_ERR_INVALID_ARG;
  if (F % 32 || F < 64 || (d != 64 && d != 128 && d != 256)) return MP_ERR_UNSUPPORTED;
  if (ldp > (1 << 20) || ldo > (1 << 20)) return MP_ERR_UNSUPPORTED;   // 32-bit lane offsets inside a 256-row block
  if (ldp % 4 || ldo % 4 || ((uintptr_t)P % 16) || ((uintptr_t)out % 16) || ((uintptr_t)W_split % 16) ||
      ((uintptr_t)bias % 4))
    return MP_ERR_ALIGNMENT;
  if (M == 0) return MP_OK;
  const int64_t nrb = ceil_div(M, X3_BM);
  const dim3 grid((unsigned)(nrb < kNumCU ? nrb : kNumCU)), block(X3_THREADS);
  const unsigned char* Ws = reinterpret_cast<const unsigned char*>(W_split);
  hipStream_t st = as_stream(stream);
  switch (d) {
    case 64: hipLaunchKernelGGL(dense_x3_kernel<2>, grid, block, 0, st, P, ldp, Ws, bias, (int)act, out, ldo, M, F); break;
    case 128: hipLaunchKernelGGL(dense_x3_kernel<4>, grid, block, 0, st, P, ldp, Ws, bias, (int)act, out, ldo, M, F); break;
    default: {
      const char* e = getenv("MP_X3_PC");                    // MP_X3_PC=0: the one-role kernel (A/B studies), read per call
      if (e && e[0] == '0') hipLaunchKernelGGL(dense_x3_kernel<8>, grid, block, 0, st, P, ldp, Ws, bias, (int)act, out, ldo, M, F);
      else hipLaunchKernelGGL(dense_x3_pc_kernel, grid, dim3(X3PC_THREADS), 0, st, P, ldp, Ws, bias, (int)act, out, ldo, M, F);
      break;
    }
  }
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // extern "C"

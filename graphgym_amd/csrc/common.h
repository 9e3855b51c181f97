// Shared helpers for libmpengine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/mp_engine.h"

namespace mp {

constexpr int kWave = 64;    // CDNA wavefront
constexpr int kBlock = 256;  // 4 waves per workgroup everywhere in this library
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kNumCU = 256;  // MI355X

void set_hip_error(hipError_t e, const char* what);

#define MP_HIP(call)                                   \
  do {                                                 \
    hipError_t _e = (call);                            \
    if (_e != hipSuccess) {                            \
      ::mp::set_hip_error(_e, #call);                  \
      return MP_ERR_HIP;                               \
    }                                                  \
  } while (0)

#define MP_LAUNCH_CHECK()                              \
  do {                                                 \
    hipError_t _e = hipGetLastError();                 \
    if (_e != hipSuccess) {                            \
      ::mp::set_hip_error(_e, "kernel launch");        \
      return MP_ERR_HIP;                               \
    }                                                  \
  } while (0)

// Workgroup barrier of the ROLE-SPLIT kernels (fused.hip agg_dense_pc_kernel, gemm.hip dense_wgrad_pc_kernel,
// dense_x3.hip dense_x3_pc_kernel): the waves of a workgroup are divided into roles by whole waves (wave-uniform), each
// role runs its OWN loop over the same item sequence, and the loops meet at barriers that pair up BY COUNT — gfx9's
// s_barrier counts arriving waves wherever they are in the code.  HIP's __syncthreads() promises a barrier only where
// every thread of the block reaches the SAME call, so these sites do not use it: they state the hardware instruction
// and the fences it needs themselves —
//   release fence (workgroup): this wave's LDS writes (and its prior global writes, as far as the workgroup sees them)
//                              are complete before it arrives: s_waitcnt lgkmcnt(0) on gfx950, no vmcnt wait, so
//                              global LOADS in flight stay in flight across the barrier (the kernels rely on that);
//   s_barrier;
//   acquire fence (workgroup): nothing after the barrier is hoisted above it.
// tests/test_emitted_barriers.py disassembles the built code objects and checks, per instantiation, that each role loop
// holds exactly the barriers its source holds (no merge, hoist or duplication by the compiler).
__device__ __forceinline__ void role_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

static inline hipStream_t as_stream(mp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// grid for a flat elementwise / grid-stride kernel: enough blocks to fill the
// chip (256 CUs x 8 blocks) and no more
static inline int flat_grid(int64_t n) {
  int64_t b = ceil_div(n, kBlock);
  if (b < 1) b = 1;
  if (b > kNumCU * 8) b = kNumCU * 8;
  return (int)b;
}

// ---- plan blob layout (int32 words) -------------------------------------
constexpr int32_t kPlanMagic = 0x4D50504C;
enum PlanWord {
  PW_MAGIC = 0, PW_NSEG, PW_SEG_COST, PW_ROW_COST, PW_HUB_DEG, PW_PIECE_EDGES,
  PW_NHUB, PW_NPIECE, PW_CAP_HUB, PW_CAP_PIECE, PW_HEADER_WORDS = 16
};

struct PlanCfg {
  int seg_cost;     // cost budget of one segment (1 per stored entry + row_cost per row)
  int row_cost;     // cost of one row (its flush: a 1 KiB store and bookkeeping)
  int hub_deg;      // rows longer than this are split into pieces
  int piece_edges;  // entries per hub piece
};

struct PlanView {   // host-side view of the device blob
  int32_t n_seg, cap_hub, cap_piece;
  const int32_t* seg_row;    // [n_seg+1]
  const int32_t* hub_row;    // [cap_hub]
  const int32_t* hub_base;   // [cap_hub]   first piece of hub h
  const int32_t* hub_np;     // [cap_hub]   pieces of hub h
  const int32_t* piece_hub;  // [cap_piece]
  const int32_t* piece_k;    // [cap_piece]
  const int32_t* header;
};

}  // namespace mp

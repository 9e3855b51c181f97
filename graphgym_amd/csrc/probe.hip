// The placement probe: a timed streaming copy between two device ranges.
//
// Background (profiles/r03_placement_retry.log): on MI355X an HBM-bound launch that reads matrix X and
// writes matrix Y runs up to ~13 % slower depending on which physical memory the two are backed by — the high address
// bits are hashed into the DRAM bank / channel selection, and a read stream and a write stream that hash alike pay bus
// turnarounds.  A 512 MiB copy between sample chunks of X and of a candidate Y shows it (+0-2 % for the positions where
// the aggregation runs at 20.8 ms, +4 % where it runs at 23.3 ms), so graphgym_amd/placement.py checks every large
// output it allocates (through torch) with this probe and re-allocates on conflict.  The library itself allocates
// nothing.  No counterpart in the reference (it never places anything).
#include "common.h"

namespace mp {

typedef float ap_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBlock) void probe_copy_kernel(const ap_f32x4* __restrict__ src,
                                                            ap_f32x4* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(src[i], dst + i);
}

// The same question asked the way the hot kernel asks it: every 1 KiB row written to dst is the sum of `fan` rows of
// src picked pseudo-randomly from ALL of src (an aggregation reads its whole input while it writes each part of its
// output, ~10 rows read per row written), a wave per row, all `fan` row loads issued before the first is consumed.
// Against the copy probe this one shows the placement effect at the size the aggregation feels it (+12-13 % between
// a good and a bad position of the output; the 1:1 copy shows +4 %).
__global__ __launch_bounds__(kBlock) void probe_gather_kernel(const ap_f32x4* __restrict__ src, uint64_t src_rows,
                                                              ap_f32x4* __restrict__ dst, int64_t dst_rows, int fan,
                                                              uint32_t seed) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / kWave;
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t r = wave; r < dst_rows; r += n_waves) {
    ap_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    uint64_t h = ((uint64_t)r + 1) * 0x9E3779B97F4A7C15ull + seed;
    for (int k0 = 0; k0 < fan; k0 += 8) {
      ap_f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;          // one splitmix round per row picked
        const uint64_t row = (k0 + j < fan) ? h % src_rows : 0;
        v[j] = (k0 + j < fan) ? src[row * kWave + lane] : ap_f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    }
    __builtin_nontemporal_store(acc, dst + r * kWave + lane);
  }
}

}  // namespace mp

using namespace mp;

extern "C" {

// Timed streaming copy src -> dst of `bytes` (multiple of 16, 16-byte aligned): `reps` launches between two
// events on `stream`, one untimed launch first.  SYNCHRONISES.  *ms_host = mean per launch.  The yardstick
// placement.py uses to find out which parts of memory conflict.
int mp_probe_copy_ms(const void* src, void* dst, size_t bytes, int32_t reps, float* ms_host, mp_stream_t stream) {
  if (!src || !dst || !ms_host || reps < 1 || bytes % 16 || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16))
    return MP_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  hipEvent_t e0, e1;
  MP_HIP(hipEventCreate(&e0));
  MP_HIP(hipEventCreate(&e1));
  const int64_t n4 = (int64_t)(bytes / 16);
  auto launch = [&]() {
    hipLaunchKernelGGL(probe_copy_kernel, dim3(kNumCU * 8), dim3(kBlock), 0, st,
                       reinterpret_cast<const ap_f32x4*>(src), reinterpret_cast<ap_f32x4*>(dst), n4);
  };
  launch();
  MP_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  MP_HIP(hipEventRecord(e1, st));
  MP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MP_HIP(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  MP_LAUNCH_CHECK();
  *ms_host = ms / (float)reps;
  return MP_OK;
}

// Timed gather probe: dst_bytes / 1024 rows of dst, each the sum of `fan` pseudo-random 1 KiB rows of
// [src, src + src_bytes); one untimed launch, then `reps` timed ones; SYNCHRONISES; *ms_host = mean per launch.
int mp_probe_gather_ms(const void* src, size_t src_bytes, void* dst, size_t dst_bytes, int32_t fan, int32_t reps,
                       float* ms_host, mp_stream_t stream) {
  if (!src || !dst || !ms_host || reps < 1 || fan < 1 || fan > 64 || src_bytes < 1024 || dst_bytes < 1024 ||
      ((uintptr_t)src % 16) || ((uintptr_t)dst % 16))
    return MP_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  hipEvent_t e0, e1;
  MP_HIP(hipEventCreate(&e0));
  MP_HIP(hipEventCreate(&e1));
  const uint64_t src_rows = src_bytes / 1024;
  const int64_t dst_rows = (int64_t)(dst_bytes / 1024);
  uint32_t seed = 1;
  auto launch = [&]() {
    hipLaunchKernelGGL(probe_gather_kernel, dim3(kNumCU * 8), dim3(kBlock), 0, st,
                       reinterpret_cast<const ap_f32x4*>(src), src_rows, reinterpret_cast<ap_f32x4*>(dst), dst_rows,
                       (int)fan, seed++);
  };
  launch();
  MP_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  MP_HIP(hipEventRecord(e1, st));
  MP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MP_HIP(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  MP_LAUNCH_CHECK();
  *ms_host = ms / (float)reps;
  return MP_OK;
}

}  // extern "C"

// Per-lane vector load/store helpers and wave broadcasts shared by the aggregation and
// attention kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int W> __device__ __forceinline__ void load_vec(const float* p, float (&v)[W]);
template <> __device__ __forceinline__ void load_vec<4>(const float* p, float (&v)[4]) {
  f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load_vec<2>(const float* p, float (&v)[2]) {
  f32x2 t = *reinterpret_cast<const f32x2*>(p);
  v[0] = t[0]; v[1] = t[1];
}
template <> __device__ __forceinline__ void load_vec<1>(const float* p, float (&v)[1]) { v[0] = *p; }

// streaming (non-temporal) load of a row that this kernel reads once: it should not push resident data (the transform's
// weights) out of the L2
template <int W> __device__ __forceinline__ void load_vec_nt(const float* p, float (&v)[W]) {
  if constexpr (W == 4) {
    f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else if constexpr (W == 2) {
    f32x2 t = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p));
    v[0] = t[0]; v[1] = t[1];
  } else {
    v[0] = __builtin_nontemporal_load(p);
  }
}

template <int W> __device__ __forceinline__ void store_vec(float* p, const float (&v)[W]);
template <> __device__ __forceinline__ void store_vec<4>(float* p, const float (&v)[4]) {
  f32x4 t = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = t;
}
template <> __device__ __forceinline__ void store_vec<2>(float* p, const float (&v)[2]) {
  f32x2 t = {v[0], v[1]};
  *reinterpret_cast<f32x2*>(p) = t;
}
template <> __device__ __forceinline__ void store_vec<1>(float* p, const float (&v)[1]) { *p = v[0]; }

// streaming (non-temporal) store of an output row: written once, never re-read by this kernel
template <int W> __device__ __forceinline__ void store_vec_nt(float* p, const float (&v)[W]) {
  if constexpr (W == 4) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
  } else if constexpr (W == 2) {
    f32x2 t = {v[0], v[1]};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x2*>(p));
  } else {
    __builtin_nontemporal_store(v[0], p);
  }
}

template <int W> __device__ __forceinline__ void store_ivec(int32_t* p, const int (&v)[W]);
template <> __device__ __forceinline__ void store_ivec<4>(int32_t* p, const int (&v)[4]) {
  i32x4 t = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<i32x4*>(p) = t;
}
template <> __device__ __forceinline__ void store_ivec<2>(int32_t* p, const int (&v)[2]) {
  i32x2 t = {v[0], v[1]};
  *reinterpret_cast<i32x2*>(p) = t;
}
template <> __device__ __forceinline__ void store_ivec<1>(int32_t* p, const int (&v)[1]) { *p = v[0]; }

template <int W> __device__ __forceinline__ void load_ivec(const int32_t* p, int (&v)[W]);
template <> __device__ __forceinline__ void load_ivec<4>(const int32_t* p, int (&v)[4]) {
  i32x4 t = *reinterpret_cast<const i32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load_ivec<2>(const int32_t* p, int (&v)[2]) {
  i32x2 t = *reinterpret_cast<const i32x2*>(p);
  v[0] = t[0]; v[1] = t[1];
}
template <> __device__ __forceinline__ void load_ivec<1>(const int32_t* p, int (&v)[1]) { v[0] = *p; }

__device__ __forceinline__ int bcast_i(int v, int lane) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane));
}
__device__ __forceinline__ float bcast_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v),
                                                             __builtin_amdgcn_readfirstlane(lane)));
}


}  // namespace mp

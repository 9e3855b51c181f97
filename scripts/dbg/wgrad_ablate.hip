// Where does the weight-gradient kernel's time go?  dense_wgrad_wide_kernel<false> at 10^7 x 256 x 256 with parts
// switched off (wrong results, timing only).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Igraphgym_amd/csrc -Iinclude scripts/dbg/wgrad_ablate.hip \
//         graphgym_amd/csrc/util.hip -o /tmp/wgrad_ablate && /tmp/wgrad_ablate
#include "../../graphgym_amd/csrc/gemm.hip"
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int ABL>
static float run(const float* P, const float* G, float* ws, int64_t M, int F, int d, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int64_t chunk = 16384, n_chunk = (M + chunk - 1) / chunk;
  const dim3 grid((unsigned)(2 * n_chunk)), block(256);
  auto go = [&]() {
    hipLaunchKernelGGL((mp::dense_wgrad_wide_kernel<false, ABL>), grid, block, 0, 0, P, (int64_t)F, G, (int64_t)d,
                       (const float*)nullptr, (int64_t)0, (float*)nullptr, (int64_t)0, M, F, d, chunk, ws, (float*)nullptr);
  };
  go();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) go();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

__global__ void fill_random(float* p, int64_t n, unsigned seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xffffff) / 8388608.0f - 1.0f;
  }
}

int main() {
  const int64_t M = 10000000; const int F = 256, d = 256;
  float *P, *G, *ws;
  CK(hipMalloc(&P, M * F * 4)); CK(hipMalloc(&G, M * d * 4)); CK(hipMalloc(&ws, (size_t)700 * F * d * 4));
  CK(hipMemset(P, 0x3c, M * F * 4)); CK(hipMemset(G, 0x3d, M * d * 4));
  CK(hipDeviceSynchronize());
  printf("full                         %7.3f ms\n", run<0>(P, G, ws, M, F, d, 5));
  printf("no global loads              %7.3f ms\n", run<1>(P, G, ws, M, F, d, 5));
  printf("no split / LDS stores        %7.3f ms\n", run<2>(P, G, ws, M, F, d, 5));
  printf("no loads, no split/stores    %7.3f ms\n", run<3>(P, G, ws, M, F, d, 5));
  printf("no MFMAs                     %7.3f ms\n", run<4>(P, G, ws, M, F, d, 5));
  printf("no MFMAs, no fragment reads  %7.3f ms\n", run<12>(P, G, ws, M, F, d, 5));
  printf("no fragment reads            %7.3f ms\n", run<8>(P, G, ws, M, F, d, 5));
  printf("MFMAs + fragment reads only  %7.3f ms\n", run<3>(P, G, ws, M, F, d, 5));
  printf("MFMAs only                   %7.3f ms\n", run<11>(P, G, ws, M, F, d, 5));
  printf("loads only                   %7.3f ms\n", run<14>(P, G, ws, M, F, d, 5));
  printf("loads + split/stores         %7.3f ms\n", run<12>(P, G, ws, M, F, d, 5));
  printf("full again                   %7.3f ms\n", run<0>(P, G, ws, M, F, d, 5));
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, P, M * F, 1u);
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, G, M * d, 7u);
  CK(hipDeviceSynchronize());
  printf("-- uniform random operands in [-1, 1)\n");
  printf("full                         %7.3f ms\n", run<0>(P, G, ws, M, F, d, 5));
  printf("no global loads              %7.3f ms\n", run<1>(P, G, ws, M, F, d, 5));
  printf("MFMAs + fragment reads only  %7.3f ms\n", run<3>(P, G, ws, M, F, d, 5));
  printf("loads + split/stores         %7.3f ms\n", run<12>(P, G, ws, M, F, d, 5));
  return 0;
}

// Where does the streaming transform's time go?  Times dense_x3_kernel<8> at 10^7 x 256 x 256 with parts switched off
// (wrong results, timing only).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igraphgym_amd/csrc -Iinclude scripts/dbg/x3_ablate.hip \
//         graphgym_amd/csrc/util.hip -o /tmp/x3_ablate && /tmp/x3_ablate
#include "../../graphgym_amd/csrc/dense_x3.hip"
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int ABL>
static float run(const float* P, const unsigned char* Ws, const float* bias, float* out, int64_t M, int F, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int64_t nrb = (M + mp::X3_BM - 1) / mp::X3_BM;
  const dim3 grid((unsigned)(nrb < 256 ? nrb : 256)), block(mp::X3_THREADS);
  hipLaunchKernelGGL((mp::dense_x3_kernel<8, ABL>), grid, block, 0, 0, P, (int64_t)F, Ws, bias, 1, out, (int64_t)256, M, F);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((mp::dense_x3_kernel<8, ABL>), grid, block, 0, 0, P, (int64_t)F, Ws, bias, 1, out, (int64_t)256, M, F);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  if (ABL & 64) {
    unsigned long long c[6] = {0};
    hipMemcpy(c, out, 48, hipMemcpyDeviceToHost);
    printf("   [workgroup 0, wave 0: %llu cycles (%.0f MHz if it ran all %.3f ms); per step of %llu: (unused %.0f) "
           "issue/flush/read/split/MFMA %.0f, vmcnt wait %.0f, barrier %.0f]\n   ", c[0], c[0] / (ms / reps) / 1e3, ms / reps, c[5],
           (double)c[1] / c[5], (double)c[2] / c[5], (double)c[3] / c[5], (double)c[4] / c[5]);
  }
  return ms / reps;
}

int main() {
  const int64_t M = 10000000; const int F = 256, d = 256;
  float *P, *W, *bias, *out; unsigned char* Ws;
  CK(hipMalloc(&P, M * F * 4)); CK(hipMalloc(&out, M * d * 4)); CK(hipMalloc(&W, F * d * 4));
  CK(hipMalloc(&bias, d * 4)); CK(hipMalloc(&Ws, (size_t)6 * F * d));
  std::vector<float> h((size_t)F * d);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 8000.f - 0.06f;
  CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, d * 4));
  // P: any finite pattern will do (MFMA time does not depend on the values, HBM time does not either)
  CK(hipMemset(P, 0x3c, M * F * 4));
  if (mp_split_w_bf16x3(W, d, F, d, 0, Ws, nullptr) != 0) return 2;
  CK(hipDeviceSynchronize());
  printf("full                         %7.3f ms\n", run<0>(P, Ws, bias, out, M, F, 5));
  printf("no P loads                   %7.3f ms\n", run<1>(P, Ws, bias, out, M, F, 5));
  printf("no W loads                   %7.3f ms\n", run<2>(P, Ws, bias, out, M, F, 5));
  printf("no P, no W loads             %7.3f ms\n", run<3>(P, Ws, bias, out, M, F, 5));
  printf("no split                     %7.3f ms\n", run<4>(P, Ws, bias, out, M, F, 5));
  printf("no stores                    %7.3f ms\n", run<8>(P, Ws, bias, out, M, F, 5));
  printf("no barrier                   %7.3f ms\n", run<16>(P, Ws, bias, out, M, F, 5));
  printf("no loads, no stores          %7.3f ms\n", run<11>(P, Ws, bias, out, M, F, 5));
  printf("no loads/stores/split        %7.3f ms\n", run<15>(P, Ws, bias, out, M, F, 5));
  printf("no loads/stores/split/barrier%7.3f ms\n", run<31>(P, Ws, bias, out, M, F, 5));
  printf("full + cycle count           %7.3f ms\n", run<64>(P, Ws, bias, out, M, F, 5));
  printf("no loads/stores + cycles     %7.3f ms\n", run<64 + 11>(P, Ws, bias, out, M, F, 5));
  printf("no loads/stores/split + cyc  %7.3f ms\n", run<64 + 15>(P, Ws, bias, out, M, F, 5));
  printf("no stores + cycles           %7.3f ms\n", run<64 + 8>(P, Ws, bias, out, M, F, 5));
  printf("no MFMAs (memory side only)  %7.3f ms\n", run<128>(P, Ws, bias, out, M, F, 5));
  printf("no MFMAs, no stores          %7.3f ms\n", run<128 + 8>(P, Ws, bias, out, M, F, 5));
  printf("no MFMAs, no loads           %7.3f ms\n", run<128 + 3>(P, Ws, bias, out, M, F, 5));
  printf("no MFMAs, no W loads         %7.3f ms\n", run<128 + 2>(P, Ws, bias, out, M, F, 5));
  printf("no MFMAs, no split           %7.3f ms\n", run<128 + 4>(P, Ws, bias, out, M, F, 5));
  printf("full again                   %7.3f ms\n", run<0>(P, Ws, bias, out, M, F, 5));
  return 0;
}

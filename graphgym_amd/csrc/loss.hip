// Softmax cross-entropy over the labelled rows, forward and backward as one HBM pass each.
// Replaces, in the training step that drives the path (graphgym/loss.py:53-68: mean softmax-CE over
// node_label_index rows; graphgym/loss.py:20-37 for the torch path), torch's nll_loss kernels, which take
// 19 ms per step on a [10^7, 7] logit matrix (profiles/r01_gcn_step_kernels.csv) against 0.4 GB of traffic.
// One lane per labelled row: C is small (7-10 classes), a wave covers 64 consecutive rows.
#include "common.h"

namespace mp {

__global__ __launch_bounds__(kBlock) void softmax_ce_rows_kernel(const float* __restrict__ logits, int64_t ld,
                                                                 const int64_t* __restrict__ labels,
                                                                 const int64_t* __restrict__ index, int64_t n_sel,
                                                                 int32_t C, float* __restrict__ row_loss) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_sel; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = index ? index[k] : k;
    const float* z = logits + i * ld;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, z[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[c] - m);
    const int64_t y = labels[k];
    row_loss[k] = logf(s) + m - z[y];
  }
}

// dlogits[i, :] = (softmax(z_i) - onehot(y)) * gscale[0] * inv_n for the labelled rows (other rows untouched: the
// caller zeroes dlogits when index selects a subset)
__global__ __launch_bounds__(kBlock) void softmax_ce_bwd_kernel(const float* __restrict__ logits, int64_t ld,
                                                                const int64_t* __restrict__ labels,
                                                                const int64_t* __restrict__ index, int64_t n_sel,
                                                                int32_t C, const float* __restrict__ gscale, float inv_n,
                                                                float* __restrict__ dlogits, int64_t ldd) {
  const float coef = gscale[0] * inv_n;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_sel; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = index ? index[k] : k;
    const float* z = logits + i * ld;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, z[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[c] - m);
    const float inv_s = 1.0f / s;
    const int64_t y = labels[k];
    float* dz = dlogits + i * ldd;
    for (int c = 0; c < C; ++c) dz[c] = (expf(z[c] - m) * inv_s - (c == y ? 1.f : 0.f)) * coef;
  }
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_softmax_ce_rows_f32(const float* logits, int64_t ld, const int64_t* labels, const int64_t* index,
                           int64_t n_sel, int32_t C, float* row_loss, mp_stream_t stream) {
  if (n_sel < 0 || C <= 0 || ld < C || (n_sel > 0 && (!logits || !labels || !row_loss))) return MP_ERR_INVALID_ARG;
  if (n_sel == 0) return MP_OK;
  hipLaunchKernelGGL(softmax_ce_rows_kernel, dim3(flat_grid(n_sel)), dim3(kBlock), 0, as_stream(stream), logits, ld,
                     labels, index, n_sel, C, row_loss);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_softmax_ce_bwd_f32(const float* logits, int64_t ld, const int64_t* labels, const int64_t* index,
                          int64_t n_sel, int32_t C, const float* gscale, float inv_n, float* dlogits, int64_t ldd,
                          mp_stream_t stream) {
  if (n_sel < 0 || C <= 0 || ld < C || ldd < C || !gscale || (n_sel > 0 && (!logits || !labels || !dlogits)))
    return MP_ERR_INVALID_ARG;
  if (n_sel == 0) return MP_OK;
  hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3(flat_grid(n_sel)), dim3(kBlock), 0, as_stream(stream), logits, ld,
                     labels, index, n_sel, C, gscale, inv_n, dlogits, ldd);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // extern "C"

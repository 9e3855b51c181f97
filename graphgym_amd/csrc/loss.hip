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
                                                                 int64_t n_rows, int32_t C,
                                                                 float* __restrict__ row_loss) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_sel; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = index ? index[k] : k;
    const int64_t y = labels[k];
    // a label outside [0, C) (e.g. torch's ignore_index -100, which this op does not implement) or a row outside the
    // logit matrix: no read through it; the row's loss is NaN, so the mean is NaN and the mistake cannot pass silently
    if ((uint64_t)y >= (uint64_t)C || (uint64_t)i >= (uint64_t)n_rows) { row_loss[k] = NAN; continue; }
    const float* z = logits + i * ld;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, z[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[c] - m);
    row_loss[k] = logf(s) + m - z[y];
  }
}

// dlogits[i, :] (+)= (softmax(z_i) - onehot(y)) * gscale[0] * inv_n for the labelled rows.  With an index the caller
// zeroes dlogits first and the rows ACCUMULATE (hardware float atomics): a row listed twice gets both terms, as
// F.cross_entropy(logits[index]) gives it; a row listed once receives one add onto zero, so the result stays bitwise
// reproducible whenever the index is unique.  Rows with an invalid label / index get no gradient (the forward's loss
// is NaN for them).
template <bool ACCUM>
__global__ __launch_bounds__(kBlock) void softmax_ce_bwd_kernel(const float* __restrict__ logits, int64_t ld,
                                                                const int64_t* __restrict__ labels,
                                                                const int64_t* __restrict__ index, int64_t n_sel,
                                                                int64_t n_rows, int32_t C,
                                                                const float* __restrict__ gscale, float inv_n,
                                                                float* __restrict__ dlogits, int64_t ldd) {
  const float coef = gscale[0] * inv_n;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_sel; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = index ? index[k] : k;
    const int64_t y = labels[k];
    if ((uint64_t)y >= (uint64_t)C || (uint64_t)i >= (uint64_t)n_rows) continue;
    const float* z = logits + i * ld;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, z[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[c] - m);
    const float inv_s = 1.0f / s;
    float* dz = dlogits + i * ldd;
    for (int c = 0; c < C; ++c) {
      const float v = (expf(z[c] - m) * inv_s - (c == y ? 1.f : 0.f)) * coef;
      if (ACCUM) unsafeAtomicAdd(dz + c, v); else dz[c] = v;
    }
  }
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_softmax_ce_rows_f32(const float* logits, int64_t ld, int64_t n_rows, const int64_t* labels,
                           const int64_t* index, int64_t n_sel, int32_t C, float* row_loss, mp_stream_t stream) {
  if (n_sel < 0 || n_rows < 0 || C <= 0 || ld < C || (n_sel > 0 && (!logits || !labels || !row_loss)))
    return MP_ERR_INVALID_ARG;
  if (!index && n_sel > n_rows) return MP_ERR_INVALID_ARG;
  if (n_sel == 0) return MP_OK;
  hipLaunchKernelGGL(softmax_ce_rows_kernel, dim3(flat_grid(n_sel)), dim3(kBlock), 0, as_stream(stream), logits, ld,
                     labels, index, n_sel, n_rows, C, row_loss);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_softmax_ce_bwd_f32(const float* logits, int64_t ld, int64_t n_rows, const int64_t* labels,
                          const int64_t* index, int64_t n_sel, int32_t C, const float* gscale, float inv_n,
                          float* dlogits, int64_t ldd, mp_stream_t stream) {
  if (n_sel < 0 || n_rows < 0 || C <= 0 || ld < C || ldd < C || !gscale ||
      (n_sel > 0 && (!logits || !labels || !dlogits)))
    return MP_ERR_INVALID_ARG;
  if (!index && n_sel > n_rows) return MP_ERR_INVALID_ARG;
  if (n_sel == 0) return MP_OK;
  if (index)
    hipLaunchKernelGGL(softmax_ce_bwd_kernel<true>, dim3(flat_grid(n_sel)), dim3(kBlock), 0, as_stream(stream), logits,
                       ld, labels, index, n_sel, n_rows, C, gscale, inv_n, dlogits, ldd);
  else
    hipLaunchKernelGGL(softmax_ce_bwd_kernel<false>, dim3(flat_grid(n_sel)), dim3(kBlock), 0, as_stream(stream), logits,
                       ld, labels, index, n_sel, n_rows, C, gscale, inv_n, dlogits, ldd);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // extern "C"

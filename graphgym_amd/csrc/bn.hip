// BatchNorm1d over the node axis in training mode, with the activation fused — the post-ops GraphGym wraps
// around every conv (graphgym/models/layer.py:26-35: BatchNorm1d(eps, mom) -> act) and the keras
// BatchNormalization inside the TF path's GIN MLPs (main_zd.py:181-186, 214-225).
//
// Why it is here: at the headline size ([10^7, 256] activations) torch's batch_norm_backward_reduce_kernel
// runs one workgroup per feature column and takes 0.52 s per call — 74 % of a GIN training step
// (profiles/r01_gin_step_library_bn.txt).  These are plain HBM-bound passes:
//   forward   stats  : one read of x            -> per-column sum / sum of squares (shifted by a pivot row)
//             apply  : read x, write y          -> y = act(x * scale + shift)
//   backward  stats  : read dy (, y), x         -> sum(g), sum(g * x)  with g = dy * [y > 0]
//             apply  : read dy (, y), x, write  -> dx = A * g + B * x + C   (per-column constants)
// Column sums are accumulated per workgroup over a contiguous run of rows (fp32, <= a few thousand terms
// per lane), written as partial slabs and combined in double precision in slab order: deterministic and
// accurate at 10^7 rows.
#include "common.h"
#include "vecio.h"

namespace mp {

constexpr int kBnMaxBlocks = kNumCU * 8;

struct BnGeom {
  int cg;        // column groups (threads along the feature axis), each owns W consecutive columns
  int rl;        // rows processed in parallel by one workgroup
  int tiles;     // column tiles when d > cg * W
};

template <int W>
__host__ __device__ inline BnGeom bn_geom(int d) {
  BnGeom g;
  const int groups = (d + W - 1) / W;
  g.cg = groups < kBlock ? groups : kBlock;
  int p = 1;
  while (p < g.cg) p <<= 1;             // power of two so that rl = 256 / cg is exact
  g.cg = p > kBlock ? kBlock : p;
  g.rl = kBlock / g.cg;
  g.tiles = (groups + g.cg - 1) / g.cg;
  return g;
}

// MODE 0: a = sum(x - pivot), b = sum((x - pivot)^2)          (forward statistics)
// MODE 1: a = sum(g),         b = sum(g * (x - pivot)),  g = dy * [y > 0] when y != nullptr, pivot = mean
// the affine scale of the apply pass, from the ROUNDED invstd the forward hands out: the forward and a backward that
// recomputes the activation from x get the same bits
__host__ __device__ __forceinline__ float bn_scale(float gamma, float invstd) {
  return (float)((double)gamma * (double)invstd);
}

template <int W, int MODE>
__global__ __launch_bounds__(kBlock) void bn_colsum_kernel(const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ dy, int64_t lddy,
                                                           const float* __restrict__ y, int64_t ldy,
                                                           const float* __restrict__ pivot, int64_t N, int32_t d,
                                                           int64_t rows_per_block, float* __restrict__ partial,
                                                           const float* __restrict__ mx_gamma = nullptr,
                                                           const float* __restrict__ mx_beta = nullptr,
                                                           const float* __restrict__ mx_invstd = nullptr) {
  // mx_invstd != NULL (MODE 1): the ReLU mask [y > 0] is recomputed from x — y = relu(fmaf(x - mean, scale, beta))
  // with scale = float(gamma * invstd), the forward's own expression on the forward's own operands, bit for bit —
  // instead of reading y
  __shared__ float red[2][kBlock][W];
  const BnGeom g = bn_geom<W>(d);
  const int cgi = threadIdx.x % g.cg;
  const int rli = threadIdx.x / g.cg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
  for (int t = 0; t < g.tiles; ++t) {
    const int c0 = (t * g.cg + cgi) * W;
    const bool on = c0 < d;
    float a[W], b[W], pv[W];
#pragma unroll
    for (int k = 0; k < W; ++k) { a[k] = 0.f; b[k] = 0.f; pv[k] = 0.f; }
    if (on) load_vec<W>(pivot + c0, pv);
    float sc[W], bt[W];
#pragma unroll
    for (int k = 0; k < W; ++k) { sc[k] = 0.f; bt[k] = 0.f; }
    if (MODE == 1 && mx_invstd != nullptr && on) {
#pragma unroll
      for (int k = 0; k < W; ++k)
        if (c0 + k < d) {
          sc[k] = bn_scale(mx_gamma ? mx_gamma[c0 + k] : 1.f, mx_invstd[c0 + k]);
          bt[k] = mx_beta ? mx_beta[c0 + k] : 0.f;
        }
    }
    if (on) {
      for (int64_t r = r0 + rli; r < r1; r += g.rl) {
        float xv[W];
        load_vec<W>(x + r * ldx + c0, xv);
        if (MODE == 0) {
#pragma unroll
          for (int k = 0; k < W; ++k) {
            const float v = xv[k] - pv[k];
            a[k] += v;
            b[k] = fmaf(v, v, b[k]);
          }
        } else {
          float gv[W];
          load_vec<W>(dy + r * lddy + c0, gv);
          if (mx_invstd != nullptr) {
#pragma unroll
            for (int k = 0; k < W; ++k) gv[k] = fmaf(xv[k] - pv[k], sc[k], bt[k]) > 0.f ? gv[k] : 0.f;
          } else if (y != nullptr) {
            float yv[W];
            load_vec<W>(y + r * ldy + c0, yv);
#pragma unroll
            for (int k = 0; k < W; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
          }
#pragma unroll
          for (int k = 0; k < W; ++k) {
            a[k] += gv[k];
            b[k] = fmaf(gv[k], xv[k] - pv[k], b[k]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) { red[0][threadIdx.x][k] = a[k]; red[1][threadIdx.x][k] = b[k]; }
    __syncthreads();
    if (rli == 0 && on) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        float sa = 0.f, sb = 0.f;
        for (int q = 0; q < g.rl; ++q) { sa += red[0][q * g.cg + cgi][k]; sb += red[1][q * g.cg + cgi][k]; }
        if (c0 + k < d) {
          partial[((int64_t)blockIdx.x * 2 + 0) * d + c0 + k] = sa;
          partial[((int64_t)blockIdx.x * 2 + 1) * d + c0 + k] = sb;
        }
      }
    }
    __syncthreads();
  }
}

// The two column sums of a finalize kernel over the nblk per-block partials, in double and in a fixed order: a
// workgroup owns 16 columns, its 16 block lanes each add every 16th partial through two independent running sums per
// quantity, and lane 0 adds the 16 lane sums in lane order.  (One thread per column walking all partials in a dependent
// chain of double adds took 0.8 ms per call at 2 048 partials.)  Returns true in the threads that hold a result.
__device__ __forceinline__ bool bn_col_sums(const float* __restrict__ partial, int nblk, int32_t d, int& c, double& s1,
                                            double& s2) {
  __shared__ double part[2][16][16];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  c = blockIdx.x * 16 + e;
  double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0;
  if (c < d) {
    int b = sl;
    for (; b + 16 < nblk; b += 32) {
      a1 += (double)partial[((int64_t)b * 2 + 0) * d + c];
      a2 += (double)partial[((int64_t)b * 2 + 1) * d + c];
      b1 += (double)partial[((int64_t)(b + 16) * 2 + 0) * d + c];
      b2 += (double)partial[((int64_t)(b + 16) * 2 + 1) * d + c];
    }
    if (b < nblk) {
      a1 += (double)partial[((int64_t)b * 2 + 0) * d + c];
      a2 += (double)partial[((int64_t)b * 2 + 1) * d + c];
    }
  }
  part[0][sl][e] = a1 + b1;
  part[1][sl][e] = a2 + b2;
  __syncthreads();
  s1 = 0.0; s2 = 0.0;
  if (sl == 0 && c < d) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1 += part[0][r][e]; s2 += part[1][r][e]; }
    return true;
  }
  return false;
}

// forward finalize: mean, invstd, unbiased variance, and the affine of the apply pass
__global__ __launch_bounds__(kBlock) void bn_fwd_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                                 const float* __restrict__ pivot, int64_t N,
                                                                 int32_t d, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps,
                                                                 float* mean, float* invstd, float* var_unbiased,
                                                                 float* scale, float* shift) {
  int c;
  double s1, s2;
  if (!bn_col_sums(partial, nblk, d, c, s1, s2)) return;
  const double m_shift = s1 / (double)N;
  double var = s2 / (double)N - m_shift * m_shift;      // biased (training normalisation)
  if (var < 0.0) var = 0.0;
  const double m = (double)pivot[c] + m_shift;
  const double istd = 1.0 / sqrt(var + (double)eps);
  mean[c] = (float)m;
  invstd[c] = (float)istd;
  var_unbiased[c] = (float)(N > 1 ? var * (double)N / (double)(N - 1) : var);
  const double gm = gamma ? (double)gamma[c] : 1.0;
  const double bt = beta ? (double)beta[c] : 0.0;
  scale[c] = bn_scale(gamma ? gamma[c] : 1.f, (float)istd);
  shift[c] = (float)bt;                                   // the apply pass computes (x - mean) * scale + beta
}

// backward finalize: dgamma, dbeta and the constants of dx = A g + B x + C
__global__ __launch_bounds__(kBlock) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                                 int64_t N, int32_t d,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, float* dgamma,
                                                                 float* dbeta, float* A, float* B, float* Cc,
                                                                 float* fwd_scale) {
  int c;
  double sg, sgx;
  if (!bn_col_sums(partial, nblk, d, c, sg, sgx)) return;
  const double m = mean[c], istd = invstd[c];
  const double gm = gamma ? (double)gamma[c] : 1.0;
  const double dgam = sgx * istd;                       // sum g * xhat (the sums were taken against the mean)
  (void)m;
  if (dgamma) dgamma[c] = (float)dgam;
  if (dbeta) dbeta[c] = (float)sg;
  const double a = gm * istd;
  const double bb = -gm * istd * istd * dgam / (double)N;
  A[c] = (float)a;
  B[c] = (float)bb;
  Cc[c] = (float)(-a * sg / (double)N);                 // dx = A g + B (x - mean) + C
  if (fwd_scale != nullptr) fwd_scale[c] = bn_scale(gamma ? gamma[c] : 1.f, invstd[c]);   // for the recomputed ReLU mask
}

// MODE 0: out = act((x - ctr) * p0 + p1)        MODE 1: out = p0 * g + p1 * (x - ctr) + p2,  g = dy * [y > 0]
template <int W, int MODE>
__global__ __launch_bounds__(kBlock) void bn_apply_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ dy, int64_t lddy,
                                                          const float* __restrict__ y, int64_t ldy,
                                                          const float* __restrict__ p0, const float* __restrict__ p1,
                                                          const float* __restrict__ p2,
                                                          const float* __restrict__ ctr, int relu, int64_t N,
                                                          int32_t d, float* __restrict__ out, int64_t ldo,
                                                          const float* __restrict__ mx_scale = nullptr,
                                                          const float* __restrict__ mx_beta = nullptr) {
  const int groups = (d + W - 1) / W;
  const int64_t total = N * groups;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / groups;
    const int c0 = (int)(i - r * groups) * W;
    float xv[W], a[W], b[W], o[W], mv[W];
    load_vec<W>(x + r * ldx + c0, xv);
    load_vec<W>(p0 + c0, a);
    load_vec<W>(p1 + c0, b);
    load_vec<W>(ctr + c0, mv);
#pragma unroll
    for (int k = 0; k < W; ++k) xv[k] -= mv[k];
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        o[k] = fmaf(xv[k], a[k], b[k]);
        if (relu) o[k] = fmaxf(o[k], 0.f);
      }
    } else {
      float gv[W], cv[W];
      load_vec<W>(dy + r * lddy + c0, gv);
      load_vec<W>(p2 + c0, cv);
      if (mx_scale != nullptr) {      // the forward's activation, recomputed (see bn_colsum_kernel); the scale
        float sck[W], btk[W];         // vector was rebuilt by the finalize kernel
        load_vec<W>(mx_scale + c0, sck);
#pragma unroll
        for (int k = 0; k < W; ++k) btk[k] = 0.f;
        if (mx_beta != nullptr) load_vec<W>(mx_beta + c0, btk);
#pragma unroll
        for (int k = 0; k < W; ++k) gv[k] = fmaf(xv[k], sck[k], btk[k]) > 0.f ? gv[k] : 0.f;
      } else if (y != nullptr) {
        float yv[W];
        load_vec<W>(y + r * ldy + c0, yv);
#pragma unroll
        for (int k = 0; k < W; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < W; ++k) o[k] = fmaf(a[k], gv[k], fmaf(b[k], xv[k], cv[k]));
    }
    store_vec<W>(out + r * ldo + c0, o);
  }
}

static int bn_blocks(int64_t N) {
  int64_t b = ceil_div(N, 64);
  if (b < 1) b = 1;
  if (b > kBnMaxBlocks) b = kBnMaxBlocks;
  return (int)b;
}

static bool bn_al(const void* p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; }

struct BnWs { float* partial; float* v[5]; size_t total; };
static void bn_ws_layout(int64_t N, int32_t d, void* base, BnWs* w) {
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes, 256); return r; };
  w->partial = (float*)take((size_t)bn_blocks(N) * 2 * d * 4);
  for (int i = 0; i < 5; ++i) w->v[i] = (float*)take((size_t)d * 4);
  w->total = off;
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_bn_ws_bytes(int64_t N, int32_t d, size_t* bytes_host) {
  if (!bytes_host || N < 0 || d <= 0) return MP_ERR_INVALID_ARG;
  BnWs w;
  bn_ws_layout(N, d, nullptr, &w);
  *bytes_host = w.total;
  return MP_OK;
}

int mp_bn_train_fwd_f32(const float* x, int64_t ldx, int64_t N, int32_t d, const float* gamma, const float* beta,
                        float eps, int relu, float* y, int64_t ldy, float* mean, float* invstd,
                        float* var_unbiased, void* ws, size_t ws_bytes, mp_stream_t stream) {
  if (N <= 0 || d <= 0 || !x || !y || !mean || !invstd || !var_unbiased || ldx < d || ldy < d) return MP_ERR_INVALID_ARG;
  BnWs L;
  bn_ws_layout(N, d, ws, &L);
  if (!ws || ws_bytes < L.total) return MP_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  const int nblk = bn_blocks(N);
  const int64_t rpb = ceil_div(N, nblk);
  const bool vec = d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && bn_al(x, 16) && bn_al(y, 16);
  const float* pivot = x;     // row 0: any sample of the column keeps the shifted sums well conditioned
  if (vec)
    hipLaunchKernelGGL((bn_colsum_kernel<4, 0>), dim3(nblk), dim3(kBlock), 0, st, x, ldx, nullptr, 0, nullptr, 0, pivot,
                       N, d, rpb, L.partial);
  else
    hipLaunchKernelGGL((bn_colsum_kernel<1, 0>), dim3(nblk), dim3(kBlock), 0, st, x, ldx, nullptr, 0, nullptr, 0, pivot,
                       N, d, rpb, L.partial);
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((unsigned)ceil_div(d, 16)), dim3(kBlock), 0, st, L.partial, nblk,
                     pivot, N, d, gamma, beta, eps, mean, invstd, var_unbiased, L.v[0], L.v[1]);
  MP_LAUNCH_CHECK();
  if (vec)
    hipLaunchKernelGGL((bn_apply_kernel<4, 0>), dim3(flat_grid(N * (d / 4))), dim3(kBlock), 0, st, x, ldx, nullptr, 0,
                       nullptr, 0, L.v[0], L.v[1], nullptr, mean, relu, N, d, y, ldy);
  else
    hipLaunchKernelGGL((bn_apply_kernel<1, 0>), dim3(flat_grid(N * d)), dim3(kBlock), 0, st, x, ldx, nullptr, 0,
                       nullptr, 0, L.v[0], L.v[1], nullptr, mean, relu, N, d, y, ldy);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

static int bn_bwd_common(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                         int64_t N, int32_t d, const float* gamma, const float* beta, int mask_from_x,
                         const float* mean, const float* invstd, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                         void* ws, size_t ws_bytes, hipStream_t st) {
  if (N <= 0 || d <= 0 || !dy || !x || !mean || !invstd || !dx || lddy < d || ldx < d || lddx < d || (y && ldy < d))
    return MP_ERR_INVALID_ARG;
  BnWs L;
  bn_ws_layout(N, d, ws, &L);
  if (!ws || ws_bytes < L.total) return MP_ERR_WORKSPACE;
  const int nblk = bn_blocks(N);
  const int64_t rpb = ceil_div(N, nblk);
  const bool vec = d % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && (!y || ldy % 4 == 0) &&
                   bn_al(x, 16) && bn_al(dy, 16) && bn_al(y, 16) && bn_al(dx, 16) && (!mask_from_x || bn_al(beta, 16));
  const float* mg = mask_from_x ? gamma : nullptr;
  const float* mb = mask_from_x ? beta : nullptr;
  const float* mi = mask_from_x ? invstd : nullptr;
  if (vec)
    hipLaunchKernelGGL((bn_colsum_kernel<4, 1>), dim3(nblk), dim3(kBlock), 0, st, x, ldx, dy, lddy, y, ldy, mean, N,
                       d, rpb, L.partial, mg, mb, mi);
  else
    hipLaunchKernelGGL((bn_colsum_kernel<1, 1>), dim3(nblk), dim3(kBlock), 0, st, x, ldx, dy, lddy, y, ldy, mean, N,
                       d, rpb, L.partial, mg, mb, mi);
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)ceil_div(d, 16)), dim3(kBlock), 0, st, L.partial, nblk,
                     N, d, gamma, mean, invstd, dgamma, dbeta, L.v[0], L.v[1], L.v[2], mask_from_x ? L.v[3] : nullptr);
  MP_LAUNCH_CHECK();
  const float* msc = mask_from_x ? L.v[3] : nullptr;
  if (vec)
    hipLaunchKernelGGL((bn_apply_kernel<4, 1>), dim3(flat_grid(N * (d / 4))), dim3(kBlock), 0, st, x, ldx, dy, lddy, y,
                       ldy, L.v[0], L.v[1], L.v[2], mean, 0, N, d, dx, lddx, msc, mb);
  else
    hipLaunchKernelGGL((bn_apply_kernel<1, 1>), dim3(flat_grid(N * d)), dim3(kBlock), 0, st, x, ldx, dy, lddy, y, ldy,
                       L.v[0], L.v[1], L.v[2], mean, 0, N, d, dx, lddx, msc, mb);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_bn_train_bwd_f32(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                        int64_t N, int32_t d, const float* gamma, const float* mean, const float* invstd,
                        float* dx, int64_t lddx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                        mp_stream_t stream) {
  return bn_bwd_common(dy, lddy, y, ldy, x, ldx, N, d, gamma, nullptr, 0, mean, invstd, dx, lddx, dgamma, dbeta, ws,
                       ws_bytes, as_stream(stream));
}

int mp_bn_train_bwd_relu_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t N, int32_t d,
                             const float* gamma, const float* beta, const float* mean, const float* invstd,
                             float* dx, int64_t lddx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                             mp_stream_t stream) {
  return bn_bwd_common(dy, lddy, nullptr, 0, x, ldx, N, d, gamma, beta, 1, mean, invstd, dx, lddx, dgamma, dbeta, ws,
                       ws_bytes, as_stream(stream));
}

}  // extern "C"

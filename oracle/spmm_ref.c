/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke, bench.py's
 * cpu_baseline leg).  Never linked into or loaded by the product (graphgym_amd/).
 *
 * Plain-C restatement of the reference's COO aggregation, edge by edge in input
 * order, the way its CPU path evaluates it:
 *   gather x[col[e]] -> scale by w[e] -> segment-reduce into out[row[e]]
 *   SparseAdj.matmul                 sparse_adj.py:91-97  (tf.gather, Mul, unsorted_segment_sum)
 *   MessagePassing.propagate+scatter idconv.py:89-92,177-180 (torch_scatter 'add'/'mean'/'max')
 *   gcn_norm_adj                     TfgIDLayer.py:528-566
 * PARITY UNPINNED: the reference holds no golden vectors for this path; this file is a
 * second, independent formulation used to cross-check oracle/ref_ops.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { REF_SUM = 0, REF_MEAN = 1, REF_MAX = 2 };

/* out [N, d] row-major.  w may be NULL (ones).  argmax (may be NULL) [N, d] receives the
 * edge position of the winner for REF_MAX (first strictly greater wins), -1 for empty rows. */
int ref_coo_aggregate_f32(const int64_t* row, const int64_t* col, const float* w, int64_t E,
                          const float* x, int64_t d, int64_t N, int reduce, float* out,
                          int64_t* argmax) {
  int64_t* cnt = (int64_t*)calloc((size_t)(N > 0 ? N : 1), sizeof(int64_t));
  if (!cnt) return 1;
  for (int64_t i = 0; i < N * d; ++i) out[i] = (reduce == REF_MAX) ? -INFINITY : 0.0f;
  if (argmax) for (int64_t i = 0; i < N * d; ++i) argmax[i] = -1;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t r = row[e], c = col[e];
    if (r < 0 || r >= N || c < 0) { free(cnt); return 2; }
    const float we = w ? w[e] : 1.0f;
    const float* xs = x + c * d;
    float* o = out + r * d;
    cnt[r] += 1;
    if (reduce == REF_MAX) {
      for (int64_t k = 0; k < d; ++k) {
        const float m = we * xs[k];
        if (m > o[k]) { o[k] = m; if (argmax) argmax[r * d + k] = e; }
      }
    } else {
      for (int64_t k = 0; k < d; ++k) o[k] += we * xs[k];   /* message then add: two roundings, as the reference */
    }
  }
  for (int64_t r = 0; r < N; ++r) {
    float* o = out + r * d;
    if (reduce == REF_MEAN && cnt[r] > 0) for (int64_t k = 0; k < d; ++k) o[k] /= (float)cnt[r];
    if (reduce == REF_MAX && cnt[r] == 0) for (int64_t k = 0; k < d; ++k) o[k] = 0.0f;
  }
  free(cnt);
  return 0;
}

/* TF-flavour GCN normalisation on a COO list that already carries its self loops:
 * deg by row, dinv = deg^-1/2 (inf/nan -> 0), w_out = dinv[row] * w * dinv[col]. */
int ref_gcn_norm_f32(const int64_t* row, const int64_t* col, const float* w, int64_t E, int64_t N,
                     int deg_by_col, float* w_out) {
  float* deg = (float*)calloc((size_t)(N > 0 ? N : 1), sizeof(float));
  if (!deg) return 1;
  for (int64_t e = 0; e < E; ++e) deg[deg_by_col ? col[e] : row[e]] += w ? w[e] : 1.0f;
  for (int64_t i = 0; i < N; ++i) {
    float v = powf(deg[i], -0.5f);
    if (isinf(v) || isnan(v)) v = 0.0f;
    deg[i] = v;
  }
  for (int64_t e = 0; e < E; ++e) w_out[e] = deg[row[e]] * (w ? w[e] : 1.0f) * deg[col[e]];
  free(deg);
  return 0;
}

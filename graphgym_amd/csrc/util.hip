// Status strings, error capture, and the host-side synthetic graph generator.
#include <algorithm>
#include "common.h"
#include <string>
#include <vector>

namespace mp {
static thread_local std::string g_last_hip_error;
void set_hip_error(hipError_t e, const char* what) {
  g_last_hip_error = std::string(what) + ": " + hipGetErrorString(e);
}
}  // namespace mp

namespace mp {
typedef float cp_f32x4 __attribute__((ext_vector_type(4)));
// plain streaming copy, 16 B per lane: the yardstick the aggregation's HBM rate is quoted against
__global__ __launch_bounds__(kBlock) void copy_probe_kernel(const cp_f32x4* __restrict__ src,
                                                            cp_f32x4* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(src[i], dst + i);
}
// read-only stream: every lane sums its 16-byte loads; one float per lane is written so nothing is elided
__global__ __launch_bounds__(kBlock) void read_probe_kernel(const cp_f32x4* __restrict__ src, int64_t n4,
                                                            float* __restrict__ sink) {
  cp_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    acc += src[i];
  sink[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
}  // namespace mp

extern "C" {

int mp_version(void) { return 100; }

int mp_read_probe_f32(const float* src, int64_t n, float* sink, mp_stream_t stream) {
  if (n < 0 || !sink || (n > 0 && !src) || n % 4 || ((uintptr_t)src % 16)) return MP_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mp::read_probe_kernel, dim3(mp::kNumCU * 8), dim3(mp::kBlock), 0, mp::as_stream(stream),
                     reinterpret_cast<const mp::cp_f32x4*>(src), n / 4, sink);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_copy_probe_f32(const float* src, float* dst, int64_t n, mp_stream_t stream) {
  if (n < 0 || (n > 0 && (!src || !dst)) || n % 4 || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16))
    return MP_ERR_INVALID_ARG;
  if (n == 0) return MP_OK;
  hipLaunchKernelGGL(mp::copy_probe_kernel, dim3(mp::kNumCU * 8), dim3(mp::kBlock), 0, mp::as_stream(stream),
                     reinterpret_cast<const mp::cp_f32x4*>(src), reinterpret_cast<mp::cp_f32x4*>(dst), n / 4);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int mp_stream_create_cu_mask(const uint32_t* mask, int n_words, mp_stream_t* stream) {
  if (!mask || n_words <= 0 || !stream) return MP_ERR_INVALID_ARG;
  hipStream_t s = nullptr;
  MP_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask));
  *stream = (mp_stream_t)s;
  return MP_OK;
}

int mp_stream_destroy(mp_stream_t stream) {
  if (!stream) return MP_ERR_INVALID_ARG;
  MP_HIP(hipStreamDestroy(mp::as_stream(stream)));
  return MP_OK;
}

const char* mp_status_str(int status) {
  switch (status) {
    case MP_OK: return "ok";
    case MP_ERR_INVALID_ARG: return "invalid argument";
    case MP_ERR_UNSUPPORTED: return "unsupported size (index does not fit int32)";
    case MP_ERR_WORKSPACE: return "workspace too small";
    case MP_ERR_HIP: return "HIP runtime error";
    case MP_ERR_ALIGNMENT: return "misaligned pointer or leading dimension";
  }
  return "unknown status";
}

const char* mp_last_hip_error(void) { return mp::g_last_hip_error.c_str(); }

// Longest-processing-time assignment of independent units (graphs / ego nets, cost = stored entries) to ranks: units by
// descending cost (ties: lower index first), each to the least-loaded rank (ties: lower rank) — the host logic of the
// data-parallel sharding (graphgym_amd/dist.py: lpt_partition), here for the per-step cadence where a
// global batch holds 10^4-10^5 units.  Deterministic: every rank computes the same owners from the same costs.
int mp_lpt_partition_host(const int64_t* costs_host, int64_t n, int32_t world, int32_t* owner_host) {
  if (n < 0 || world < 1 || (n > 0 && (!costs_host || !owner_host))) return MP_ERR_INVALID_ARG;
  std::vector<int64_t> order((size_t)n);
  for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
    return costs_host[a] != costs_host[b] ? costs_host[a] > costs_host[b] : a < b;
  });
  std::vector<int64_t> load((size_t)world, 0);
  for (int64_t k = 0; k < n; ++k) {
    int32_t best = 0;
    for (int32_t r = 1; r < world; ++r)
      if (load[(size_t)r] < load[(size_t)best]) best = r;
    owner_host[order[(size_t)k]] = best;
    load[(size_t)best] += costs_host[order[(size_t)k]];
  }
  return MP_OK;
}

// Barabasi-Albert by the repeated-endpoints list: a new node t draws m targets
// uniformly from the list of all edge endpoints so far (probability ∝ degree),
// then both endpoints of each new edge are appended.  The m seed nodes start
// with one list entry each so they can be drawn.  Duplicate targets within one
// node's m draws are kept here; the CSR build of the caller may merge them.
int mp_gen_ba_edges_host(int64_t n, int32_t m, uint64_t seed, int64_t* u_host, int64_t* v_host,
                         int64_t* n_edges_host) {
  if (n <= m || m < 1 || !u_host || !v_host || !n_edges_host) return MP_ERR_INVALID_ARG;
  if (n >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  std::vector<int32_t> ends;
  ends.reserve((size_t)(2 * (int64_t)m * n + m));
  for (int32_t i = 0; i < m; ++i) ends.push_back(i);
  uint64_t s = seed ? seed : 0x9E3779B97F4A7C15ull;
  auto next = [&]() {  // splitmix64
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  int64_t k = 0;
  for (int64_t t = m; t < n; ++t) {
    const size_t pool = ends.size();
    for (int32_t j = 0; j < m; ++j) {
      // multiply-shift maps a 64-bit draw onto [0, pool)
      const size_t r = (size_t)(((unsigned __int128)next() * pool) >> 64);
      const int32_t tgt = ends[r];
      u_host[k] = t;
      v_host[k] = tgt;
      ++k;
    }
    for (int32_t j = 0; j < m; ++j) {
      ends.push_back((int32_t)t);
      ends.push_back((int32_t)v_host[k - m + j]);
    }
  }
  *n_edges_host = k;
  return MP_OK;
}

// Holme-Kim "powerlaw cluster" growth (networkx.powerlaw_cluster_graph, the generator behind the reference's
// datasets/scalefree.pkl, datasets/syn_graph.py:42): each new node makes m links; after a preferential
// attachment step to target t, with probability p the next link closes a triangle by going to a uniformly
// chosen neighbour of t that the new node is not linked to yet, else it is another preferential step.
// Adjacency lists are kept on the host (about 0.7 GB at n = 10^7, m = 5).  Links of one node are distinct.
int mp_gen_powerlaw_cluster_edges_host(int64_t n, int32_t m, double p, uint64_t seed, int64_t* u_host,
                                       int64_t* v_host, int64_t* n_edges_host) {
  if (n <= m || m < 1 || p < 0.0 || p > 1.0 || !u_host || !v_host || !n_edges_host) return MP_ERR_INVALID_ARG;
  if (n >= INT32_MAX) return MP_ERR_UNSUPPORTED;
  std::vector<std::vector<int32_t>> adj((size_t)n);
  std::vector<int32_t> ends;                 // every edge endpoint once: degree-proportional sampling
  ends.reserve((size_t)(2 * (int64_t)m * n + m));
  for (int32_t i = 0; i < m; ++i) ends.push_back(i);
  uint64_t s = seed ? seed : 0x9E3779B97F4A7C15ull;
  auto next = [&]() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  auto below = [&](size_t k) { return (size_t)(((unsigned __int128)next() * k) >> 64); };
  auto unit = [&]() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); };
  int64_t k = 0;
  std::vector<int32_t> mine;
  mine.reserve((size_t)m);
  for (int64_t t = m; t < n; ++t) {
    mine.clear();
    auto linked = [&](int32_t x) { for (int32_t y : mine) if (y == x) return true; return false; };
    const size_t pool = ends.size();
    int32_t last = -1;
    int guard = 0;
    while ((int32_t)mine.size() < m && guard < 64 * m) {
      ++guard;
      int32_t tgt = -1;
      if (last >= 0 && unit() < p && !adj[(size_t)last].empty()) {
        const int32_t cand = adj[(size_t)last][below(adj[(size_t)last].size())];   // triad formation
        if (cand != (int32_t)t && !linked(cand)) tgt = cand;
      }
      if (tgt < 0) {
        const int32_t cand = ends[below(pool)];                                     // preferential attachment
        if (linked(cand)) continue;
        tgt = cand;
      }
      mine.push_back(tgt);
      last = tgt;
    }
    for (int32_t tgt : mine) {
      u_host[k] = t;
      v_host[k] = tgt;
      ++k;
      adj[(size_t)t].push_back(tgt);
      adj[(size_t)tgt].push_back((int32_t)t);
      ends.push_back((int32_t)t);
      ends.push_back(tgt);
    }
  }
  *n_edges_host = k;
  return MP_OK;
}

}  // extern "C"

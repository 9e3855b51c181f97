// GPU ego-net batcher: the ID-GNN "Full" sampler of graphgym/models/transform.py:11-38 for a
// batch of centre nodes, on the device.
//
// The reference loops over every node in Python (nx.ego_graph + nx.relabel_nodes), a ~60x data
// blow-up built on one CPU thread.  Here a batch of B centres is expanded at once:
//   * one bitmap of N bits per centre; `radius` level-synchronous sweeps mark the members
//     (frontier bitmap -> atomicOr into visited / next), radius > 4 means the whole graph
//     (transform.py:18-19);
//   * new ids follow the reference: centre c keeps id c (0..B-1), the other members of ego c get
//     fresh consecutive ids in ascending original-id order (the order nx's subgraph view iterates),
//     egos laid out one after another (transform.py:24-36);
//   * the induced edges are emitted as a COO list in the new ids; `orig` maps every new node back
//     to its original id (to gather features / labels).
// The COO goes straight into mp_csr_from_coo; node_id_index is arange(B) (transform.py:38).
#include "common.h"

namespace mp {

struct EgoWs {
  uint32_t* visited;   // [B, W]
  uint32_t* frontier;  // [B, W]
  uint32_t* next;      // [B, W]
  int32_t* wprefix;    // [B, W]  members before word w
  int32_t* members;    // [B]
  int64_t* node_off;   // [B+1]   first fresh id of ego c (node_off[0] = B)
  int64_t* edge_cnt;   // [B]
  int64_t* edge_off;   // [B+1]
  unsigned long long* cursor;  // [B]
  size_t total;
};

static void ego_layout(int64_t N, int64_t B, void* base, EgoWs* w) {
  const size_t W = (size_t)ceil_div(N, 32);
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align_up(bytes, 256); return r; };
  w->visited = (uint32_t*)take(B * W * 4);
  w->frontier = (uint32_t*)take(B * W * 4);
  w->next = (uint32_t*)take(B * W * 4);
  w->wprefix = (int32_t*)take(B * W * 4);
  w->members = (int32_t*)take(B * 4);
  w->node_off = (int64_t*)take((B + 1) * 8);
  w->edge_cnt = (int64_t*)take(B * 8);
  w->edge_off = (int64_t*)take((B + 1) * 8);
  w->cursor = (unsigned long long*)take(B * 8);
  w->total = off;
}

__global__ __launch_bounds__(kBlock) void ego_seed_kernel(const int64_t* __restrict__ centres, int64_t B,
                                                          int64_t W, int whole_graph, int64_t N,
                                                          uint32_t* visited, uint32_t* frontier) {
  // whole_graph: every bit of [0, N) set (radius > 4); else only the centre's bit
  const int64_t total = B * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i / W, w = i - c * W;
    uint32_t bits = 0;
    if (whole_graph) {
      const int64_t lo = w * 32;
      const int64_t n = N - lo >= 32 ? 32 : N - lo;
      bits = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
    } else {
      const int64_t v = centres[c];
      if ((v >> 5) == w) bits = 1u << (v & 31);
    }
    visited[i] = bits;
    frontier[i] = whole_graph ? 0u : bits;
  }
}

// Wave-wide walk over the set bits of a bitmap: a wave loads 64 words at once, skips empty ones with a
// ballot, and hands every member node v to `body(v)` with v wave-uniform — so the 64 lanes can split
// v's neighbour list (coalesced col reads; a 10^4-neighbour hub no longer serialises one thread).
template <class Body>
__device__ __forceinline__ void for_each_member(const uint32_t* __restrict__ bitmap, int64_t W, Body body) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * kWave;
  for (int64_t wbase = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * kWave; wbase < W; wbase += stride) {
    const int64_t w = wbase + lane;
    const uint32_t mine = w < W ? bitmap[w] : 0u;
    unsigned long long nz = __ballot(mine != 0u);
    while (nz) {
      const int src = __builtin_ctzll(nz);
      nz &= nz - 1;
      uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)mine, src);
      while (bits) {
        const int b = __builtin_ctz(bits);
        bits &= bits - 1;
        body((uint32_t)((wbase + src) * 32 + b));
      }
    }
  }
}

// one BFS level: grid (chunks, B); every member of `frontier` pushes its neighbours
__global__ __launch_bounds__(kBlock) void ego_level_kernel(const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col, int64_t W,
                                                           const uint32_t* __restrict__ frontier,
                                                           uint32_t* visited, uint32_t* next) {
  const int64_t c = blockIdx.y;
  const int lane = threadIdx.x & 63;
  uint32_t* vis = visited + c * W;
  uint32_t* nx = next + c * W;
  for_each_member(frontier + c * W, W, [&](uint32_t v) {
    const int s = rowptr[v], e = rowptr[v + 1];
    for (int k = s + lane; k < e; k += kWave) {
      const uint32_t u = (uint32_t)col[k] & 0x7fffffffu;
      const uint32_t m = 1u << (u & 31);
      if (!(vis[u >> 5] & m)) {
        const uint32_t old = atomicOr(&vis[u >> 5], m);
        if (!(old & m)) atomicOr(&nx[u >> 5], m);
      }
    }
  });
}

// per centre: exclusive prefix of popcounts over the bitmap words; one workgroup per centre
__global__ __launch_bounds__(kBlock) void ego_prefix_kernel(const uint32_t* __restrict__ visited, int64_t W,
                                                            int32_t* wprefix, int32_t* members) {
  __shared__ int32_t part[kBlock];
  __shared__ int32_t carry_s;
  const int64_t c = blockIdx.x;
  const uint32_t* vis = visited + c * W;
  int32_t* wp = wprefix + c * W;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < W; base += kBlock) {
    const int64_t w = base + threadIdx.x;
    const int32_t cnt = w < W ? __popc(vis[w]) : 0;
    part[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {   // Hillis-Steele inclusive scan
      int32_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    const int32_t carry = carry_s;
    if (w < W) wp[w] = carry + part[threadIdx.x] - cnt;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry_s = carry + part[kBlock - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) members[c] = carry_s;
}

// node_off[c] = B + sum_{c' < c} (members[c'] - 1); single workgroup, B is a batch size
__global__ void ego_node_off_kernel(const int32_t* __restrict__ members, int64_t B, int64_t* node_off) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int64_t acc = B;
    for (int64_t c = 0; c < B; ++c) { node_off[c] = acc; acc += members[c] - 1; }
    node_off[B] = acc;
  }
}

__device__ __forceinline__ int64_t ego_new_id(const uint32_t* vis, const int32_t* wp, int64_t node_off_c,
                                              int64_t centre, int64_t c, uint32_t u) {
  if ((int64_t)u == centre) return c;
  const uint32_t word = vis[u >> 5];
  int64_t rank = wp[u >> 5] + __popc(word & ((1u << (u & 31)) - 1u));
  if ((int64_t)u > centre) rank -= 1;   // the centre is not among the fresh ids
  return node_off_c + rank;
}

// pass 0: count induced edges per centre; pass 1: emit them (atomic cursor per centre) and orig ids
template <int PASS>
__global__ __launch_bounds__(kBlock) void ego_edges_kernel(const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col, int64_t W,
                                                           const int64_t* __restrict__ centres,
                                                           const uint32_t* __restrict__ visited,
                                                           const int32_t* __restrict__ wprefix,
                                                           const int64_t* __restrict__ node_off,
                                                           const int64_t* __restrict__ edge_off,
                                                           int64_t* edge_cnt, unsigned long long* cursor,
                                                           int64_t* out_src, int64_t* out_dst, int64_t* orig,
                                                           int32_t* ego_of) {
  const int64_t c = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const uint32_t* vis = visited + c * W;
  const int32_t* wp = wprefix + c * W;
  const int64_t centre = centres[c];
  const int64_t noff = PASS ? node_off[c] : 0;
  const int64_t eoff = PASS ? edge_off[c] : 0;
  long long local = 0;
  for_each_member(vis, W, [&](uint32_t v) {
    const int s = rowptr[v], e = rowptr[v + 1];
    int64_t vid = 0;
    if (PASS) {
      vid = ego_new_id(vis, wp, noff, centre, c, v);
      if (lane == 0) {
        orig[vid] = (int64_t)v;
        if (ego_of) ego_of[vid] = (int32_t)c;
      }
    }
    for (int k = s + lane; k < e; k += kWave) {
      const uint32_t u = (uint32_t)col[k] & 0x7fffffffu;
      if (vis[u >> 5] & (1u << (u & 31))) {
        if (PASS) {
          const unsigned long long slot = atomicAdd(&cursor[c], 1ull);   // one add per wave after hipcc's coalescing
          const int64_t o = eoff + (int64_t)slot;
          out_dst[o] = vid;                                              // row v holds v's in-edges
          out_src[o] = ego_new_id(vis, wp, noff, centre, c, u);
        } else {
          ++local;
        }
      }
    }
  });
  if (!PASS) {
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, kWave);
    if (lane == 0 && local) atomicAdd((unsigned long long*)&edge_cnt[c], (unsigned long long)local);
  }
}

__global__ void ego_edge_off_kernel(const int64_t* __restrict__ edge_cnt, int64_t B, int64_t* edge_off) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int64_t acc = 0;
    for (int64_t c = 0; c < B; ++c) { edge_off[c] = acc; acc += edge_cnt[c]; }
    edge_off[B] = acc;
  }
}

static dim3 ego_grid(int64_t W, int64_t B) {
  int64_t bx = ceil_div(W, kBlock);   // one wave per 64 words, four waves per block
  if (bx < 1) bx = 1;
  if (bx > 512) bx = 512;
  return dim3((unsigned)bx, (unsigned)B);
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_ego_ws_bytes(int64_t N, int64_t n_centres, size_t* bytes_host) {
  if (!bytes_host || N < 0 || n_centres < 0) return MP_ERR_INVALID_ARG;
  if (N >= INT32_MAX || n_centres > 65535) return MP_ERR_UNSUPPORTED;
  EgoWs w;
  ego_layout(N, n_centres, nullptr, &w);
  *bytes_host = w.total;
  return MP_OK;
}

int mp_ego_expand_count(const int32_t* rowptr, const int32_t* col, int64_t N, const int64_t* centres,
                        int64_t n_centres, int32_t radius, void* ws, size_t ws_bytes, int64_t* counts_host,
                        mp_stream_t stream) {
  if (!rowptr || !centres || !counts_host || N <= 0 || n_centres <= 0 || radius < 0) return MP_ERR_INVALID_ARG;
  if (N >= INT32_MAX || n_centres > 65535) return MP_ERR_UNSUPPORTED;
  EgoWs L;
  ego_layout(N, n_centres, ws, &L);
  if (!ws || ws_bytes < L.total) return MP_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  const int64_t B = n_centres, W = ceil_div(N, 32);
  const int whole = radius > 4;   // transform.py:18-19
  hipLaunchKernelGGL(ego_seed_kernel, dim3(flat_grid(B * W)), dim3(kBlock), 0, st, centres, B, W, whole, N,
                     L.visited, L.frontier);
  MP_LAUNCH_CHECK();
  if (!whole) {
    uint32_t* fr = L.frontier;
    uint32_t* nx = L.next;
    for (int lvl = 0; lvl < radius; ++lvl) {
      MP_HIP(hipMemsetAsync(nx, 0, (size_t)B * W * 4, st));
      hipLaunchKernelGGL(ego_level_kernel, ego_grid(W, B), dim3(kBlock), 0, st, rowptr, col, W, fr, L.visited, nx);
      MP_LAUNCH_CHECK();
      uint32_t* t = fr; fr = nx; nx = t;
    }
  }
  hipLaunchKernelGGL(ego_prefix_kernel, dim3((unsigned)B), dim3(kBlock), 0, st, L.visited, W, L.wprefix, L.members);
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(ego_node_off_kernel, dim3(1), dim3(64), 0, st, L.members, B, L.node_off);
  MP_LAUNCH_CHECK();
  MP_HIP(hipMemsetAsync(L.edge_cnt, 0, (size_t)B * 8, st));
  hipLaunchKernelGGL(ego_edges_kernel<0>, ego_grid(W, B), dim3(kBlock), 0, st, rowptr, col, W, centres, L.visited,
                     L.wprefix, L.node_off, L.edge_off, L.edge_cnt, L.cursor, nullptr, nullptr, nullptr, nullptr);
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(ego_edge_off_kernel, dim3(1), dim3(64), 0, st, L.edge_cnt, B, L.edge_off);
  MP_LAUNCH_CHECK();
  int64_t tot[2];
  MP_HIP(hipMemcpyAsync(&tot[0], L.node_off + B, 8, hipMemcpyDeviceToHost, st));
  MP_HIP(hipMemcpyAsync(&tot[1], L.edge_off + B, 8, hipMemcpyDeviceToHost, st));
  MP_HIP(hipStreamSynchronize(st));
  counts_host[0] = tot[0];
  counts_host[1] = tot[1];
  return MP_OK;
}

int mp_ego_expand_emit(const int32_t* rowptr, const int32_t* col, int64_t N, const int64_t* centres,
                       int64_t n_centres, void* ws, size_t ws_bytes, int64_t* out_src, int64_t* out_dst,
                       int64_t* orig_node, int32_t* ego_of_node, mp_stream_t stream) {
  if (!rowptr || !centres || !out_src || !out_dst || !orig_node || N <= 0 || n_centres <= 0) return MP_ERR_INVALID_ARG;
  EgoWs L;
  ego_layout(N, n_centres, ws, &L);
  if (!ws || ws_bytes < L.total) return MP_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  const int64_t B = n_centres, W = ceil_div(N, 32);
  MP_HIP(hipMemsetAsync(L.cursor, 0, (size_t)B * 8, st));
  hipLaunchKernelGGL(ego_edges_kernel<1>, ego_grid(W, B), dim3(kBlock), 0, st, rowptr, col, W, centres, L.visited,
                     L.wprefix, L.node_off, L.edge_off, L.edge_cnt, L.cursor, out_src, out_dst, orig_node,
                     ego_of_node);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // extern "C"

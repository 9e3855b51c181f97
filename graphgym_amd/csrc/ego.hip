// GPU ego-net batcher: the ID-GNN "Full" sampler of graphgym/models/transform.py:11-38 for a
// batch of centre nodes, on the device — with cost that follows the EGO NETS, not the graph.
//
// The reference loops over every node in Python (nx.ego_graph + nx.relabel_nodes), a ~60x data
// blow-up built on one CPU thread.  Rounds 1-3 kept one N-bit bitmap per centre (four [B, N/32]
// word arrays, every sweep scanning all of them): B * N / 2 bytes — 20 GB at N = 10^7, B = 4096.
// This version holds only the members:
//
//   * a member is a 64-bit key  (ego c) << 33 | (original node u) << 1 | fresh-bit ; the member
//     list is ONE array sorted by (c, u) — which is the id order of transform.py:24-36: egos one
//     after another, inside an ego ascending original id;
//   * one level of the breadth-first expansion: the members added by the previous level (fresh
//     bit set) push their neighbour lists as candidate keys (one thread per candidate: balanced
//     whatever the degrees, a 10^4-neighbour hub is 10^4 threads), members + candidates are
//     radix-sorted on the key bits in use, and a unique pass keeps the first of every (c, u) run —
//     an old member sorts in front of a candidate with the same (c, u), so the survivor's fresh
//     bit says whether it is new.  `radius` such levels (transform.py:19: nx.ego_graph(G, i,
//     radius)); radius > 4 takes every node of the graph (transform.py:17-18) — the member list
//     is then written directly;
//   * new ids: the centre of ego c keeps id c (0..B-1); the member at sorted position p of ego c
//     gets B + p - c - [u > centre_c]: egos laid out one after another, fresh ids ascending with
//     the original id (transform.py:27-33; the order of fresh ids INSIDE an ego is the iteration
//     order of a Python set in the reference, i.e. not defined by it);
//   * induced edges: a workgroup per (ego, chunk of its members) keeps the ego's membership test
//     in LDS (a hash filter in front of the sorted member list) and walks its members' neighbour
//     lists, a wave per member, the next member's neighbour ids requested one member ahead.
//     Pass 1 counts, an exclusive scan places every member's edges (ordered: by destination
//     member, then by source — no atomics, same output every run), pass 2 emits COO in the new
//     ids together with `orig` (new id -> original id) and `ego_of`.
//
// Memory: every buffer is sized by members + candidates of the level at hand; the caller's
// allocator hands them out (mp_alloc_fn: torch's caching allocator through ctypes) and takes them
// back as soon as a level is done.  Peak scratch is ~45 B per emitted node; nothing scales with N.
// The host reads three counters per level (one stream synchronisation each): sizes are data.
#include "common.h"

#include <cstdlib>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace mp {

__device__ __host__ __forceinline__ uint64_t ego_key(uint64_t c, uint64_t u, uint64_t fresh) {
  return (c << 33) | (u << 1) | fresh;
}
__device__ __forceinline__ uint32_t key_c(uint64_t k) { return (uint32_t)(k >> 33); }
__device__ __forceinline__ uint32_t key_u(uint64_t k) { return (uint32_t)(k >> 1) & 0x7fffffffu; }

// degree of member i if it was added by the previous level, else 0; 0 past the device-side count
struct FreshDegree {
  const uint64_t* keys;
  const int32_t* rowptr;
  const int64_t* count;
  __device__ int64_t operator()(int64_t i) const {
    if (i >= *count) return 0;
    const uint64_t k = keys[i];
    if (!(k & 1ull)) return 0;
    const uint32_t u = key_u(k);
    return (int64_t)(rowptr[u + 1] - rowptr[u]);
  }
};

struct SameMember {   // equality of (c, u), the fresh bit aside
  __device__ bool operator()(uint64_t a, uint64_t b) const { return (a >> 1) == (b >> 1); }
};

__global__ __launch_bounds__(kBlock) void ego_seed_kernel(const int64_t* __restrict__ centres, int64_t B,
                                                          uint64_t* members, int64_t* count) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < B) members[c] = ego_key((uint64_t)c, (uint64_t)centres[c], 1);
  if (c == 0) *count = B;
}

// radius > 4: every node of the graph is a member of every ego (transform.py:17-18)
__global__ __launch_bounds__(kBlock) void ego_whole_kernel(int64_t B, int64_t N, uint64_t* members) {
  const int64_t total = B * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i / N;
    members[i] = ego_key((uint64_t)c, (uint64_t)(i - c * N), 0);
  }
}

// keys_in[0, M) = the members with the fresh bit cleared; keys_in[M + q] = candidate q: neighbour j of the fresh member i
// with offs[i] <= q < offs[i + 1] (one thread per candidate, i by binary search)
__global__ __launch_bounds__(kBlock) void ego_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          const uint64_t* __restrict__ members, int64_t M,
                                                          const int64_t* __restrict__ offs, int64_t C,
                                                          uint64_t* __restrict__ keys_in) {
  const int64_t total = M + C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    if (t < M) {
      keys_in[t] = members[t] & ~1ull;
      continue;
    }
    const int64_t q = t - M;
    int64_t lo = 0, hi = M;                 // last i with offs[i] <= q
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (offs[mid] <= q) lo = mid; else hi = mid;
    }
    const uint64_t k = members[lo];
    const uint32_t u = key_u(k);
    const uint32_t nb = (uint32_t)col[rowptr[u] + (int32_t)(q - offs[lo])] & 0x7fffffffu;
    keys_in[t] = ego_key(key_c(k), nb, 1);
  }
}

// seg[c] = first member of ego c (c = 0..B; seg[B] = M)
__global__ __launch_bounds__(kBlock) void ego_seg_kernel(const uint64_t* __restrict__ members, int64_t M, int64_t B,
                                                         int64_t* __restrict__ seg) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > B) return;
  const uint64_t want = ego_key((uint64_t)c, 0, 0);
  int64_t lo = 0, hi = M;                   // first i with members[i] >= want
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (members[mid] < want) lo = mid + 1; else hi = mid;
  }
  seg[c] = lo;
}

// the member ids as 32-bit words: what the edge passes read
__global__ __launch_bounds__(kBlock) void ego_index_kernel(const uint64_t* __restrict__ members, int64_t M,
                                                           uint32_t* __restrict__ mu) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x)
    mu[i] = key_u(members[i]);
}

// radius > 4 (every node of the graph is a member of every ego: transform.py:17-18): a wave per member copies the
// member's row — every neighbour is a member, its position is c N + u.  PASS 0: cnt[p] = the row's length;
// PASS 1: the entries at eoff[p]..., orig / ego_of of p's new id.  (The edge list is in member order here; the CSR-order
// emission and the CSR output belong to the chunk kernel below.)
template <int PASS>
__global__ __launch_bounds__(kBlock) void ego_edges_whole_kernel(const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col,
                                                                 const int64_t* __restrict__ centres, int64_t B, int64_t N,
                                                                 const uint64_t* __restrict__ members, int64_t M,
                                                                 int32_t* __restrict__ cnt, const int64_t* __restrict__ eoff,
                                                                 int64_t* __restrict__ out_src, int64_t* __restrict__ out_dst,
                                                                 int64_t* __restrict__ orig, int32_t* __restrict__ ego_of) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t p = wave0; p < M; p += nwaves) {
    const uint64_t k = members[p];
    const uint32_t c = key_c(k), v = key_u(k);
    const int64_t centre = centres[c];
    const int64_t s0 = (int64_t)c * N;
    auto new_id = [&](int64_t q, uint32_t u) -> int64_t {
      if ((int64_t)u == centre) return (int64_t)c;
      return B + q - (int64_t)c - ((int64_t)u > centre ? 1 : 0);
    };
    const int rs = rowptr[v], re = rowptr[v + 1];
    if (!PASS) {
      if (lane == 0) cnt[p] = re - rs;
      continue;
    }
    const int64_t vid = new_id(p, v), base = eoff[p];
    if (lane == 0) {
      orig[vid] = (int64_t)v;
      if (ego_of) ego_of[vid] = (int32_t)c;
    }
    for (int j = rs + lane; j < re; j += kWave) {
      const uint32_t u = (uint32_t)col[j] & 0x7fffffffu;
      out_dst[base + (j - rs)] = vid;         // row v holds v's in-edges
      out_src[base + (j - rs)] = new_id(s0 + u, u);
    }
  }
}

// ---- the induced edges of the ego nets, one workgroup per (ego, chunk of its members) -----------------------------------
// This round's first form — a wave per member testing every neighbour against a hash filter in L2, then a binary search
// in the ego's segment — made 1.1 * 10^8 random line reads per pass at 4096 centres of a 10^7-node graph (4.2 ms per pass,
// latency-bound: member key -> row starts -> neighbour ids -> filter word, one dependent round trip each).  An ego net is
// small (hundreds of members): its membership test belongs in LDS.  A workgroup takes up to kEgoChunk consecutive members of ONE ego and
//   * builds the ego's membership table in LDS — original id -> position, open addressing over kEgoTable slots — when
//     the ego has at most kEgoList members (larger egos: every stride-th member in LDS, the search ends in global memory);
//   * loads the row starts / ends of ITS members into LDS with all gathers in flight at once;
//   * then every wave walks members (wave w: w, w + waves, ...): the neighbour ids of the next member are requested
//     before the current one is tested; a test is one LDS read for a non-member, two for a member.
// Order and numbering are the wave-per-member kernel's (edges by destination member, then by source): same output.
constexpr int kEgoChunkMax = 512;
constexpr int kEgoList = 4096;             // LDS words for the ego's sorted member list (or, for a larger ego, a sample of it)
constexpr int kEgoTable = 8192;            // slots of the LDS hash table (load <= 0.5)
constexpr int kEgoBlock = 512;
constexpr uint32_t kEgoEmpty = 0xffffffffu; // (node ids are < 2^31)
constexpr int kEgoRec = 4;                 // hits of a row pass 0 records (16-bit positions): pass 1 of such a row is a copy
constexpr int kEgoHeavyBit = 1 << 30;      // bit 30 of a row's count: more hits than the record holds

struct ChunkCount {   // chunks of ego c (0 for c == B: the scan's total lands there)
  const int64_t* seg;
  int64_t B;
  int64_t chunk;
  __device__ int64_t operator()(int64_t c) const {
    return c < B ? (seg[c + 1] - seg[c] + chunk - 1) / chunk : 0;
  }
};

__device__ __forceinline__ uint32_t ego_hash1(uint32_t u) {   // multiplicative hashing: the top 13 bits of u * golden ratio
  static_assert(kEgoTable == 1 << 13, "ego_hash1 returns 13 bits");
  return (u * 0x9E3779B1u) >> 19;
}

// Output order: rows by NEW id (centres 0..B-1 first, then every ego's other members), inside a row by new source id —
// the order of the engine's CSR (csr_build.hip sorts by (row, col)), so the COO list written here IS the CSR's entry
// list and, with `csr_col` given, the CSR itself is written beside it (int32 col / eid; the scan of the per-row counts
// is its rowptr): a batch needs no sort of its own.  The centre of an ego has the smallest id of its row but sits in
// the middle of the ascending-original-id walk: pass 0 notes whether a row meets it (bit 31 of the count), pass 1 then
// shifts the entries in front of it by one.  LOOPS: a self entry (r, r) per row at its sorted place — the TF path's
// add_self_loop (sparse_adj.py:58-63) — in the CSR only (eid = -1 - r, as mp_csr_from_coo marks inserted loops); it
// needs a base graph without explicit self loops (the caller checks).
template <int PASS, int kEgoChunk, bool LOOPS>
__global__ __launch_bounds__(kEgoBlock) void ego_edges_chunk_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int64_t* __restrict__ centres, int64_t B,
    const uint32_t* __restrict__ mu, const int64_t* __restrict__ seg, const int64_t* __restrict__ chunk_off,
    const int32_t* __restrict__ wg_ego, int32_t* __restrict__ cnt, const int64_t* __restrict__ eoff,
    int64_t* __restrict__ out_src, int64_t* __restrict__ out_dst, int64_t* __restrict__ orig,
    int32_t* __restrict__ ego_of, int32_t* __restrict__ csr_col, int32_t* __restrict__ csr_eid,
    uint16_t* __restrict__ rec, int32_t* __restrict__ qc, int32_t* __restrict__ wg_heavy) {
  // membership of the ego in LDS: an open-addressing table keyed by original id (tab[h] = id, kEgoEmpty = free; the
  // member's position sits in pos_t[h]) for egos of at most kEgoList members — a miss is one LDS read, a hit two; a
  // larger ego keeps every stride-th member in list[] and finishes its searches in global memory
  __shared__ uint32_t tab[kEgoTable];         // hashed egos: the member's original id (kEgoEmpty = free slot)
  __shared__ uint32_t list[kEgoList];         // hashed egos: uint16 positions of the slots (2 per word); else the sample
  uint16_t* const pos_t = reinterpret_cast<uint16_t*>(list);
  static_assert(kEgoTable * 2 <= kEgoList * 4, "the positions of the table's slots fit the list's words");
  __shared__ int32_t rs_l[kEgoChunk], re_l[kEgoChunk];
  __shared__ uint32_t v_l[kEgoChunk];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int kWaves = kEgoBlock / kWave;
  const int64_t wg = blockIdx.x;
  if (wg >= chunk_off[B]) return;             // (the grid is an upper bound: M / chunk + B)
  if (PASS && wg_heavy[wg] == 0) return;      // pass 1 here is for the members pass 0 could not record (see kEgoRec)
  for (int i = tid; i < kEgoTable; i += kEgoBlock) tab[i] = kEgoEmpty;
  __syncthreads();
  const int64_t c = wg_ego[wg];               // (ego_wg_kernel: the ego whose chunk range holds wg)
  const int64_t s0 = seg[c], s1 = seg[c + 1];
  const int S = (int)(s1 - s0);               // (members of one ego: < 2^31, the graph has fewer nodes)
  const bool hashed = S <= kEgoList;
  const int stride = (S + kEgoList - 1) / kEgoList;
  const int n_s = (S + stride - 1) / stride;
  const int64_t m0 = s0 + (wg - chunk_off[c]) * kEgoChunk;
  const int nm = (int)((s1 - m0) < kEgoChunk ? (s1 - m0) : kEgoChunk);
  const int64_t centre = centres[c];
  const uint32_t* __restrict__ mu_c = mu + s0;
  if (hashed) {
    for (int i = tid; i < S; i += kEgoBlock) {
      const uint32_t u = mu_c[i];
      uint32_t h = ego_hash1(u);
      while (atomicCAS(&tab[h], kEgoEmpty, u) != kEgoEmpty) h = (h + 1u) & (uint32_t)(kEgoTable - 1);
      pos_t[h] = (uint16_t)i;
    }
  } else {
    for (int i = tid; i < n_s; i += kEgoBlock) list[i] = mu_c[(int64_t)i * stride];
  }
  __shared__ int32_t cw_l[PASS ? kEgoChunk : 1];     // pass 1: the members' counts (flags) and row offsets, fetched by the
  __shared__ int64_t base_l[PASS ? kEgoChunk : 1];   // whole workgroup at once instead of one dependent load per member
  for (int i = tid; i < nm; i += kEgoBlock) {
    const uint32_t v = mu[m0 + i];
    v_l[i] = v;
    rs_l[i] = rowptr[v];
    re_l[i] = rowptr[v + 1];
    if (PASS) {
      const int64_t vid = (int64_t)v == centre ? c : B + (m0 + i) - c - ((int64_t)v > centre ? 1 : 0);
      cw_l[i] = cnt[vid];
      base_l[i] = eoff[vid];
    }
  }
  __syncthreads();
  auto new_id = [&](int q, uint32_t u) -> int64_t {          // q: position inside the ego
    if ((int64_t)u == centre) return c;
    return B + (s0 + q) - c - ((int64_t)u > centre ? 1 : 0);
  };
  // position of u among the ego's members, or -1
  auto find = [&](uint32_t u) -> int {
    if (hashed) {                                             // one LDS read decides a miss; a hit reads its position
      uint32_t h = ego_hash1(u);
      for (;;) {
        const uint32_t t = tab[h];
        if (t == kEgoEmpty) return -1;
        if (t == u) return (int)pos_t[h];
        h = (h + 1u) & (uint32_t)(kEgoTable - 1);
      }
    }
    int lo = 0, hi = n_s;                                     // first sample > u
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (list[mid] <= u) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) return -1;                                   // below the ego's smallest member
    int a = (lo - 1) * stride, b = a + stride < S ? a + stride : S;    // mu_c[a] <= u < next sample
    while (a < b) {
      const int mid = (a + b) >> 1;
      if (mu_c[mid] < u) a = mid + 1; else b = mid;
    }
    return (a < S && mu_c[a] == u) ? a : -1;
  };
  if (!PASS && wg == chunk_off[c] && tid == 0) qc[c] = find((uint32_t)centre);   // the centre's position in its ego
  __shared__ int heavy_s;
  if (!PASS && tid == 0) heavy_s = 0;
  if (!PASS) __syncthreads();
  const bool can_rec = S <= 65535;            // positions fit the 16-bit records
  // neighbour ids are requested one 64-entry batch ahead: the next batch of this member's row, or the first batch of the
  // wave's next member
  auto fetch = [&](int i, int j0) -> uint32_t {
    if (i >= nm) return 0u;
    const int j = j0 + lane;
    return j < re_l[i] ? ((uint32_t)col[j] & 0x7fffffffu) : 0u;
  };
  // One member: its row's neighbour ids in batches of 64 (the next batch of a long row is requested one ahead).
  // What the wave shares — row bounds, the member's ids, counters, offsets — is made SCALAR (readfirstlane: the scalar unit
  // issues beside the vector one), masks are counted on the scalar unit, the first probe of the table is inline and
  // everything behind it sits under wave-uniform branches that a batch without hits or collisions skips.  (Ablations of
  // this pass at 4 096 centres of BA(2 * 10^6, 5), round 4: whole pass 1.35 ms; without the member loop 0.10; without the
  // membership test 0.59; without the neighbour-id loads 1.05; without both 0.38.  Prefetch depth 1 -> 4, an LDS stage of
  // the chunk's neighbour ids, a keyed table, a deal of the members by work: 1.31-1.35 each; this scalar form: 1.19.)
  const uint32_t centre32 = (uint32_t)centre;
  const int64_t idbase = B + s0 - c;          // new id of the member at position q (not the centre): idbase + q - [q > qc]
  auto process = [&](int i, uint32_t u_first) {
    const int rs = __builtin_amdgcn_readfirstlane(rs_l[i]), re = __builtin_amdgcn_readfirstlane(re_l[i]);
    const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v_l[i]);
    const int qv = (int)(m0 - s0) + i;
    const int64_t p = m0 + i;
    const int64_t vid = v == centre32 ? c : idbase + qv - (v > centre32 ? 1 : 0);
    int64_t base = 0;
    int has_c = 0;
    if (PASS) {
      const int cw = __builtin_amdgcn_readfirstlane(cw_l[i]);
      if (!(cw & kEgoHeavyBit)) return;       // recorded in pass 0: ego_emit_light_kernel writes it
      const int64_t bl = base_l[i];
      base = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bl >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)bl));
      has_c = (cw >> 31) & 1;
      if (lane == 0) {
        orig[vid] = (int64_t)v;
        if (ego_of) ego_of[vid] = (int32_t)c;
      }
    }
    int total = 0, below = 0;                 // non-centre hits so far; those with a smaller original id than v
    bool met_c = false;
    uint32_t u_ahead = u_first;
    for (int j0 = rs; j0 < re; j0 += kWave) {
      const uint32_t u = u_ahead;
      if (j0 + kWave < re) u_ahead = fetch(i, j0 + kWave);
      const bool valid = j0 + lane < re;
      int q = -1;
      if (hashed) {
        uint32_t h = ego_hash1(u);
        const uint32_t t = valid ? tab[h] : kEgoEmpty;
        const bool eq = t == u;
        if (__ballot(t != kEgoEmpty && !eq) != 0ull) {       // some lane met another key: that lane walks on
          uint32_t tt = t;
          while (tt != kEgoEmpty && tt != u) {
            h = (h + 1u) & (uint32_t)(kEgoTable - 1);
            tt = tab[h];
          }
          if (tt == u) q = (int)pos_t[h];
        } else if (__ballot(eq) != 0ull) {
          if (eq) q = (int)pos_t[h];
        }
      } else if (valid) {
        q = find(u);
      }
      const unsigned long long ma = __ballot(q >= 0);
      if (ma == 0ull) continue;               // no neighbour of this batch is a member
      const bool is_c = q >= 0 && u == centre32;
      const bool hit = q >= 0 && !is_c;
      const unsigned long long m = __ballot(hit);
      const unsigned long long mb = __ballot(hit && u < v);
      const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (!PASS && can_rec && q >= 0) {       // the row's first kEgoRec hits (the centre among them), in walk order
        const int k = total + (met_c ? 1 : 0) +
                      (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
        if (k < kEgoRec) rec[p * kEgoRec + k] = (uint16_t)q;
      }
      met_c = met_c || __ballot(is_c) != 0ull;
      if (PASS && (hit || is_c)) {
        // slot in the row: the centre first, then the other sources by new id (= by original id), the row's own self
        // entry (LOOPS) between the smaller and the larger ones; a centre's row holds only larger ones
        const bool after_self = LOOPS && !is_c && (vid < B || u > v);   // (the centre's id is below every row's)
        const int slot = is_c ? 0 : has_c + total + before + (after_self ? 1 : 0);
        const int64_t hit_id = is_c ? c : idbase + q - (u > centre32 ? 1 : 0);
        const int64_t o = base + slot - (LOOPS ? vid + (after_self ? 1 : 0) : 0);      // position without the self entries
        out_dst[o] = vid;                     // row v holds v's in-edges
        out_src[o] = hit_id;
        if (csr_col) {
          csr_col[base + slot] = (int32_t)hit_id;
          csr_eid[base + slot] = (int32_t)o;
        }
      }
      total += (int)__popcll(m);
      below += (int)__popcll(mb);
    }
    if (!PASS && lane == 0) {
      const bool heavy = !can_rec || total + (met_c ? 1 : 0) > kEgoRec;
      cnt[vid] = (total + (met_c ? 1 : 0) + (LOOPS ? 1 : 0)) | (met_c ? (int)0x80000000 : 0) | (heavy ? kEgoHeavyBit : 0);
      if (heavy) heavy_s = 1;
    }
    if (PASS && LOOPS && csr_col && lane == 0) {
      const int slot = vid < B ? 0 : has_c + below;
      csr_col[base + slot] = (int32_t)vid;
      csr_eid[base + slot] = -1 - (int32_t)vid;
    }
  };
  // The chunk's members are dealt to the waves by WORK, not by count: rows of a scale-free graph differ by three orders of
  // magnitude in length, and a wave that drew two hub rows in a deal by index kept its workgroup (and its LDS) alive
  // long after the other seven had finished.  off_s = exclusive prefix of the rows' batch counts (one wave scans the
  // chunk, two members per lane); wave w takes the members whose prefix starts in its eighth of the total.
  __shared__ int32_t off_s[kEgoChunk + 1];
  static_assert(kEgoChunk == 2 * kWave, "the batch counts are scanned by one wave, two members per lane");
  if (wave == 0) {
    const int i0 = 2 * lane, i1 = 2 * lane + 1;
    const int a = i0 < nm ? (re_l[i0] - rs_l[i0] + kWave - 1) / kWave + 1 : 0;      // (+ 1: a row costs a visit even if empty)
    const int bb = i1 < nm ? (re_l[i1] - rs_l[i1] + kWave - 1) / kWave + 1 : 0;
    int incl = a + bb;
    for (int off = 1; off < kWave; off <<= 1) {
      const int t = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += t;
    }
    const int excl = incl - (a + bb);
    off_s[i0] = excl;
    off_s[i1] = excl + a;
    if (lane == kWave - 1) off_s[kEgoChunk] = incl;
  }
  __syncthreads();
  {
    const int T = off_s[kEgoChunk];
    const int w_lo = (int)((int64_t)T * wave / kWaves), w_hi = (int)((int64_t)T * (wave + 1) / kWaves);
    int lo = 0, hi = nm;                      // first member whose prefix is >= w_lo
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (off_s[mid] < w_lo) lo = mid + 1; else hi = mid;
    }
    for (int i = lo; i < nm && off_s[i] < w_hi; ++i) process(i, fetch(i, rs_l[i]));
  }
  if (!PASS) {
    __syncthreads();
    if (tid == 0) wg_heavy[wg] = heavy_s;
  }
}

// Pass 1 for the recorded members — nearly all of them: a member of the outermost level has its parent and seldom more
// than a neighbour or two inside the ego.  One thread per member writes its row from the record: positions inside the ego
// give the new ids (the centre's position qc[c] tells which one is the centre and which ids lie behind it), the member's
// own position where its self entry goes.  No walk over the neighbour lists, no membership test.
template <bool LOOPS>
__global__ __launch_bounds__(kBlock) void ego_emit_light_kernel(
    const uint64_t* __restrict__ members, int64_t M, int64_t B, const int64_t* __restrict__ seg,
    const int32_t* __restrict__ qc, const int32_t* __restrict__ cnt, const int64_t* __restrict__ eoff,
    const uint16_t* __restrict__ rec, int64_t* __restrict__ out_src, int64_t* __restrict__ out_dst,
    int64_t* __restrict__ orig, int32_t* __restrict__ ego_of, int32_t* __restrict__ csr_col,
    int32_t* __restrict__ csr_eid) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < M; p += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = members[p];
    const int64_t c = key_c(key);
    const uint32_t v = key_u(key);
    const int64_t s0 = seg[c];
    const int qv = (int)(p - s0), qcc = qc[c];
    auto id_of = [&](int q) -> int64_t { return q == qcc ? c : B + s0 + q - c - (q > qcc ? 1 : 0); };
    const int64_t vid = id_of(qv);
    const int cw = cnt[vid];
    if (cw & kEgoHeavyBit) continue;          // the chunk kernel's pass 1 writes it
    orig[vid] = (int64_t)v;
    if (ego_of) ego_of[vid] = (int32_t)c;
    const int has_c = (cw >> 31) & 1;
    const int H = (cw & 0x3fffffff) - (LOOPS ? 1 : 0);      // recorded hits, the centre among them
    const int64_t base = eoff[vid];
    int rank = 0, below = 0;
    for (int k = 0; k < H; ++k) {
      const int q = rec[p * kEgoRec + k];
      const bool is_c = q == qcc;
      const bool after_self = LOOPS && !is_c && (vid < B || q > qv);
      const int slot = is_c ? 0 : has_c + rank + (after_self ? 1 : 0);
      const int64_t hit_id = id_of(q);
      const int64_t o = base + slot - (LOOPS ? vid + (after_self ? 1 : 0) : 0);
      out_dst[o] = vid;
      out_src[o] = hit_id;
      if (csr_col) {
        csr_col[base + slot] = (int32_t)hit_id;
        csr_eid[base + slot] = (int32_t)o;
      }
      if (!is_c) {
        ++rank;
        if (q < qv) ++below;
      }
    }
    if (LOOPS && csr_col) {
      const int slot = vid < B ? 0 : has_c + below;
      csr_col[base + slot] = (int32_t)vid;
      csr_eid[base + slot] = -1 - (int32_t)vid;
    }
  }
}

struct CountAsI64 {   // (bits 31 / 30 of a count: "meets its centre" / "not recorded" flags of the chunk kernel)
  const int32_t* cnt;
  int64_t M;
  __device__ int64_t operator()(int64_t i) const { return i < M ? (int64_t)(cnt[i] & 0x3fffffff) : 0; }
};

// wg_ego[w] = the ego whose chunk range holds workgroup w (last c with chunk_off[c] <= w): one thread per workgroup, once,
// instead of a serial search at the head of every workgroup of both edge passes
__global__ __launch_bounds__(kBlock) void ego_wg_kernel(const int64_t* __restrict__ chunk_off, int64_t B, int64_t n_wg,
                                                        int32_t* __restrict__ wg_ego) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_wg || w >= chunk_off[B]) return;
  int64_t lo = 0, hi = B;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (chunk_off[mid] <= w) lo = mid; else hi = mid;
  }
  wg_ego[w] = (int32_t)lo;
}

__global__ __launch_bounds__(kBlock) void ego_rowptr_kernel(const int64_t* __restrict__ eoff, int64_t n, int32_t* rowptr32) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x)
    rowptr32[i] = (int32_t)eoff[i];
}

// ---- host side ----------------------------------------------------------------------------------------------------
struct EgoMem {           // scratch through the caller's allocator, with the live total tracked
  mp_alloc_fn alloc;
  mp_free_fn release;
  void* user;
  size_t live = 0, peak = 0;
  struct Blk { void* p; size_t n; };
  Blk blks[32];
  int nblk = 0;
  void* take(size_t bytes) {
    if (bytes == 0) bytes = 8;
    void* p = alloc(bytes, MP_EGO_TAG_SCRATCH, user);
    if (!p) return nullptr;
    if (nblk < 32) blks[nblk++] = {p, bytes};
    live += bytes;
    if (live > peak) peak = live;
    return p;
  }
  void give(void* p) {
    if (!p) return;
    for (int i = 0; i < nblk; ++i)
      if (blks[i].p == p) {
        live -= blks[i].n;
        blks[i] = blks[--nblk];
        break;
      }
    if (release) release(p, user);
  }
  void give_all() {
    while (nblk > 0) give(blks[nblk - 1].p);
  }
};

static unsigned bits_for(uint64_t x) {   // bits needed to hold values < x
  unsigned b = 0;
  while (b < 63 && (1ull << b) < x) ++b;
  return b;
}

}  // namespace mp

using namespace mp;

#define EGO_TAKE(var, type, bytes)                                  \
  type* var = reinterpret_cast<type*>(mem.take((size_t)(bytes)));   \
  if (!var) { mem.give_all(); return MP_ERR_WORKSPACE; }
#define EGO_HIP(call)                                               \
  do {                                                              \
    hipError_t _e = (call);                                         \
    if (_e != hipSuccess) {                                         \
      ::mp::set_hip_error(_e, #call);                               \
      mem.give_all();                                               \
      return MP_ERR_HIP;                                            \
    }                                                               \
  } while (0)
#define EGO_LAUNCH_CHECK() EGO_HIP(hipGetLastError())

extern "C" {

int mp_ego_expand(const int32_t* rowptr, const int32_t* col, int64_t N, const int64_t* centres, int64_t n_centres,
                  int32_t radius, int32_t flags, mp_alloc_fn alloc, mp_free_fn release, void* user, mp_ego_result_t* out,
                  mp_stream_t stream) {
  if (!rowptr || !centres || !alloc || !out || N <= 0 || n_centres <= 0 || radius < 0) return MP_ERR_INVALID_ARG;
  if (!col && radius > 0) return MP_ERR_INVALID_ARG;
  if (N >= INT32_MAX || n_centres >= (1ll << 30)) return MP_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  const int64_t B = n_centres;
  const bool whole = radius > 4;   // transform.py:17-18
  EgoMem mem{alloc, release, user};
  *out = mp_ego_result_t{};
  const unsigned end_bit = 33 + bits_for((uint64_t)B);

  EGO_TAKE(count_d, int64_t, 64);             // [0] members, [1] unique count
  int64_t M = B, cap = B, cand_total = 0;
  uint64_t* members = nullptr;
  if (whole) {
    if (B > (int64_t)((1ull << 40) / (uint64_t)N)) { mem.give_all(); return MP_ERR_UNSUPPORTED; }
    M = cap = B * N;
    members = reinterpret_cast<uint64_t*>(mem.take((size_t)M * 8));
    if (!members) { mem.give_all(); return MP_ERR_WORKSPACE; }
    hipLaunchKernelGGL(ego_whole_kernel, dim3(flat_grid(M)), dim3(kBlock), 0, st, B, N, members);
    EGO_LAUNCH_CHECK();
  } else {
    members = reinterpret_cast<uint64_t*>(mem.take((size_t)B * 8));
    if (!members) { mem.give_all(); return MP_ERR_WORKSPACE; }
    hipLaunchKernelGGL(ego_seed_kernel, dim3((unsigned)ceil_div(B, kBlock)), dim3(kBlock), 0, st, centres, B, members,
                       count_d);
    EGO_LAUNCH_CHECK();
    for (int lvl = 0; lvl < radius; ++lvl) {
      // candidates of this level: exclusive scan of the fresh members' degrees; offs[cap] = their number
      EGO_TAKE(offs, int64_t, (size_t)(cap + 1) * 8);
      auto deg_it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0),
                                                     FreshDegree{members, rowptr, count_d});
      size_t tb = 0;
      EGO_HIP(rocprim::exclusive_scan(nullptr, tb, deg_it, offs, (int64_t)0, (size_t)(cap + 1), rocprim::plus<int64_t>(), st));
      EGO_TAKE(tmp, char, tb);
      EGO_HIP(rocprim::exclusive_scan(tmp, tb, deg_it, offs, (int64_t)0, (size_t)(cap + 1), rocprim::plus<int64_t>(), st));
      int64_t host[2] = {0, 0};
      EGO_HIP(hipMemcpyAsync(&host[0], count_d, 8, hipMemcpyDeviceToHost, st));
      EGO_HIP(hipMemcpyAsync(&host[1], offs + cap, 8, hipMemcpyDeviceToHost, st));
      EGO_HIP(hipStreamSynchronize(st));
      mem.give(tmp);
      M = host[0];
      const int64_t C = host[1];
      if (C == 0) {            // nothing to add: the expansion has reached its components' ends
        mem.give(offs);
        break;
      }
      if (M + C >= (1ll << 40)) { mem.give_all(); return MP_ERR_UNSUPPORTED; }
      cand_total += C;
      const int64_t T = M + C;
      EGO_TAKE(keys_in, uint64_t, (size_t)T * 8);
      hipLaunchKernelGGL(ego_fill_kernel, dim3(flat_grid(T)), dim3(kBlock), 0, st, rowptr, col, members, M, offs, C, keys_in);
      EGO_LAUNCH_CHECK();
      EGO_TAKE(keys_out, uint64_t, (size_t)T * 8);
      tb = 0;
      EGO_HIP(rocprim::radix_sort_keys(nullptr, tb, keys_in, keys_out, (size_t)T, 0u, end_bit, st));
      EGO_TAKE(tmp2, char, tb);
      EGO_HIP(rocprim::radix_sort_keys(tmp2, tb, keys_in, keys_out, (size_t)T, 0u, end_bit, st));
      // (stream order: the buffers go back to the caller's stream-ordered allocator only after the work that reads
      // them has been enqueued on the same stream)
      mem.give(tmp2);
      mem.give(keys_in);
      mem.give(offs);
      mem.give(members);
      members = keys_in = nullptr;
      EGO_TAKE(next, uint64_t, (size_t)T * 8);
      tb = 0;
      EGO_HIP(rocprim::unique(nullptr, tb, keys_out, next, count_d, (size_t)T, SameMember(), st));
      EGO_TAKE(tmp3, char, tb);
      EGO_HIP(rocprim::unique(tmp3, tb, keys_out, next, count_d, (size_t)T, SameMember(), st));
      mem.give(tmp3);
      mem.give(keys_out);
      members = next;
      cap = T;
    }
    EGO_HIP(hipMemcpyAsync(&M, count_d, 8, hipMemcpyDeviceToHost, st));
    EGO_HIP(hipStreamSynchronize(st));
  }

  // ---- induced edges ----
  const bool want_csr = (flags & MP_EGO_CSR) != 0 && !whole;
  const bool loops = want_csr && (flags & MP_EGO_CSR_SELF_LOOPS) != 0;
  EGO_TAKE(mu, uint32_t, (size_t)M * 4);
  hipLaunchKernelGGL(ego_index_kernel, dim3(flat_grid(M)), dim3(kBlock), 0, st, members, M, mu);
  EGO_LAUNCH_CHECK();
  EGO_TAKE(seg, int64_t, (size_t)(B + 1) * 8);
  hipLaunchKernelGGL(ego_seg_kernel, dim3((unsigned)ceil_div(B + 1, kBlock)), dim3(kBlock), 0, st, members, M, B, seg);
  EGO_LAUNCH_CHECK();
  EGO_TAKE(cnt, int32_t, (size_t)M * 4);
  EGO_TAKE(eoff, int64_t, (size_t)(M + 1) * 8);
  const dim3 egrid((unsigned)(ceil_div(M, kWavesPerBlock) < kNumCU * 16 ? ceil_div(M, kWavesPerBlock) : kNumCU * 16));
  // one workgroup per (ego, chunk of its members): chunk_off[c] = chunks of the egos before c; the grid is the bound
  // M / chunk + B, workgroups past chunk_off[B] leave at once
  constexpr int kChunk = 128;
  int64_t* chunk_off = nullptr;
  int32_t* wg_ego = nullptr;
  const int64_t n_wg = M / kChunk + B;
  const dim3 cgrid((unsigned)n_wg);
  if (!whole) {
    chunk_off = reinterpret_cast<int64_t*>(mem.take((size_t)(B + 1) * 8));
    wg_ego = reinterpret_cast<int32_t*>(mem.take((size_t)n_wg * 4));
    if (!chunk_off || !wg_ego) { mem.give_all(); return MP_ERR_WORKSPACE; }
    auto ch_it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), ChunkCount{seg, B, (int64_t)kChunk});
    size_t tb = 0;
    EGO_HIP(rocprim::exclusive_scan(nullptr, tb, ch_it, chunk_off, (int64_t)0, (size_t)(B + 1), rocprim::plus<int64_t>(), st));
    EGO_TAKE(tmp, char, tb);
    EGO_HIP(rocprim::exclusive_scan(tmp, tb, ch_it, chunk_off, (int64_t)0, (size_t)(B + 1), rocprim::plus<int64_t>(), st));
    mem.give(tmp);
    hipLaunchKernelGGL(ego_wg_kernel, dim3((unsigned)ceil_div(n_wg, kBlock)), dim3(kBlock), 0, st, chunk_off, B, n_wg, wg_ego);
    EGO_LAUNCH_CHECK();
  }
  uint16_t* rec = nullptr;
  int32_t *qc = nullptr, *wg_heavy = nullptr;
  if (!whole) {
    rec = reinterpret_cast<uint16_t*>(mem.take((size_t)M * kEgoRec * 2));
    qc = reinterpret_cast<int32_t*>(mem.take((size_t)B * 4));
    wg_heavy = reinterpret_cast<int32_t*>(mem.take((size_t)n_wg * 4));
    if (!rec || !qc || !wg_heavy) { mem.give_all(); return MP_ERR_WORKSPACE; }
  }
#define EGO_CHUNK_LAUNCH(PASS, LOOPS, ...) \
  hipLaunchKernelGGL((ego_edges_chunk_kernel<PASS, kChunk, LOOPS>), cgrid, dim3(kEgoBlock), 0, st, rowptr, col, centres, B, \
                     mu, seg, chunk_off, wg_ego, __VA_ARGS__, rec, qc, wg_heavy)
  if (whole)
    hipLaunchKernelGGL((ego_edges_whole_kernel<0>), egrid, dim3(kBlock), 0, st, rowptr, col, centres, B, N, members, M, cnt,
                       eoff, nullptr, nullptr, nullptr, nullptr);
  else if (loops)
    EGO_CHUNK_LAUNCH(0, true, cnt, eoff, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr, (int32_t*)nullptr,
                     (int32_t*)nullptr, (int32_t*)nullptr);
  else
    EGO_CHUNK_LAUNCH(0, false, cnt, eoff, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr, (int32_t*)nullptr,
                     (int32_t*)nullptr, (int32_t*)nullptr);
  EGO_LAUNCH_CHECK();
  {
    auto cnt_it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), CountAsI64{cnt, M});
    size_t tb = 0;
    EGO_HIP(rocprim::exclusive_scan(nullptr, tb, cnt_it, eoff, (int64_t)0, (size_t)(M + 1), rocprim::plus<int64_t>(), st));
    EGO_TAKE(tmp, char, tb);
    EGO_HIP(rocprim::exclusive_scan(tmp, tb, cnt_it, eoff, (int64_t)0, (size_t)(M + 1), rocprim::plus<int64_t>(), st));
    mem.give(tmp);
  }
  int64_t nnz = 0;                             // entries incl. the self entries of the CSR (LOOPS)
  EGO_HIP(hipMemcpyAsync(&nnz, eoff + M, 8, hipMemcpyDeviceToHost, st));
  EGO_HIP(hipStreamSynchronize(st));
  const int64_t E = loops ? nnz - M : nnz;
  if (want_csr && nnz >= INT32_MAX) { mem.give_all(); return MP_ERR_UNSUPPORTED; }

  // ---- outputs (the caller's, tagged) ----
  // (one block for both rows of the COO list — PyG's edge_index [2, E] without a copy: dst = src + E)
  int64_t* o_src = reinterpret_cast<int64_t*>(alloc((size_t)(E > 0 ? 2 * E : 2) * 8, MP_EGO_TAG_EDGES, user));
  int64_t* o_dst = o_src ? o_src + E : nullptr;
  int64_t* o_orig = reinterpret_cast<int64_t*>(alloc((size_t)M * 8, MP_EGO_TAG_ORIG, user));
  int32_t* o_ego = reinterpret_cast<int32_t*>(alloc((size_t)M * 4, MP_EGO_TAG_EGO_OF, user));
  if (!o_src || !o_dst || !o_orig || !o_ego) { mem.give_all(); return MP_ERR_WORKSPACE; }
  int32_t *o_rp = nullptr, *o_col = nullptr, *o_eid = nullptr;
  if (want_csr) {
    o_rp = reinterpret_cast<int32_t*>(alloc((size_t)(M + 1) * 4, MP_EGO_TAG_ROWPTR, user));
    o_col = reinterpret_cast<int32_t*>(alloc((size_t)(nnz > 0 ? nnz : 1) * 4, MP_EGO_TAG_COL, user));
    o_eid = reinterpret_cast<int32_t*>(alloc((size_t)(nnz > 0 ? nnz : 1) * 4, MP_EGO_TAG_EID, user));
    if (!o_rp || !o_col || !o_eid) { mem.give_all(); return MP_ERR_WORKSPACE; }
    hipLaunchKernelGGL(ego_rowptr_kernel, dim3(flat_grid(M + 1)), dim3(kBlock), 0, st, eoff, M, o_rp);
    EGO_LAUNCH_CHECK();
  }
  if (whole)
    hipLaunchKernelGGL((ego_edges_whole_kernel<1>), egrid, dim3(kBlock), 0, st, rowptr, col, centres, B, N, members, M,
                       nullptr, eoff, o_src, o_dst, o_orig, o_ego);
  else if (loops) {
    hipLaunchKernelGGL((ego_emit_light_kernel<true>), dim3(flat_grid(M)), dim3(kBlock), 0, st, members, M, B, seg, qc, cnt,
                       eoff, rec, o_src, o_dst, o_orig, o_ego, o_col, o_eid);
    EGO_CHUNK_LAUNCH(1, true, cnt, eoff, o_src, o_dst, o_orig, o_ego, o_col, o_eid);
  } else {
    hipLaunchKernelGGL((ego_emit_light_kernel<false>), dim3(flat_grid(M)), dim3(kBlock), 0, st, members, M, B, seg, qc, cnt,
                       eoff, rec, o_src, o_dst, o_orig, o_ego, o_col, o_eid);
    EGO_CHUNK_LAUNCH(1, false, cnt, eoff, o_src, o_dst, o_orig, o_ego, o_col, o_eid);
  }
  EGO_LAUNCH_CHECK();
#undef EGO_CHUNK_LAUNCH
  out->n_nodes = M;
  out->n_edges = E;
  out->src = o_src;
  out->dst = o_dst;
  out->orig = o_orig;
  out->ego_of = o_ego;
  out->rowptr = o_rp;
  out->col = o_col;
  out->eid = o_eid;
  out->nnz = want_csr ? nnz : 0;
  out->candidates = cand_total;
  out->scratch_peak_bytes = mem.peak;
  mem.give_all();   // (stream-ordered: the emit kernel above is already enqueued)
  return MP_OK;
}

}  // extern "C"

// Placement-aware device arena: where the large matrices of the aggregation path live.
//
// Background (profiles/r01_placement.log, profiles/r02_placement_map.log): on MI355X an
// HBM-bound launch that reads matrix X and writes matrix Y runs up to ~15 % slower depending on which
// physical memory the two are backed by.  A timed copy between 1 GiB chunks of one 160 GiB slab shows a
// block structure (blocks of 8-16 GiB; pairs of blocks cost +0-3 %, +6-9 % or +10-17 %): the high address
// bits are hashed into the DRAM bank / channel selection, and a read stream and a write stream that hash
// alike pay bus turnarounds.  Which physical memory a separately allocated tensor gets is the driver's
// choice, so the engine owns one slab per device, measures the pairwise cost between its granules once
// (graphgym_amd/placement.py with mp_probe_copy_ms) and places every large output where it conflicts least
// with the matrices the launch reads (mp_arena_alloc_placed).  Buffers are handed to torch as DLPack
// tensors whose deleter returns them to the arena, so they have ordinary tensor lifetime.
//
// No counterpart in the reference (it never places anything): this is part of the path's data layout in
// HBM.  One arena per device; all entry points act on the calling thread's current HIP device.
#include "common.h"
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace mp {

constexpr size_t kArenaAlign = 2u << 20;   // 2 MiB: the granule of HBM page mappings
constexpr int kMaxDev = 16;

struct Arena {
  char* base = nullptr;
  size_t bytes = 0;
  std::map<size_t, size_t> free_list;   // offset -> bytes, coalesced
  std::map<size_t, size_t> live;        // offset -> bytes
  size_t in_use = 0;
};

static Arena g_arena[kMaxDev];
static std::mutex g_arena_mu;

static int cur_dev() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) return -1;
  return d;
}

// mean penalty of the byte range [off, off + need) under a per-granule penalty vector
static double range_cost(size_t off, size_t need, const float* pen, int32_t n_gran, size_t gran) {
  if (!pen || n_gran <= 0 || gran == 0) return 0.0;
  double acc = 0.0;
  size_t g = off / gran;
  size_t pos = off;
  const size_t end = off + need;
  while (pos < end) {
    const size_t g_end = (g + 1) * gran;
    const size_t stop = g_end < end ? g_end : end;
    const float p = g < (size_t)n_gran ? pen[g] : pen[n_gran - 1];
    acc += (double)(stop - pos) * (double)p;
    pos = stop;
    ++g;
  }
  return acc / (double)need;
}

static void arena_release_locked(Arena& a, size_t off) {
  auto it = a.live.find(off);
  if (it == a.live.end()) return;
  const size_t bytes = it->second;
  a.live.erase(it);
  a.in_use -= bytes;
  auto ins = a.free_list.emplace(off, bytes).first;
  auto nxt = std::next(ins);
  if (nxt != a.free_list.end() && ins->first + ins->second == nxt->first) {   // merge with the next run
    ins->second += nxt->second;
    a.free_list.erase(nxt);
  }
  if (ins != a.free_list.begin()) {                                           // and with the previous one
    auto prv = std::prev(ins);
    if (prv->first + prv->second == ins->first) {
      prv->second += ins->second;
      a.free_list.erase(ins);
    }
  }
}

typedef float ap_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBlock) void arena_copy_kernel(const ap_f32x4* __restrict__ src,
                                                            ap_f32x4* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(src[i], dst + i);
}

// ---- DLPack (dlpack.h v0.8 layout, restated: the struct torch.from_dlpack consumes) ----
struct DLDev { int32_t device_type; int32_t device_id; };
struct DLType { uint8_t code; uint8_t bits; uint16_t lanes; };
struct DLTens {
  void* data; DLDev device; int32_t ndim; DLType dtype; int64_t* shape; int64_t* strides; uint64_t byte_offset;
};
struct DLManaged {
  DLTens dl_tensor;
  void* manager_ctx;
  void (*deleter)(DLManaged*);
};
constexpr int32_t kDLROCM = 10;

struct ManagedBlock {
  DLManaged m;
  int64_t shape[4];
  int32_t device;
};

static void managed_deleter(DLManaged* self) {
  if (!self) return;
  ManagedBlock* b = reinterpret_cast<ManagedBlock*>(self->manager_ctx);
  {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    Arena& a = g_arena[b->device];
    if (a.base) arena_release_locked(a, (size_t)((char*)self->dl_tensor.data - a.base));
  }
  free(b);
}

}  // namespace mp

using namespace mp;

extern "C" {

int mp_arena_create(size_t bytes) {
  const int d = cur_dev();
  if (d < 0 || bytes < kArenaAlign) return MP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g_arena_mu);
  Arena& a = g_arena[d];
  if (a.base) return MP_ERR_INVALID_ARG;   // one arena per device
  bytes = bytes / kArenaAlign * kArenaAlign;
  void* p = nullptr;
  MP_HIP(hipMalloc(&p, bytes));
  a.base = (char*)p;
  a.bytes = bytes;
  a.live.clear();
  a.free_list.clear();
  a.free_list[0] = bytes;
  a.in_use = 0;
  return MP_OK;
}

int mp_arena_destroy(void) {
  const int d = cur_dev();
  if (d < 0) return MP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g_arena_mu);
  Arena& a = g_arena[d];
  if (!a.base) return MP_OK;
  if (!a.live.empty()) return MP_ERR_INVALID_ARG;   // buffers still handed out
  MP_HIP(hipFree(a.base));
  a = Arena();
  return MP_OK;
}

int mp_arena_info(void** base_host, size_t* bytes_host, size_t* in_use_host, size_t* largest_free_host) {
  const int d = cur_dev();
  if (d < 0) return MP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g_arena_mu);
  const Arena& a = g_arena[d];
  if (base_host) *base_host = a.base;
  if (bytes_host) *bytes_host = a.bytes;
  if (in_use_host) *in_use_host = a.in_use;
  if (largest_free_host) {
    size_t m = 0;
    for (auto& kv : a.free_list) m = kv.second > m ? kv.second : m;
    *largest_free_host = m;
  }
  return MP_OK;
}

// A free range of `bytes` (rounded up to 2 MiB) whose mean penalty is smallest: penalty_host[g] prices
// granule g (bytes [g * granule_bytes, (g + 1) * granule_bytes) of the slab); NULL = first fit.  Candidate
// starts inside a free run: its start, its end minus the request, and every granule boundary in between
// (as a start or as an end) — the mean is piecewise linear in the start, so its minimum is at one of them.
// Ties go to the lowest address.  MP_ERR_WORKSPACE when no free run is large enough.
int mp_arena_alloc_placed(size_t bytes, const float* penalty_host, int32_t n_granules, size_t granule_bytes,
                          void** ptr_host) {
  const int d = cur_dev();
  if (d < 0 || !ptr_host || bytes == 0) return MP_ERR_INVALID_ARG;
  if (penalty_host && (n_granules <= 0 || granule_bytes < kArenaAlign || granule_bytes % kArenaAlign))
    return MP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g_arena_mu);
  Arena& a = g_arena[d];
  if (!a.base) return MP_ERR_INVALID_ARG;
  const size_t need = align_up(bytes, kArenaAlign);
  bool found = false;
  size_t best_off = 0, best_run = 0;
  double best_cost = 0.0;
  for (auto& kv : a.free_list) {
    const size_t lo = kv.first, have = kv.second;
    if (have < need) continue;
    const size_t hi = lo + have - need;   // last admissible start
    auto consider = [&](size_t off) {
      if (off < lo || off > hi) return;
      const double c = range_cost(off, need, penalty_host, n_granules, granule_bytes);
      if (!found || c < best_cost - 1e-12) { found = true; best_cost = c; best_off = off; best_run = lo; }
    };
    consider(lo);
    if (penalty_host) {
      const size_t g0 = lo / granule_bytes, g1 = (hi + need) / granule_bytes + 1;
      for (size_t g = g0; g <= g1; ++g) {
        const size_t b = g * granule_bytes;
        consider(b);                           // start on a boundary
        if (b >= need) consider(b - need);     // end on a boundary
      }
      consider(hi);
    } else if (found) {
      break;                                   // first fit
    }
  }
  if (!found) return MP_ERR_WORKSPACE;
  // carve [best_off, best_off + need) out of the run that starts at best_run
  auto it = a.free_list.find(best_run);
  const size_t run_lo = it->first, run_len = it->second;
  a.free_list.erase(it);
  if (best_off > run_lo) a.free_list[run_lo] = best_off - run_lo;
  if (best_off + need < run_lo + run_len) a.free_list[best_off + need] = run_lo + run_len - (best_off + need);
  a.live[best_off] = need;
  a.in_use += need;
  *ptr_host = a.base + best_off;
  return MP_OK;
}

int mp_arena_release(void* ptr) {
  const int d = cur_dev();
  if (d < 0 || !ptr) return MP_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g_arena_mu);
  Arena& a = g_arena[d];
  if (!a.base || (char*)ptr < a.base || (char*)ptr >= a.base + a.bytes) return MP_ERR_INVALID_ARG;
  const size_t off = (size_t)((char*)ptr - a.base);
  if (a.live.find(off) == a.live.end()) return MP_ERR_INVALID_ARG;
  arena_release_locked(a, off);
  return MP_OK;
}

// A DLManagedTensor (DLPack) over an arena buffer: contiguous, 1-4 dimensions, element type given as DLPack
// (code, bits): float = (2, 32), int = (0, 32), uint8 = (1, 8).  The deleter returns the buffer to the arena
// (mp_arena_release) — the caller wraps the result in a PyCapsule named "dltensor" and gives it to
// torch.from_dlpack, which takes ownership.  *managed_host must be consumed exactly once.
int mp_arena_dlpack(void* ptr, int32_t ndim, const int64_t* shape_host, int32_t type_code, int32_t type_bits,
                    void** managed_host) {
  const int d = cur_dev();
  if (d < 0 || !ptr || !shape_host || !managed_host || ndim < 1 || ndim > 4) return MP_ERR_INVALID_ARG;
  {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    Arena& a = g_arena[d];
    if (!a.base || (char*)ptr < a.base || (char*)ptr >= a.base + a.bytes) return MP_ERR_INVALID_ARG;
    if (a.live.find((size_t)((char*)ptr - a.base)) == a.live.end()) return MP_ERR_INVALID_ARG;
  }
  ManagedBlock* b = (ManagedBlock*)calloc(1, sizeof(ManagedBlock));
  if (!b) return MP_ERR_INVALID_ARG;
  for (int i = 0; i < ndim; ++i) b->shape[i] = shape_host[i];
  b->device = d;
  b->m.dl_tensor.data = ptr;
  b->m.dl_tensor.device = {kDLROCM, d};
  b->m.dl_tensor.ndim = ndim;
  b->m.dl_tensor.dtype = {(uint8_t)type_code, (uint8_t)type_bits, 1};
  b->m.dl_tensor.shape = b->shape;
  b->m.dl_tensor.strides = nullptr;
  b->m.dl_tensor.byte_offset = 0;
  b->m.manager_ctx = b;
  b->m.deleter = managed_deleter;
  *managed_host = &b->m;
  return MP_OK;
}

// Timed streaming copy src -> dst of `bytes` (multiple of 16, 16-byte aligned): `reps` launches between two
// events on `stream`, one untimed launch first.  SYNCHRONISES.  *ms_host = mean per launch.  The yardstick
// placement.py uses to find out which parts of memory conflict.
int mp_probe_copy_ms(const void* src, void* dst, size_t bytes, int32_t reps, float* ms_host, mp_stream_t stream) {
  if (!src || !dst || !ms_host || reps < 1 || bytes % 16 || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16))
    return MP_ERR_INVALID_ARG;
  hipStream_t st = as_stream(stream);
  hipEvent_t e0, e1;
  MP_HIP(hipEventCreate(&e0));
  MP_HIP(hipEventCreate(&e1));
  const int64_t n4 = (int64_t)(bytes / 16);
  auto launch = [&]() {
    hipLaunchKernelGGL(arena_copy_kernel, dim3(kNumCU * 8), dim3(kBlock), 0, st,
                       reinterpret_cast<const ap_f32x4*>(src), reinterpret_cast<ap_f32x4*>(dst), n4);
  };
  launch();
  MP_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  MP_HIP(hipEventRecord(e1, st));
  MP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MP_HIP(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  MP_LAUNCH_CHECK();
  *ms_host = ms / (float)reps;
  return MP_OK;
}

}  // extern "C"

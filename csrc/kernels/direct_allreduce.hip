// Direct (all-links) all-reduce over HIP IPC peer buffers for one xGMI node.
//
// MI355X's intra-node fabric is a fully connected mesh: 7 xGMI links per GPU.  A ring collective
// drives one outbound link per GPU; this kernel instead lets every rank read a distinct 1/N shard
// from each of its N-1 peers at once, so all links carry traffic:
//
//   two-shot (large buckets): barrier -> rank r sums shard r of every peer's buffer (fixed peer order,
//             fp32 accumulation), stores it to its own buffer and to the output -> barrier ->
//             rank r copies the reduced shard p from peer p for every p != r.
//   one-shot (small, latency-bound): barrier -> every rank sums all N full buffers itself.
//
// Every rank exports two allocations through hipIpcGetMemHandle: a data region with two slots
// (calls alternate, which removes the end-of-call barrier: a slot is rewritten two calls later, and the
// call in between has a barrier every peer passes only after it finished reading) and a flag region in
// uncached memory.  Barriers are per workgroup: block b of rank r exchanges monotonically increasing
// sequence numbers with block b of every peer; both phases use the same index->block mapping, so the data a
// block reads in phase 2 was written by the block it synchronised with.  Hand-off follows the gfx950
// visibility rules: storing waves drain (vmcnt(0)), workgroup barrier, one system-scope release fence,
// flag stores with system scope; the reader polls with system-scope loads, then one system-scope
// acquire, vmcnt(0), workgroup barrier, plain loads.  Spins are bounded: a rank whose peer never
// arrives sets a status word and leaves, so the grid always drains.
//
// Registered regions: a caller that allocates its buffers with dlamd_dar_alloc (gradient buckets) registers them
// once, and dlamd_dar_allreduce_region reduces them IN PLACE - no staging copy.  Without the second slot the call
// needs one more barrier: a rank may rewrite its buffer (next backward) only after every peer has finished reading
// it, so in-place calls end with a third barrier; the one-shot form holds its sums in registers across a barrier
// (every peer has read) before it stores them.
//
// The reference has no collective of any kind; this is new for DP recognizer training.
#include "common.h"
#include <stdlib.h>
#include <string.h>

#define DAR_MAX_WORLD 8
#define DAR_THREADS 512
#define DAR_FLAG_STRIDE 8   // uint32 words per (phase, block): one per source rank
#define DAR_PHASES 3        // barriers per call: inputs complete, shards reduced, (in place) shards copied
#define DAR_MAX_REGIONS 64

struct DarArgs {
  const char* peer_data[DAR_MAX_WORLD];   // slot base of each rank's data region (own entry = local pointer)
  uint32_t* peer_flags[DAR_MAX_WORLD];
  char* own_data;
  void* out;
  long nvec;                              // 16-byte vectors
  int rank, world, nblocks;
  uint32_t seq;
  uint32_t spin_limit;
  float scale;
  uint32_t* status;
  int inplace;                            // peer_data are the callers' own buffers: out == own_data, end barrier
};

__device__ __forceinline__ void dar_barrier(const DarArgs& a, int phase, bool release) {
  if (release) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int t = threadIdx.x;
  if (t < a.world) {
    if (release) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long slot = ((long)phase * a.nblocks + blockIdx.x) * DAR_FLAG_STRIDE;
    __hip_atomic_store(a.peer_flags[t] + slot + a.rank, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* mine = a.peer_flags[a.rank] + slot + t;
    uint32_t spins = 0;
    while ((int)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.seq) < 0) {
      if (++spins > a.spin_limit) {
        __hip_atomic_store(a.status, (uint32_t)(1 + phase), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

template <bool BF16>
__device__ __forceinline__ void dar_accum(float (&acc)[8], u32x4 v) {
#pragma clang fp reassociate(off)   // rank order is the summation order on every rank (-ffast-math would regroup it)
  if constexpr (BF16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[2 * j] += bf16_bits_to_f32(v[j] & 0xffffu);
      acc[2 * j + 1] += __builtin_bit_cast(float, v[j] & 0xffff0000u);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t w = v[j];   // a bit_cast applied to the vector-element lvalue itself reads element 0
      acc[j] += __builtin_bit_cast(float, w);
    }
  }
}

__device__ __forceinline__ uint32_t dar_f32_to_bf16_bits(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7fffu + ((u >> 16) & 1u);   // round to nearest even (gradients are finite)
  return u >> 16;
}

template <bool BF16>
__device__ __forceinline__ u32x4 dar_pack(const float (&acc)[8], float scale) {
  u32x4 o;
  if constexpr (BF16) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = dar_f32_to_bf16_bits(acc[2 * j] * scale) | (dar_f32_to_bf16_bits(acc[2 * j + 1] * scale) << 16);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __builtin_bit_cast(uint32_t, acc[j] * scale);
  }
  return o;
}

// WORLD > 0: compile-time rank count (peer loads fully unrolled, all in flight); WORLD == 0: runtime loop.
template <bool BF16, int WORLD>
__device__ __forceinline__ u32x4 dar_sum_at(const DarArgs& a, long i) {
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (WORLD > 0) {
    u32x4 v[WORLD];
#pragma unroll
    for (int p = 0; p < WORLD; ++p) v[p] = *(const u32x4*)(a.peer_data[p] + i * 16);
#pragma unroll
    for (int p = 0; p < WORLD; ++p) dar_accum<BF16>(acc, v[p]);
  } else {
    for (int p = 0; p < a.world; ++p) dar_accum<BF16>(acc, *(const u32x4*)(a.peer_data[p] + i * 16));
  }
  return dar_pack<BF16>(acc, a.scale);
}

template <bool BF16, int WORLD, bool ONE_SHOT>
__global__ __launch_bounds__(DAR_THREADS) void dar_allreduce_kernel(DarArgs a) {
  const long stride = (long)gridDim.x * DAR_THREADS;
  const long first = (long)blockIdx.x * DAR_THREADS + threadIdx.x;
  u32x4* out = (u32x4*)a.out;
  dar_barrier(a, 0, true);   // every peer's input slot is complete
  if constexpr (ONE_SHOT) {
    if (a.inplace) {
      // out is the buffer the peers read: sum first, store after every peer has read (host: nvec <= grid threads)
      const bool have = first < a.nvec;
      u32x4 r = {0u, 0u, 0u, 0u};
      if (have) r = dar_sum_at<BF16, WORLD>(a, first);
      dar_barrier(a, 1, true);
      if (have) out[first] = r;
      return;
    }
    for (long i = first; i < a.nvec; i += stride) out[i] = dar_sum_at<BF16, WORLD>(a, i);
  } else {
    const int world = WORLD > 0 ? WORLD : a.world;
    const long per = (a.nvec + world - 1) / world;
    const long lo = (long)a.rank * per;
    const long hi = lo + per < a.nvec ? lo + per : a.nvec;
    for (long i = lo + first; i < hi; i += stride) {
      u32x4 r = dar_sum_at<BF16, WORLD>(a, i);
      *(u32x4*)(a.own_data + i * 16) = r;
      if (!a.inplace) out[i] = r;
    }
    dar_barrier(a, 1, true);   // every peer's reduced shard is stored and released
    for (int s = 1; s < world; ++s) {
      const int p = (a.rank + s) % world;   // rank r starts at peer r+1: spreads the reads over the links
      const long plo = (long)p * per;
      const long phi = plo + per < a.nvec ? plo + per : a.nvec;
      const char* src = a.peer_data[p];
      for (long i = plo + first; i < phi; i += stride) out[i] = *(const u32x4*)(src + i * 16);
    }
    if (a.inplace) dar_barrier(a, 2, true);   // every peer has copied this rank's shard: the buffer may be rewritten
  }
}

struct DarContext {
  int rank = 0, world = 1, blocks = 0, device = 0;
  long max_bytes = 0;
  char* data = nullptr;          // 2 * max_bytes
  uint32_t* flags = nullptr;     // DAR_PHASES * blocks * DAR_FLAG_STRIDE words
  size_t flag_bytes = 0;
  bool flags_uncached = false;
  char* peer_data[DAR_MAX_WORLD] = {};
  uint32_t* peer_flags[DAR_MAX_WORLD] = {};
  bool imported[DAR_MAX_WORLD] = {};
  uint32_t* status = nullptr;    // separate local allocation
  uint32_t seq = 0;
  int nregions = 0;
  char* region[DAR_MAX_REGIONS][DAR_MAX_WORLD] = {};   // registered caller buffers: [id][rank]
  long region_bytes[DAR_MAX_REGIONS] = {};
  uint32_t spin_limit = 64u << 20;   // polls before a barrier gives up on a peer: tens of seconds (ranks skew by whole
                                     // data-loader stalls and checkpoint writes; DLAMD_DAR_SPIN_LIMIT / set_spin_limit override)
};

static size_t dar_flag_bytes(int blocks) {
  size_t n = (size_t)DAR_PHASES * blocks * DAR_FLAG_STRIDE * sizeof(uint32_t);
  return (n + 4095) / 4096 * 4096;
}

extern "C" int dlamd_dar_create(int rank, int world, long max_bytes, int blocks, void** out) {
  if (!out || world < 2 || world > DAR_MAX_WORLD || rank < 0 || rank >= world) return -1;
  if (max_bytes <= 0 || (max_bytes & 15) || blocks < 1 || blocks > 1024) return -2;
  DarContext* c = new DarContext();
  c->rank = rank; c->world = world; c->blocks = blocks; c->max_bytes = max_bytes;
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; return -3; }
  if (hipMalloc((void**)&c->data, (size_t)2 * max_bytes) != hipSuccess) { delete c; return -4; }
  c->flag_bytes = dar_flag_bytes(blocks);
  hipIpcMemHandle_t probe;
  if (hipExtMallocWithFlags((void**)&c->flags, c->flag_bytes, hipDeviceMallocUncached) == hipSuccess &&
      hipIpcGetMemHandle(&probe, c->flags) == hipSuccess) {
    c->flags_uncached = true;
  } else {
    (void)hipGetLastError();
    if (c->flags) (void)hipFree(c->flags);
    c->flags = nullptr;
    if (hipMalloc((void**)&c->flags, c->flag_bytes) != hipSuccess) { (void)hipFree(c->data); delete c; return -5; }
  }
  if (hipMalloc((void**)&c->status, 256) != hipSuccess) { (void)hipFree(c->flags); (void)hipFree(c->data); delete c; return -6; }
  (void)hipMemset(c->flags, 0, c->flag_bytes);
  (void)hipMemset(c->status, 0, 256);
  (void)hipMemset(c->data, 0, (size_t)2 * max_bytes);
  if (hipDeviceSynchronize() != hipSuccess) return -7;
  c->peer_data[rank] = c->data;
  c->peer_flags[rank] = c->flags;
  c->imported[rank] = true;
  if (const char* e = getenv("DLAMD_DAR_SPIN_LIMIT")) {
    long v = atol(e);
    if (v > 0 && v < (1l << 31)) c->spin_limit = (uint32_t)v;
  }
  *out = c;
  return 0;
}

// handles: 2 x hipIpcMemHandle_t (data, flags) = 128 bytes
extern "C" int dlamd_dar_export(void* ctx, void* handles) {
  DarContext* c = (DarContext*)ctx;
  if (!c || !handles) return -1;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  hipIpcMemHandle_t* h = (hipIpcMemHandle_t*)handles;
  if (hipIpcGetMemHandle(&h[0], c->data) != hipSuccess) return -2;
  if (hipIpcGetMemHandle(&h[1], c->flags) != hipSuccess) return -3;
  return 0;
}

extern "C" int dlamd_dar_import(void* ctx, int peer, const void* handles) {
  DarContext* c = (DarContext*)ctx;
  if (!c || !handles || peer < 0 || peer >= c->world || peer == c->rank || c->imported[peer]) return -1;
  hipIpcMemHandle_t h[2];
  memcpy(h, handles, sizeof(h));
  void* d = nullptr;
  void* f = nullptr;
  if (hipIpcOpenMemHandle(&d, h[0], hipIpcMemLazyEnablePeerAccess) != hipSuccess) return -2;
  if (hipIpcOpenMemHandle(&f, h[1], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipIpcCloseMemHandle(d); return -3; }
  c->peer_data[peer] = (char*)d;
  c->peer_flags[peer] = (uint32_t*)f;
  c->imported[peer] = true;
  return 0;
}

extern "C" int dlamd_dar_info(void* ctx, int what) {
  DarContext* c = (DarContext*)ctx;
  if (!c) return -1;
  switch (what) {
    case 0: return c->flags_uncached ? 1 : 0;
    case 1: return (int)c->seq;
    case 2: return c->blocks;
    default: return -1;
  }
}

extern "C" int dlamd_dar_set_spin_limit(void* ctx, long n) {
  DarContext* c = (DarContext*)ctx;
  if (!c || n <= 0 || n >= (1l << 31)) return -1;
  c->spin_limit = (uint32_t)n;
  return 0;
}

// Blocking read of the status word: 0 ok, 1 / 2 / 3 = a peer never reached the first / second / third barrier.
extern "C" int dlamd_dar_status(void* ctx) {
  DarContext* c = (DarContext*)ctx;
  if (!c) return -1;
  uint32_t v = 0;
  if (hipMemcpy(&v, c->status, 4, hipMemcpyDeviceToHost) != hipSuccess) return -2;
  return (int)v;
}

template <bool BF16, bool ONE_SHOT>
static void dar_launch(const DarArgs& a, int world, int blocks, hipStream_t st) {
  dim3 g(blocks), b(DAR_THREADS);
  switch (world) {
    case 2: hipLaunchKernelGGL((dar_allreduce_kernel<BF16, 2, ONE_SHOT>), g, b, 0, st, a); break;
    case 4: hipLaunchKernelGGL((dar_allreduce_kernel<BF16, 4, ONE_SHOT>), g, b, 0, st, a); break;
    case 8: hipLaunchKernelGGL((dar_allreduce_kernel<BF16, 8, ONE_SHOT>), g, b, 0, st, a); break;
    default: hipLaunchKernelGGL((dar_allreduce_kernel<BF16, 0, ONE_SHOT>), g, b, 0, st, a); break;
  }
}

static int dar_run(DarContext* c, char* const* peer_base, long base_off, void* out, long bytes, int dtype, float scale,
                   int algo, bool inplace, hipStream_t st) {
  DarArgs a;
  for (int p = 0; p < DAR_MAX_WORLD; ++p) {
    a.peer_data[p] = p < c->world ? peer_base[p] + base_off : nullptr;
    a.peer_flags[p] = p < c->world ? c->peer_flags[p] : nullptr;
  }
  a.own_data = peer_base[c->rank] + base_off;
  a.out = out;
  a.nvec = bytes / 16;
  a.rank = c->rank; a.world = c->world; a.nblocks = c->blocks;
  a.seq = c->seq;
  a.spin_limit = c->spin_limit;
  a.scale = scale;
  a.status = c->status;
  a.inplace = inplace ? 1 : 0;
  if (dtype == 0) {
    if (algo == 1) dar_launch<true, true>(a, c->world, c->blocks, st);
    else dar_launch<true, false>(a, c->world, c->blocks, st);
  } else {
    if (algo == 1) dar_launch<false, true>(a, c->world, c->blocks, st);
    else dar_launch<false, false>(a, c->world, c->blocks, st);
  }
  return hipGetLastError() == hipSuccess ? 0 : -8;
}

// dtype: 0 = bf16, 1 = fp32.  algo: 0 auto, 1 one-shot, 2 two-shot.  `in` may equal `out`.
// Collective contract: every rank calls with the same count / dtype / algo, in the same order.
extern "C" int dlamd_dar_allreduce(void* ctx, const void* in, void* out, long count, int dtype, float scale, int algo,
                                   hipStream_t st) {
  DarContext* c = (DarContext*)ctx;
  if (!c || !in || !out || count <= 0) return -1;
  if (dtype != 0 && dtype != 1) return -2;
  const long bytes = count * (dtype == 0 ? 2 : 4);
  if ((bytes & 15) || bytes > c->max_bytes) return -3;
  if (((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return -4;
  for (int p = 0; p < c->world; ++p)
    if (!c->imported[p]) return -5;
  if (algo == 0) algo = bytes <= (512l << 10) ? 1 : 2;
  if (algo != 1 && algo != 2) return -6;
  c->seq += 1;
  const long slot_off = (long)(c->seq & 1u) * c->max_bytes;
  if (hipMemcpyAsync(c->data + slot_off, in, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return -7;
  return dar_run(c, c->peer_data, slot_off, out, bytes, dtype, scale, algo, false, st);
}

// ---- registered regions: buffers the caller owns (gradient buckets), reduced in place
extern "C" int dlamd_dar_alloc(long bytes, void** out) {
  if (!out || bytes <= 0) return -1;
  void* p = nullptr;
  if (hipMalloc(&p, (size_t)bytes) != hipSuccess) { (void)hipGetLastError(); return -2; }
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return -3; }
  *out = p;
  return 0;
}

extern "C" int dlamd_dar_free(void* p) {
  if (!p) return -1;
  return hipFree(p) == hipSuccess ? 0 : -2;
}

// base: an allocation of dlamd_dar_alloc (IPC handles exist per allocation, not per sub-range).  Returns the region id
// (the same on every rank when all register in the same order) and writes the 64-byte handle for the peers.
extern "C" int dlamd_dar_register(void* ctx, void* base, long bytes, void* handle) {
  DarContext* c = (DarContext*)ctx;
  if (!c || !base || !handle || bytes <= 0 || (bytes & 15) || ((uintptr_t)base & 15)) return -1;
  if (c->nregions >= DAR_MAX_REGIONS) return -2;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, base) != hipSuccess) { (void)hipGetLastError(); return -3; }
  memcpy(handle, &h, sizeof(h));
  const int id = c->nregions++;
  c->region[id][c->rank] = (char*)base;
  c->region_bytes[id] = bytes;
  return id;
}

extern "C" int dlamd_dar_register_peer(void* ctx, int region, int peer, const void* handle) {
  DarContext* c = (DarContext*)ctx;
  if (!c || !handle || region < 0 || region >= c->nregions || peer < 0 || peer >= c->world || peer == c->rank) return -1;
  if (c->region[region][peer]) return -2;
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* d = nullptr;
  if (hipIpcOpenMemHandle(&d, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return -3; }
  c->region[region][peer] = (char*)d;
  return 0;
}

// In-place all-reduce of the first `count` elements of a registered region (every rank: same region / count / dtype).
extern "C" int dlamd_dar_allreduce_region(void* ctx, int region, long count, int dtype, float scale, int algo,
                                          hipStream_t st) {
  DarContext* c = (DarContext*)ctx;
  if (!c || region < 0 || region >= c->nregions || count <= 0) return -1;
  if (dtype != 0 && dtype != 1) return -2;
  const long bytes = count * (dtype == 0 ? 2 : 4);
  if ((bytes & 15) || bytes > c->region_bytes[region]) return -3;
  for (int p = 0; p < c->world; ++p)
    if (!c->imported[p] || !c->region[region][p]) return -5;
  // the one-shot form keeps one 16-byte vector per thread in registers across its barrier
  const long grid_vecs = (long)c->blocks * DAR_THREADS;
  if (algo == 0) algo = (bytes <= (512l << 10) && bytes / 16 <= grid_vecs) ? 1 : 2;
  if (algo != 1 && algo != 2) return -6;
  if (algo == 1 && bytes / 16 > grid_vecs) return -7;
  c->seq += 1;
  return dar_run(c, c->region[region], 0, c->region[region][c->rank], bytes, dtype, scale, algo, true, st);
}

extern "C" int dlamd_dar_destroy(void* ctx) {
  DarContext* c = (DarContext*)ctx;
  if (!c) return -1;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank || !c->imported[p]) continue;
    (void)hipIpcCloseMemHandle(c->peer_data[p]);
    (void)hipIpcCloseMemHandle(c->peer_flags[p]);
    for (int r = 0; r < c->nregions; ++r)
      if (c->region[r][p]) (void)hipIpcCloseMemHandle(c->region[r][p]);
  }
  (void)hipFree(c->status);
  (void)hipFree(c->flags);
  (void)hipFree(c->data);
  delete c;
  return 0;
}

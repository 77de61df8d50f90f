// Shared by conv_chain.hip (a run of residual blocks per persistent launch) and conv_rcab_chain.hip (a run of RCABs): where a workgroup's strip comes
// from (claimed per XCD), which of its hand-offs may stay inside the XCD's L2, and the store / load forms of the two hand-offs.  Design: conv_chain.hip.
#pragma once
#include "block_common.hpp"

// work buffer (unsigned words): [0] epoch of the LAST launch, [CH_W_DONE] workgroups of this launch that hold a strip, [CH_W_STALL] test hook: non-zero = the
// workgroup of strip 0 publishes its first hand-off 0.7 s late (tests/test_chain_gpu.py: a DETERMINISTIC watchdog time-out; zero in every buffer the engine allocates), [CH_W_COUNT + x] strips claimed on XCD x, [CH_W_WHERE + s] (epoch << 8) + XCD strip s runs on,
// then per (strip, row half) ONE 128-BYTE LINE whose first word is the flag: (epoch << 8) + last block whose rows of that half are visible.  A line
// per flag, because flags may be stored sc0: a line that is dirty in an XCD's L2 for ONE word would serve that XCD's polls of its other words stale
constexpr int CH_W_DONE = 1, CH_W_STALL = 2, CH_W_COUNT = 8, CH_W_WHERE = 32, CH_MAX_XCD = 16, CH_FLAG_STRIDE = 32;
typedef unsigned int ch_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ch_store16_sc1(uint16_t* p, uint4 v) {
  const ch_u32x4 w = (ch_u32x4){v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" :: "v"(p), "v"(w) : "memory");   // s_nop: the >64-bit store data hazard is ours inside asm
}
__device__ __forceinline__ void ch_store16_sc0(uint16_t* p, uint4 v) {
  const ch_u32x4 w = (ch_u32x4){v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 0" :: "v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void ch_store_flag_sc0(unsigned* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ uint4 ch_load16_sc1(const uint16_t* p) {
  ch_u32x4 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w) : "v"(p) : "memory");
  return make_uint4(w.x, w.y, w.z, w.w);
}

// One round of a poll that has not seen its value yet: sleep, and every 256th round look at the clock and at the launch's status word.
//   -> 0: poll again; 1: this poll has waited CH_TIMEOUT (0.5 s of the 100 MHz s_memrealtime clock - round 6: a watchdog in TIME; the 2^20 rounds of round 5 were
//   meant as 0.1 s and took more than 3 s) - the caller stores its code in *status and goes on; 2: ANOTHER poll of the launch has timed out (co-residency did
//   not hold: the results are lost anyway) - stop waiting, so that the launch drains within milliseconds of the first time-out instead of one time-out per block.
// The clock is started at the first look (round 256): a hand-off that completes in microseconds never executes the s_memrealtime.
constexpr unsigned long long CH_TIMEOUT = 50000000ull;
// the test hook of CH_W_STALL: the calling lane sleeps 0.7 s (more than CH_TIMEOUT) before it publishes
__device__ __forceinline__ void ch_stall() {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 70000000ull) __builtin_amdgcn_s_sleep(64);
}
__device__ __forceinline__ int ch_poll_round(unsigned& spins, unsigned long long& t0, const unsigned* status) {
  __builtin_amdgcn_s_sleep(2);
  if ((++spins & 255u) != 0u) return 0;
  if (spins == 256u) { t0 = __builtin_amdgcn_s_memrealtime(); return 0; }
  if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return 2;
  return (__builtin_amdgcn_s_memrealtime() - t0 > CH_TIMEOUT) ? 1 : 0;
}


// Every pointer of a chain's block record comes out of device memory and is therefore a GENERIC pointer to the compiler: loads and stores through it are flat_*
// instructions, which count on BOTH wait counters and complete out of order - every counted s_waitcnt lgkmcnt(N) of the fragment reads around a T store
// degrades to a full drain (common.hpp::load_global_ptr: the same finding for the tail kernel, round 3; round 6 for the chains).  These helpers go through
// the global address space: global_load / global_store.
#define CH_GLOBAL(T, p) ((T __attribute__((address_space(1)))*)(unsigned long long)(p))
typedef unsigned int ch_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 ch_gld16(const void* p) { const ch_u32x4 v = *CH_GLOBAL(const ch_u32x4, p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 ch_gld8b(const void* p) { const ch_u32x2 v = *CH_GLOBAL(const ch_u32x2, p); return make_uint2(v.x, v.y); }
__device__ __forceinline__ void ch_gst16_nt(void* p, uint4 v) { __builtin_nontemporal_store((ch_u32x4){v.x, v.y, v.z, v.w}, CH_GLOBAL(ch_u32x4, p)); }
__device__ __forceinline__ unsigned ch_gld8(const unsigned char* p) { return *CH_GLOBAL(const unsigned char, p); }
__device__ __forceinline__ void ch_gst8(unsigned char* p, unsigned v) { *CH_GLOBAL(unsigned char, p) = (unsigned char)v; }
__device__ __forceinline__ f32x4 ch_gldf4(const float* p) { return *CH_GLOBAL(const f32x4, p); }

struct ChainPlace { int strip; unsigned xcc, epoch; int stall; };
// Thread 0 of the workgroup reads the launch's epoch (the last launch's + 1: tags of flags and placement words), claims a strip (own XCD first, then
// the others in turn: workgroups = slots, so a free one exists while this one has none), publishes where it physically runs, and hands all three
// to the workgroup through `claim` (LDS, four ints: the last one is the test hook CH_W_STALL); ends with a workgroup barrier.  XCD x runs the images x, x + nx, x + 2 nx, ...: whole
// images, all their strips behind ONE L2.  The workgroup whose claim is the launch's last puts the counters back to zero and stores the epoch for
// the next launch - every workgroup has read the old one by then (one thread per workgroup reads it, BEFORE its claim is counted) - so a chain needs
// no launch in front of it (a one-thread kernel there cost 4.6 us per chain: the launch boundary, not the work).
__device__ __forceinline__ ChainPlace chain_claim(unsigned* work, int N, int sy_n, int nx, int fake_xcc, int* claim) {
  if (threadIdx.x == 0) {
    const unsigned epoch = (__hip_atomic_load(work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u) & 0xffffffu;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    if (fake_xcc > 0) xcc = blockIdx.x % (unsigned)fake_xcc;      // (test hook: the claims' bookkeeping under heavy oversubscription; the host forces sc1 hand-offs with it)
    const int me = (int)(xcc % (unsigned)nx);
    int slot = -1;
    for (int d = 0; slot < 0; d = (d + 1) % nx) {
      const int x = (me + d) % nx;
      const int quota = (N / nx + (x < N % nx ? 1 : 0)) * sy_n;
      if ((int)__hip_atomic_load(work + CH_W_COUNT + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= quota) continue;
      const int t = (int)__hip_atomic_fetch_add(work + CH_W_COUNT + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t < quota) slot = (x + nx * (t / sy_n)) * sy_n + t % sy_n;
    }
    claim[0] = slot;
    claim[1] = (int)xcc;
    claim[2] = (int)epoch;
    claim[3] = (slot == 0) ? (int)__hip_atomic_load(work + CH_W_STALL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __hip_atomic_store(work + CH_W_WHERE + slot, (epoch << 8) + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // memory side: the neighbours may sit on any XCD
    // this claim is complete (its fetch_add has returned) and the epoch is read: count it; the last one re-arms the buffer for the next launch
    if (__hip_atomic_fetch_add(work + CH_W_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x) {
      for (int x = 0; x < CH_MAX_XCD; ++x) __hip_atomic_store(work + CH_W_COUNT + x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(work + CH_W_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(work, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  ChainPlace p;
  p.strip = claim[0];
  p.xcc = (unsigned)claim[1];
  p.epoch = (unsigned)__builtin_amdgcn_readfirstlane(claim[2]);
  p.stall = __builtin_amdgcn_readfirstlane(claim[3]);
  return p;
}
// does strip `nb` run on this workgroup's XCD?  (polls the word its workgroup publishes at claim time)
__device__ __forceinline__ bool chain_same_xcd(unsigned* work, unsigned epoch, int nb, unsigned my_xcc, unsigned* status) {
  unsigned w, spins = 0;
  unsigned long long t0 = 0;
  for (;;) {
    w = __hip_atomic_load(work + CH_W_WHERE + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((w >> 8) == epoch) break;
    const int r = ch_poll_round(spins, t0, status);
    if (r == 1 && (threadIdx.x & 63) == 0) atomicExch(status, 0x4ffu);
    if (r) break;
  }
  return ((w >> 8) == epoch) && ((w & 255u) == my_xcc);
}
__device__ __forceinline__ unsigned* chain_flags(unsigned* work, unsigned nstrips) { return work + CH_W_WHERE + ((nstrips + 31u) & ~31u); }
static inline int64_t chain_work_bytes(int64_t strips) { return (CH_W_WHERE + ((strips + 31) & ~(int64_t)31) + 2 * strips * CH_FLAG_STRIDE) * 4; }

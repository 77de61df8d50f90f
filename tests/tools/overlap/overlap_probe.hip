// Measurement tool (GPU box), not product: can TWO dependent launch chains on two streams hide each other's launch boundary and tile load
// when their workgroups are co-resident on a CU?  
//
// The residual-block launch of the training step (conv_block.hip) is one 512-thread workgroup per CU with 115 KB of LDS: nothing else fits
// beside it, so the boundary between two dependent launches (1.3 us), the 64 KB tile load (3 us) and the store tail are paid with an idle
// matrix pipe.  A workgroup of HALF the strip (3 output rows: 7 input rows + 5 rows of the intermediate activation = 76.8 KB, 4 waves) would
// leave room for a second one of ANOTHER chain (the other half of the batch, on a second stream) - at 14 % more MFMAs (5/3 instead of 8/6
// halo recompute).  This probe models a launch as: coalesced tile load -> LDS, barrier, NMFMA bf16 MFMAs per wave fed by ds_read_b128 (one
// read per two MFMAs, as block_sweep), whole-line non-temporal stores that the next launch of the chain reads, and times 16 dependent
// launches of
//   full : 256 workgroups x 512 threads, 64 KB in, 36.9 KB out, 378 MFMAs per wave                      (today's launch), one stream
//   half : 256 workgroups x 256 threads, 44.8 KB in, 18.4 KB out, 432 MFMAs per wave (76.8 KB of LDS), ONE chain alone,
//          TWO chains on two streams (co-resident pairs), and the two chains alternating on ONE stream (no overlap possible).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void st16_nt(uint4* p, uint4 v) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
  __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w); }

// MODE bits (parts of a launch compiled out, for the cost breakdown): 1 = no tile load, 2 = no stores, 4 = stores with the default cache policy, 8 = no MFMA loop
template <int THREADS, int IN_BYTES, int OUT_BYTES, int NMFMA, int LDS_BYTES, int MODE = 0>
__global__ void __launch_bounds__(THREADS) strip_probe(const uint4* __restrict__ in, uint4* __restrict__ out, const uint4* __restrict__ filt, unsigned total16) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  bf16x8 F[18];
#pragma unroll
  for (int s = 0; s < 18; ++s) { const uint4 v = filt[s * 64 + lane]; F[s] = *reinterpret_cast<const bf16x8*>(&v); }
  // tile: this workgroup's own OUT_BYTES of the previous launch's output plus what lies around them (the halo rows), coalesced
  constexpr int NIN = IN_BYTES / 16 / THREADS;
  const unsigned base = blockIdx.x * (OUT_BYTES / 16) + total16 - (IN_BYTES - OUT_BYTES) / 32;
  uint4 R[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) R[i] = (MODE & 1) ? make_uint4(tid, i, 0x3c003c00u, 0x3c003c00u) : in[(base + i * THREADS + tid) % total16];
#pragma unroll
  for (int i = 0; i < NIN; ++i) *reinterpret_cast<uint4*>(lds + (i * THREADS + tid) * 16) = R[i];
  __syncthreads();
  f32x4 acc[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // block_sweep's pipelining: the 6 fragment reads of group i + 1 are issued ahead of the 12 MFMAs of group i
  unsigned off = (unsigned)tid * 16u;
  auto rd = [&](bf16x8 (&I)[6]) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      I[r] = *reinterpret_cast<const bf16x8*>(lds + off);
      off += 1024u * 5u; if (off >= (unsigned)IN_BYTES) off -= (unsigned)IN_BYTES;
    }
  };
  auto mm = [&](const bf16x8 (&I)[6], int gsel) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      acc[2 * r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[(3 * gsel + r) % 18], I[r], acc[2 * r], 0, 0, 0);
      acc[2 * r + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[(3 * gsel + r + 9) % 18], I[r], acc[2 * r + 1], 0, 0, 0);
    }
  };
  bf16x8 Ia[6], Ib[6];
  rd(Ia);
#pragma unroll 1
  for (int it = 0; it < ((MODE & 8) ? 0 : NMFMA / 24); ++it) {
    rd(Ib);
    __builtin_amdgcn_sched_barrier(0);
    mm(Ia, 0);
    rd(Ia);
    __builtin_amdgcn_sched_barrier(0);
    mm(Ib, 1);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int j = 1; j < 12; ++j) s += acc[j];
  const unsigned bit = (__float_as_uint(s[0] + s[1] + s[2] + s[3]) >> 13) & 0x00010001u;      // data-dependent, keeps the values bf16 numbers of the same size
  __syncthreads();
  constexpr int NOUT = (OUT_BYTES / 16 + THREADS - 1) / THREADS;
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    if (i * THREADS + tid >= OUT_BYTES / 16) break;
    uint4 v = *reinterpret_cast<const uint4*>(lds + ((i * THREADS + tid) * 16 + (IN_BYTES - OUT_BYTES) / 2) % IN_BYTES);
    v.x ^= bit; v.z ^= bit;
    if (MODE & 2) { if (v.x == 0x12345678u && v.y == 0x9abcdef0u) out[tid] = v; }      // (never true on this data; keeps the values alive)
    else if (MODE & (4 | 128)) out[blockIdx.x * (OUT_BYTES / 16) + i * THREADS + tid] = v;
    else if (MODE & 16) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(out + blockIdx.x * (OUT_BYTES / 16) + i * THREADS + tid), "v"((u32x4){v.x, v.y, v.z, v.w}) : "memory");
    else if (MODE & 32) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(out + blockIdx.x * (OUT_BYTES / 16) + i * THREADS + tid), "v"((u32x4){v.x, v.y, v.z, v.w}) : "memory");
    else if (MODE & 64) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(out + blockIdx.x * (OUT_BYTES / 16) + i * THREADS + tid), "v"((u32x4){v.x, v.y, v.z, v.w}) : "memory");
    else st16_nt(out + blockIdx.x * (OUT_BYTES / 16) + i * THREADS + tid, v);
  }
}

// full: 10 x 50 px x 128 B = 64,000 -> 65,536 B in (rounded to whole loads); 6 x 48 x 128 = 36,864 B out; LDS 115,200
// half:  7 x 50 x 128 = 44,800 -> 45,056 B in; 3 x 48 x 128 = 18,432 B out; LDS 76,800
#define FULL strip_probe<512, 65536, 36864, 384, 115200>
#define HALF strip_probe<256, 45056, 18432, 432, 76800>

static float elapsed(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv) {
  const int nblocks = 16, reps = argc > 1 ? atoi(argv[1]) : 200, wgs = 256;
  const size_t full_bytes = (size_t)wgs * 36864, half_bytes = (size_t)wgs * 18432;
  std::vector<uint16_t> h(full_bytes / 2);
  srand(1);
  for (auto& v : h) { const float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
  std::vector<uint16_t> hf(18 * 64 * 8);
  for (auto& v : hf) { const float f = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.04f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
  uint4 *fa, *fb, *ha[2], *hb[2], *filt;
  CK(hipMalloc(&fa, full_bytes)); CK(hipMalloc(&fb, full_bytes)); CK(hipMalloc(&filt, hf.size() * 2));
  for (int c = 0; c < 2; ++c) { CK(hipMalloc(&ha[c], half_bytes)); CK(hipMalloc(&hb[c], half_bytes)); }
  CK(hipMemcpy(fa, h.data(), full_bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(fb, h.data(), full_bytes, hipMemcpyHostToDevice));
  for (int c = 0; c < 2; ++c) { CK(hipMemcpy(ha[c], h.data() + c * half_bytes / 2, half_bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(hb[c], h.data(), half_bytes, hipMemcpyHostToDevice)); }
  CK(hipMemcpy(filt, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  int occ_full = 0, occ_half = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_full, FULL, 512, 0)); CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_half, HALF, 256, 0));
  printf("workgroups per CU: full %d, half %d\n", occ_full, occ_half);
  auto full_chain = [&](hipStream_t s) { for (int b = 0; b < nblocks; ++b) hipLaunchKernelGGL(FULL, dim3(wgs), dim3(512), 0, s, (b & 1) ? fb : fa, (b & 1) ? fa : fb, filt, (unsigned)(full_bytes / 16)); };
  auto half_chain = [&](hipStream_t s, int c) { for (int b = 0; b < nblocks; ++b) hipLaunchKernelGGL(HALF, dim3(wgs), dim3(256), 0, s, (b & 1) ? hb[c] : ha[c], (b & 1) ? ha[c] : hb[c], filt, (unsigned)(half_bytes / 16)); };
  for (int round = 0; round < 3; ++round) {
    // (1) today's launch
    for (int w = 0; w < 20; ++w) full_chain(s0);
    CK(hipEventRecord(e0, s0));
    for (int r = 0; r < reps; ++r) full_chain(s0);
    CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
    const float t_full = elapsed(e0, e1) * 1e3f / (reps * nblocks);
    // (2) one half chain alone
    for (int w = 0; w < 20; ++w) half_chain(s0, 0);
    CK(hipEventRecord(e0, s0));
    for (int r = 0; r < reps; ++r) half_chain(s0, 0);
    CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
    const float t_half1 = elapsed(e0, e1) * 1e3f / (reps * nblocks);
    // (3) both half chains alternating on one stream
    CK(hipEventRecord(e0, s0));
    for (int r = 0; r < reps; ++r) {
      for (int b = 0; b < nblocks; ++b) for (int c = 0; c < 2; ++c)
        hipLaunchKernelGGL(HALF, dim3(wgs), dim3(256), 0, s0, (b & 1) ? hb[c] : ha[c], (b & 1) ? ha[c] : hb[c], filt, (unsigned)(half_bytes / 16));
    }
    CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
    const float t_half_seq = elapsed(e0, e1) * 1e3f / (reps * nblocks);
    // (4) the two half chains on two streams
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s0)); CK(hipStreamWaitEvent(s1, e0, 0));
    for (int r = 0; r < reps; ++r) { half_chain(s0, 0); half_chain(s1, 1); }
    CK(hipEventRecord(e2, s1)); CK(hipStreamWaitEvent(s0, e2, 0));
    CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
    const float t_half_2s = elapsed(e0, e1) * 1e3f / (reps * nblocks);
    // (5) the whole batch as 512 half workgroups in ONE launch (co-resident pairs that start together)
    CK(hipEventRecord(e0, s0));
    for (int r = 0; r < reps; ++r)
      for (int b = 0; b < nblocks; ++b) hipLaunchKernelGGL(HALF, dim3(2 * wgs), dim3(256), 0, s0, (b & 1) ? fb : fa, (b & 1) ? fa : fb, filt, (unsigned)(full_bytes / 16));
    CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
    const float t_half_512 = elapsed(e0, e1) * 1e3f / (reps * nblocks);
    printf("   (512 half workgroups in one launch: %.2f)\n", t_half_512);
    if (round == 2) {
      auto timed = [&](auto kern, const char* what) {
        for (int w = 0; w < 10; ++w) for (int b = 0; b < nblocks; ++b) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, s0, (b & 1) ? fb : fa, (b & 1) ? fa : fb, filt, (unsigned)(full_bytes / 16));
        CK(hipEventRecord(e0, s0));
        for (int r = 0; r < reps; ++r) for (int b = 0; b < nblocks; ++b) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, s0, (b & 1) ? fb : fa, (b & 1) ? fa : fb, filt, (unsigned)(full_bytes / 16));
        CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
        printf("   full launch, %-44s %.2f us\n", what, elapsed(e0, e1) * 1e3f / (reps * nblocks));
      };
      timed(strip_probe<512, 65536, 36864, 384, 115200, 8>, "no MFMA loop:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 1>, "no tile load:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 2>, "no stores:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 4>, "default-policy stores:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 16>, "sc0 stores:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 32>, "sc1 stores:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 64>, "sc0 sc1 stores:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 128>, "default-policy stores + early agent release:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 8 + 4>, "no MFMA loop, default-policy stores:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 8 + 32>, "no MFMA loop, sc1 stores:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 9>, "no tile load, no MFMA loop:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 10>, "no stores, no MFMA loop:");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 11>, "nothing but the launch (filter fetch, barriers):");
      timed(strip_probe<512, 65536, 36864, 384, 115200, 3>, "MFMA loop only:");
    }
    printf("per residual block of the whole batch, us: full launches %.2f | half launches: one chain alone %.2f per launch, both chains on one stream %.2f, "
           "on two streams %.2f  (two streams / full = %.3f)\n", t_full, t_half1, t_half_seq, t_half_2s, t_half_2s / t_full);
  }
  return 0;
}

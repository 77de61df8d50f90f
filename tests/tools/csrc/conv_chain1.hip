// EXPERIMENT (tests/tools/csrc, never launched by the engine; outcome: profiles/r06_block_body.txt - isolated 3 % faster than conv_chain.hip, the training step
// 1.7 % SLOWER: not shipped).  The chain of residual blocks (conv_chain.hip) with ONE wave per SIMD - round 6, VERDICT r5 item 2 ("the block body, not the hand-off").
//
// What round 5 measured: with no hand-off at all a block of the 512-thread chain costs 12.2 us against 6.9 us of MFMA issue.  The ISA of its block loop says
// why (profiles/r06_block_body.txt): per SIMD and block 756 MFMAs (12,096 cycles of the matrix pipe) AND ~2,000 vector instructions of the two waves that
// share the SIMD (~8,000 issue cycles) - and the two waves are the two row halves of one strip, coupled twice per block, so they sweep together and run their
// epilogues together: the sum, not the overlap.  A third of those vector instructions is lane geometry that does not change from block to block (the five
// windows' read bases: 5 x 16 addresses, the offsets of the T pairs, of the store pieces and of the halo pieces) and is recomputed per block because the
// kernel sits at 253-255 of the 256 registers a wave has at two waves per SIMD (r05 negative 12: every attempt to hoist it spilled).
//
// Here a workgroup is 256 threads, one wave per SIMD, 512 registers per wave (256 VGPRs + 256 AGPRs: filter and accumulators live in the AGPRs):
//   * wave q owns output channels 16q .. 16q + 15 of ALL rows of the strip (the two row halves of conv_chain.hip in one instruction stream);
//   * every per-lane address is computed ONCE in front of the block loop: two sets of 16 read bases (input image, T image) serve every window of every block -
//     a window at row R is the same set with its XOR classes rotated by 2 R (50 columns = 2 mod 8) and a compile-time immediate;
//   * taller sweeps: T rows 2 .. 5 in one four-row sweep, the six output rows in one six-row sweep (a B fragment feeds three output rows instead of two:
//     8 fragment reads per 18 MFMAs where the row halves needed 12);
//   * one filter fetch per phase and wave instead of two half-workgroups fetching the same slice;
//   * hand-offs always through the memory side (write-through stores, sc1 loads: r05 measured the XCD-local form within 1 % of it once every flag has its
//     own line), one flag per strip; strips are still CLAIMED per XCD (chain_common.hpp) so that vertical neighbours share an L2.
// Per accumulator the MFMA order is conv_block.hip's ((channel half, tap column) groups, tap row innermost): the results are BITWISE those of one launch per
// block and of conv_chain.hip (tests/test_chain_gpu.py runs the same cases through both forms).
#include "chain_common.hpp"
#include "rumpy_experimental.h"
#ifndef C1_AHEAD
#define C1_AHEAD 1      // fragment reads this many groups ahead of the MFMAs
#endif
#ifndef C1_HV
#define C1_HV 2          // vector instructions of a hosted epilogue slice pinned per MFMA of the hosting sweep (0: the slice trails the group's MFMAs)
#endif
#ifndef C1_E1A
#define C1_E1A 2         // where the epilogue of the four halo-free T rows runs: 0 = behind the halo sweeps with the rest, 1 = under the halo sweeps' MFMAs, 2 = in the waits of the hand-off
#endif
#ifndef C1_MERGE_D
#define C1_MERGE_D 1     // the two two-row halo windows in one loop
#endif
#ifndef C1_ABL
#define C1_ABL 0          // measurement builds only (results WRONG): 1 = the sweeps' fragment reads compiled out, 2 = their MFMAs
#endif
#ifndef C1_INTERLEAVE
#define C1_INTERLEAVE 1  // the next group's fragment reads pinned BETWEEN this group's MFMAs (0: in front of them, A/B)
#endif

struct C1Blk {
  const uint16_t* x; const uint4* w1; const float* b1; const uint4* w2; const float* b2;
  const uint16_t* res2; uint16_t* t; uint16_t* out; unsigned char* mbits; float scale1, scale2;
};
static_assert(sizeof(C1Blk) == sizeof(rumpy_res_chain_block), "rumpy_res_chain_block is the device-side block record");
struct C1Dev { const C1Blk* blk; int nblk, N, H, W, sy_n; unsigned* work; unsigned* status; int nxcd, fake_xcc; };


// (block pointers go through the global address space and the neighbour's rows through agent-scope atomic loads: chain_common.hpp::CH_GLOBAL)
#define C1_GLOBAL CH_GLOBAL
#define gld16 ch_gld16
#define gst16_nt ch_gst16_nt
#define gld8 ch_gld8
#define gst8 ch_gst8
#define gldf4 ch_gldf4
// 16 bytes another workgroup has published (write-through stores behind a flag): two 8-byte agent-scope atomic loads - sc1, never served from a stale line,
// and tracked by the compiler's wait counters (an inline-asm load's destination registers look ready to the register allocator)
__device__ __forceinline__ uint4 gld16_agent(const void* p) {
  const unsigned long long lo = __hip_atomic_load(C1_GLOBAL(const unsigned long long, p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(C1_GLOBAL(const unsigned long long, p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
}

constexpr int C1T = 256;
constexpr int C1_SREGS = BSH * BSW * 8 / C1T;        // 9: the strip's 6 x 48 pixels as 16-byte pieces per thread
constexpr int C1_HREGS = 2 * BSW * 8 / C1T;          // 3: two halo rows per side
static_assert(C1_SREGS * C1T == BSH * BSW * 8 && C1_HREGS * C1T == 2 * BSW * 8, "piece counts divide the workgroup");

// -DC1_STAMPS (measurement builds only): phase time stamps (s_memrealtime, 100 MHz) of every wave in the MIDDLE block
#ifdef C1_STAMPS
__device__ unsigned long long* g_c1_stamps;
// (second half of the buffer: the same stamps in shader-clock cycles, s_memtime - their ratio is the clock the kernel really ran at)
#define C1_STAMP(k) do { if (b == a.nblk / 2 && (threadIdx.x & 63) == 0 && g_c1_stamps) { const size_t si_ = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (k); \
    g_c1_stamps[si_] = __builtin_amdgcn_s_memrealtime(); g_c1_stamps[(size_t)gridDim.x * 64 + si_] = __builtin_amdgcn_s_memtime(); } } while (0)
extern "C" int rumpy_debug_c1_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_c1_stamps), &buf, sizeof(buf)); }
#else
#define C1_STAMP(k) do { } while (0)
#endif

// block_common.hpp::block_sweep for a window that starts at row R0 of the image the bases were prepared for (sweep_bases with row0 = 0): the XOR class of
// a read is ((R0 + r) * 50 + kx) & 7 - a compile-time index into the SAME 16 base registers - and the rest of the address is an immediate
template <int ROWS, int R0, int FMT, class Hook = NoHook, int HV = 0, int AHEAD = C1_AHEAD>
__device__ __forceinline__ void sweep_at(f32x4 (&acc)[ROWS][3], const bf16x8 (&F)[18], const unsigned char* lds, const unsigned (&off)[8][2], Hook hook = Hook()) {
  constexpr int NG = 18;
  bf16x8 I[AHEAD + 1][ROWS + 2];
  auto load_group = [&](int grp, bf16x8 (&dst)[ROWS + 2]) {
    const int half = grp / 9, kx = (grp % 9) / 3, c = grp % 3;
#pragma unroll
    for (int r = 0; r < ROWS + 2; ++r)
      dst[r] = *reinterpret_cast<const bf16x8*>(lds + off[((R0 + r) * BCOLS + kx) & 7][half] + ((R0 + r) * BCOLS + 16 * c + kx) * 128);
  };
#pragma unroll
  for (int g0 = 0; g0 < AHEAD; ++g0) load_group(g0, I[g0]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int grp = 0; grp < NG; ++grp) {
    if (grp + AHEAD < NG && !(C1_ABL & 1)) load_group(grp + AHEAD, I[(grp + AHEAD) % (AHEAD + 1)]);
    const int half = grp / 9, kx = (grp % 9) / 3, c = grp % 3;
    if (!(C1_ABL & 2)) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
        acc[r][c] = mfma16<FMT>(F[(ky * 3 + kx) * 2 + half], I[(C1_ABL & 1) ? 0 : grp % (AHEAD + 1)][r + ky], acc[r][c]);
    } else if (grp + AHEAD >= NG) {
#pragma unroll
      for (int r = 0; r < ROWS + 2; ++r) acc[r % ROWS][c] = mfma16<FMT>(F[0], I[grp % (AHEAD + 1)][r], acc[r % ROWS][c]);      // (keeps the reads alive)
    }
    hook(grp);
#if C1_INTERLEAVE
    // ONE wave per SIMD: nobody else fills the matrix pipe while this wave issues its fragment reads.  In front of the group's MFMAs (round-5 order) the
    // ROWS + 2 reads of the next group cost the pipe ~80 idle cycles per group (stamps: the six-row sweep at 64 % of its MFMA time); issued BETWEEN the MFMAs
    // they are nearly free (MI355X_MICROARCH.md, LDS: 2 ds_read_b128 per MFMA gap cost at most 3 cycles).  The group's region is pinned to
    // (1 read, PER MFMAs) x (ROWS + 2); what is left (MFMAs, the hook's instructions) follows.
    // HV > 0: the hook's vector instructions (a slice of an epilogue that does not depend on this sweep) are pinned between the MFMAs too, HV per MFMA:
    // an MFMA 16x16x32 holds the SIMD's vector issue for 8 of its 16 cycles - two 4-cycle instructions fit into every gap
    constexpr int M = 3 * ROWS, PER = M / (ROWS + 2) > 0 ? M / (ROWS + 2) : 1;
    if (grp + AHEAD < NG) {
#pragma unroll
      for (int i = 0; i < ROWS + 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (HV > 0) __builtin_amdgcn_sched_group_barrier(0x002, HV * PER, 0);
      }
      if (HV > 0) {
#pragma unroll
        for (int i = 0; i < M - PER * (ROWS + 2); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, HV, 0);
        }
      }
    } else if (HV > 0) {
#pragma unroll
      for (int i = 0; i < M; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, HV, 0);
      }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);   // a group's instructions stay inside the group
  }
}

// two two-row windows (rows RA .. RA + 3 and RB .. RB + 3 of the image) in ONE loop: a group = 8 fragment reads and 12 MFMAs - alone, a two-row window's
// group is 6 MFMAs = 96 cycles, shorter than the latency of the reads it has to cover.  Per accumulator the order is sweep_at's.
template <int RA, int RB, int FMT, class Hook = NoHook, int AHEAD = C1_AHEAD>
__device__ __forceinline__ void sweep_two(f32x4 (&accA)[2][3], f32x4 (&accB)[2][3], const bf16x8 (&F)[18], const unsigned char* lds, const unsigned (&off)[8][2], Hook hook = Hook()) {
  constexpr int NG = 18;
  bf16x8 I[AHEAD + 1][8];
  auto load_group = [&](int grp, bf16x8 (&dst)[8]) {
    const int half = grp / 9, kx = (grp % 9) / 3, c = grp % 3;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = (r < 4 ? RA : RB - 4) + r;
      dst[r] = *reinterpret_cast<const bf16x8*>(lds + off[(row * BCOLS + kx) & 7][half] + (row * BCOLS + 16 * c + kx) * 128);
    }
  };
#pragma unroll
  for (int g0 = 0; g0 < AHEAD; ++g0) load_group(g0, I[g0]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int grp = 0; grp < NG; ++grp) {
    if (grp + AHEAD < NG && !(C1_ABL & 1)) load_group(grp + AHEAD, I[(grp + AHEAD) % (AHEAD + 1)]);
    const int half = grp / 9, kx = (grp % 9) / 3, c = grp % 3;
    if (!(C1_ABL & 2)) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        accA[r][c] = mfma16<FMT>(F[(ky * 3 + kx) * 2 + half], I[(C1_ABL & 1) ? 0 : grp % (AHEAD + 1)][r + ky], accA[r][c]);
        accB[r][c] = mfma16<FMT>(F[(ky * 3 + kx) * 2 + half], I[(C1_ABL & 1) ? 0 : grp % (AHEAD + 1)][4 + r + ky], accB[r][c]);
      }
    } else if (grp + AHEAD >= NG) {
#pragma unroll
      for (int r = 0; r < 8; ++r) accA[r % 2][c] = mfma16<FMT>(F[0], I[grp % (AHEAD + 1)][r], accA[r % 2][c]);
    }
    hook(grp);
#if C1_INTERLEAVE
    if (grp + AHEAD < NG) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
}

// FORM 1: forward (ReLU, mask bytes written if given); FORM 3: data gradient (* scale1, mask bytes read)
template <int FORM, int FMT = RUMPY_FMT_BF16>
__global__ void __launch_bounds__(C1T, 1) block_chain1_kernel(C1Dev a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[BXBYTES + BTBYTES];
  __shared__ unsigned gate[5];             // waves that have: written their OUT channels [0], written the halo rows [1], written their T channels [2], seen their OUT stores acknowledged [3]
  __shared__ int claim[4];
  const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, g = lane >> 4;
  // ---- which strip this workgroup runs: CLAIMED, per XCD (chain_common.hpp) ----
  const ChainPlace place = chain_claim(a.work, a.N, a.sy_n, a.nxcd, a.fake_xcc, claim);
  const unsigned epoch = place.epoch;
  const int strip = place.strip;
  const int n = strip / a.sy_n, sy = strip - n * a.sy_n;
  const bool has_top = sy > 0, has_bot = sy + 1 < a.sy_n;
  unsigned* const flags = chain_flags(a.work, gridDim.x);
  unsigned* const my_flag = flags + (2 * strip) * CH_FLAG_STRIDE;
  const C1Blk b0 = a.blk[0];

  // ---- block 0: input rows 6sy-2 .. 6sy+7, columns -1 .. 48 -> LDS ----
  {
    constexpr int XR = (BPIECES + C1T - 1) / C1T;      // 16
    uint4 R[XR];
    const int y0 = sy * BSH - 2;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int p = tid + C1T * i;
      const int pix = p >> 3, part = p & 7;
      const int lr = pix / BCOLS, lc = pix - lr * BCOLS;
      const int y = y0 + lr, x = lc - 1;
      const bool ok = (p < BPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
      const int e = ok ? ((n * a.H + y) * a.W + x) * 64 + part * 8 : 0;
      R[i] = keep_if(gld16(b0.x + (unsigned)e), ok);
    }
    if (tid < 5) gate[tid] = 0u;
    if (tid < BTROWS * 2 * 8) {            // border columns of the T image: convB's zero padding, never written by the epilogues
      const int row = tid >> 4, side = (tid >> 3) & 1, chunk = tid & 7;
      *reinterpret_cast<uint4*>(lds + BXBYTES + swz(row * BCOLS + side * (BCOLS - 1), chunk)) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int p = tid + C1T * i;
      if (p < BPIECES) *reinterpret_cast<uint4*>(lds + swz(p >> 3, p & 7)) = R[i];
    }
  }
  bf16x8 F[18];
  {
    const uint4* wp = b0.w1 + (size_t)q * 18 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(gld16(wp + t * 64));
  }

  // ---- lane geometry, ONCE for the whole chain ----
  const int c0 = 16 * q + 4 * g;
  const int gpair = 4 * (g & ~1);
  const int chunk8 = 2 * q + (gpair >> 3);            // 16-byte chunk of this lane's 8 channels in the paired layout
  // T pairs k < 8: (T row k, column tile 0 | 1); k = 8 .. 11: (T rows 2(k-8) | 2(k-8)+1, column tile 2): element offset in a [N,H,W,64] tensor (or outside) and LDS cell
  unsigned moff[12], tcell[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int jr = (k < 8) ? k : (2 * (k - 8) + (g & 1)), c = (k < 8) ? (g & 1) : 2;
    const int y = sy * BSH - 1 + jr, xx = 16 * c + px;
    const bool in = ((unsigned)y < (unsigned)a.H) & (xx < a.W);
    moff[k] = in ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
    tcell[k] = (unsigned)BXBYTES + swz(jr * BCOLS + xx + 1, chunk8);
  }
  // OUT pairs k < 6: (strip row k, column tile 0 | 1); k = 6 .. 8: (rows 2(k-6) | 2(k-6)+1, column tile 2)
  unsigned ooff[9], xcell[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int r = (k < 6) ? k : (2 * (k - 6) + (g & 1)), c = (k < 6) ? (g & 1) : 2;
    const int y = sy * BSH + r, xx = 16 * c + px;
    ooff[k] = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
    xcell[k] = swz((r + 2) * BCOLS + xx + 1, chunk8);
  }
  // whole-line store pieces of the strip's own 6 x 48 pixels: p = tid + 256 i = chunk p & 7 of strip pixel p >> 3
  unsigned soff[C1_SREGS], sldt[C1_SREGS], sldx[C1_SREGS];
#pragma unroll
  for (int i = 0; i < C1_SREGS; ++i) {
    const int p = tid + C1T * i, pix = p >> 3, r = pix / BSW, col = pix - r * BSW, y = sy * BSH + r;
    soff[i] = (y < a.H && col < a.W) ? (unsigned)(((n * a.H + y) * a.W + col) * 64 + (p & 7) * 8) : 0xffffffffu;
    sldt[i] = (unsigned)BXBYTES + swz((r + 1) * BCOLS + col + 1, p & 7);
    sldx[i] = swz((r + 2) * BCOLS + col + 1, p & 7);
  }
  // halo pieces: rows 6sy-2, 6sy-1 (from the strip above) and 6sy+6, 6sy+7 (from the strip below): 768 pieces per side
  unsigned hoff[2 * C1_HREGS], hlds[2 * C1_HREGS];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < C1_HREGS; ++i) {
      const int p = tid + C1T * i, pix = p >> 3, r = pix / BSW, col = pix - r * BSW;
      const int y = (s == 0) ? sy * BSH - 2 + r : sy * BSH + BSH + r;
      const bool has = (s == 0) ? has_top : has_bot;
      hoff[C1_HREGS * s + i] = (has && (unsigned)y < (unsigned)a.H && col < a.W) ? (unsigned)(((n * a.H + y) * a.W + col) * 64 + (p & 7) * 8) : 0xffffffffu;
      hlds[C1_HREGS * s + i] = swz(((s == 0) ? r : BSH + 2 + r) * BCOLS + col + 1, p & 7);
    }
  unsigned offX[8][2], offT[8][2];
  sweep_bases(offX, 0u, 0, px, g);
  sweep_bases(offT, (unsigned)BXBYTES, 0, px, g);
  __syncthreads();

  // Order of a block (round 6, third form): everything that needs NO halo row first - T rows 2 .. 5, their epilogue, and from them OUT rows 2, 3 with theirs -
  // then the four T rows that do, the rest of the second conv, the stores.  The hand-off of block b - 1 (stores acknowledged -> flag -> the neighbours' flags ->
  // their rows) runs beside ~5 us of this strip's own work instead of in front of the halo sweep (stamps of the first form: 2.2 us of waiting per block).
  // Both filters of a block are resident (144 of the 256 AGPRs).
  bf16x8 F2[18];
  for (int b = 0; b < a.nblk; ++b) {
    const C1Blk blk = a.blk[b];
    C1_STAMP(0);
    {
      const uint4* wp = blk.w2 + (size_t)q * 18 * 64 + lane;      // the second filter: L2 hits that land under the first sweep
#pragma unroll
      for (int t = 0; t < 18; ++t) F2[t] = as_bf16x8(gld16(wp + t * 64));
    }
    unsigned MB[FORM == 3 ? 12 : 1];
    if (FORM == 3) {
#pragma unroll
      for (int k = 0; k < 12; ++k) MB[FORM == 3 ? k : 0] = gld8(blk.mbits + ((moff[k] != 0xffffffffu ? moff[k] : 0u) >> 3));
    }
    f32x4 acc[8][3];                                     // T row j = image row 6sy - 1 + j
    f32x4 acc2[6][3];                                    // OUT row r = image row 6sy + r
    {
      f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (blk.b1) b4 = gldf4(blk.b1 + c0);
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] = b4;
      b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (blk.b2) b4 = gldf4(blk.b2 + c0);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc2[r][c] = b4;
    }
    // ---- epilogue 1 of one pair of T tiles: T = post1(acc) -> the T image in LDS (k is a constant after unrolling) ----
    auto e1_pair = [&](int k) {
      f32x4 tx = (k < 8) ? acc[k < 8 ? k : 0][0] : acc[2 * (k < 8 ? 0 : k - 8)][2];
      f32x4 ty = (k < 8) ? acc[k < 8 ? k : 0][1] : acc[2 * (k < 8 ? 0 : k - 8) + 1][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tx[i] = (FORM == 1) ? relu_f32(tx[i]) : tx[i] * blk.scale1;
        ty[i] = (FORM == 1) ? relu_f32(ty[i]) : ty[i] * blk.scale1;
      }
      float v[8];
      pair_up(tx, ty, g, v);
      uint4 o = make_uint4(0, 0, 0, 0);                  // outside the image: convB's zero padding
      if (moff[k] != 0xffffffffu) {
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (FORM == 3) o = relu_mask_bits(o, MB[FORM == 3 ? k : 0]);
      }
      *reinterpret_cast<uint4*>(lds + tcell[k]) = o;
    };
    // ---- epilogue 2 of one pair of OUT tiles: OUT = X + scale2 * acc2 [+ res2], in place over the input image's centre rows ----
    auto e2_pair = [&](int k) {
      const f32x4 tx = (k < 6) ? acc2[k < 6 ? k : 0][0] : acc2[2 * (k < 6 ? 0 : k - 6)][2];
      const f32x4 ty = (k < 6) ? acc2[k < 6 ? k : 0][1] : acc2[2 * (k < 6 ? 0 : k - 6) + 1][2];
      float v[8], m[8];
      pair_up(tx, ty, g, v);
      if (ooff[k] != 0xffffffffu) {
        unsigned char* cell = lds + xcell[k];
        unpack8<FMT>(*reinterpret_cast<const uint4*>(cell), m);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], blk.scale2, m[j]);
        if (blk.res2) {
          unpack8<FMT>(gld16(blk.res2 + ooff[k]), m);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += m[j];
        }
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<uint4*>(cell) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    };

    // (1) T rows 2 .. 5: input rows 2 .. 7 = the strip's own rows (block b - 1's OUT, complete behind the gate at its end)
    sweep_at<4, 2, FMT>(*reinterpret_cast<f32x4(*)[4][3]>(&acc[2]), F, lds, offX);
    C1_STAMP(1);
    if (b > 0) {
      // (2) publish block b - 1: this wave's OUT stores are acknowledged (under the sweep above) -> count in -> one lane stores the strip's flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      gate_arrive(&gate[3], lane);
      if (q == 0) {
        gate_wait(&gate[3], 4u * (unsigned)b);
        if (lane == 0) __hip_atomic_store(my_flag, (epoch << 8) + (unsigned)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    C1_STAMP(2);
    // (3) their epilogue: pairs 2 .. 5 (rows 2 .. 5 x column tiles 0 | 1), 9 and 10 (rows 2 | 3 and 4 | 5 of column tile 2)
    e1_pair(2); e1_pair(3); e1_pair(4); e1_pair(5); e1_pair(9); e1_pair(10);
    gate_arrive(&gate[4], lane);
    C1_STAMP(3);
    uint4 Hr[2 * C1_HREGS];
    if (b > 0) {
      // (4) the neighbours' flags (published a sweep and an epilogue ago if the strips run in step), then their rows: requested here, used behind (5) and (6)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if ((s == 0) ? has_top : has_bot) {
          const unsigned* nf = flags + (2 * ((s == 0) ? strip - 1 : strip + 1)) * CH_FLAG_STRIDE;
          unsigned spins = 0;
          unsigned long long t0 = 0;
          for (;;) {
            const unsigned f = __hip_atomic_load(nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((f >> 8) == epoch && (f & 0xffu) >= ((unsigned)b & 0xffu)) break;
            const int tr = ch_poll_round(spins, t0, a.status);
            if (tr == 1 && lane == 0) atomicExch(a.status, 0x500u + (unsigned)b);
            if (tr) break;
          }
        }
      }
      // (agent-scope atomic loads = global_load ... sc1 that the COMPILER tracks: the rows are used two phases later, and an inline-asm load's
      // destination registers look ready to the register allocator - it moved them to AGPRs before the data had arrived, found the hard way)
#pragma unroll
      for (int i = 0; i < 2 * C1_HREGS; ++i) Hr[i] = gld16_agent(blk.x + (hoff[i] != 0xffffffffu ? hoff[i] : 0u));
    }
    C1_STAMP(4);
    // (5) OUT rows 2, 3 from T rows 2 .. 5 (every wave's channels of them: the gate also says that nobody sweeps over input rows 4, 5 any more)
    gate_wait(&gate[4], 4u * (unsigned)(b + 1));
    sweep_at<2, 2, FMT>(*reinterpret_cast<f32x4(*)[2][3]>(&acc2[2]), F2, lds, offT);
    C1_STAMP(5);
    // (6) their epilogue: pairs 2, 3 (rows 2, 3 x column tiles 0 | 1) and 7 (rows 2 | 3 of column tile 2) - in place over input rows 4, 5, which the halo
    // sweeps do not read
    e2_pair(2); e2_pair(3); e2_pair(7);
    C1_STAMP(6);
    if (b > 0) {
#pragma unroll
      for (int i = 0; i < 2 * C1_HREGS; ++i)
        if (hoff[i] != 0xffffffffu) *reinterpret_cast<uint4*>(lds + hlds[i]) = Hr[i];
      gate_arrive(&gate[1], lane);
      gate_wait(&gate[1], 4u * (unsigned)b);
    }
    C1_STAMP(7);
    // (7) the four T rows that need the halo rows: rows 0, 1 (input rows 0 .. 3) and 6, 7 (input rows 6 .. 9), and their epilogue
    sweep_two<0, 6, FMT>(*reinterpret_cast<f32x4(*)[2][3]>(&acc[0]), *reinterpret_cast<f32x4(*)[2][3]>(&acc[6]), F, lds, offX);
    C1_STAMP(8);
    e1_pair(0); e1_pair(1); e1_pair(8); e1_pair(6); e1_pair(7); e1_pair(11);
    gate_arrive(&gate[2], lane);
    gate_wait(&gate[2], 4u * (unsigned)(b + 1));         // the T image is complete - and nobody sweeps over the input image any more
    C1_STAMP(9);
    // the strip's own rows of T (+ mask bytes) -> HBM from the LDS image: whole lines, non-temporal, one piece after every second group of the sweep below
    uint4 S[C1_SREGS];
    const bool t_out = blk.t != nullptr;
    if (t_out) {
#pragma unroll
      for (int i = 0; i < C1_SREGS; ++i) S[i] = *reinterpret_cast<const uint4*>(lds + sldt[i]);
    }
    auto t_store = [&](int grp) {
      if (grp % 2 == 0 && grp / 2 < C1_SREGS) {
        const int i = grp / 2 < C1_SREGS ? grp / 2 : 0;
        if (t_out && soff[i] != 0xffffffffu) {
          gst16_nt(blk.t + soff[i], S[i]);
          if (FORM == 1 && blk.mbits) gst8(blk.mbits + (soff[i] >> 3), relu_bits(S[i]));
        }
      }
    };
    // (8) OUT rows 0, 1 (T rows 0 .. 3) and 4, 5 (T rows 4 .. 7)
    sweep_two<0, 4, FMT, decltype(t_store)>(*reinterpret_cast<f32x4(*)[2][3]>(&acc2[0]), *reinterpret_cast<f32x4(*)[2][3]>(&acc2[4]), F2, lds, offT, t_store);
    C1_STAMP(10);
    if (b + 1 < a.nblk) {                                // the next block's first filter lands under the epilogue
      const uint4* wp = a.blk[b + 1].w1 + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(gld16(wp + t * 64));
    }
    e2_pair(0); e2_pair(1); e2_pair(6); e2_pair(4); e2_pair(5); e2_pair(8);
    C1_STAMP(11);
    gate_arrive(&gate[0], lane);
    gate_wait(&gate[0], 4u * (unsigned)(b + 1));
    // the strip's 6 OUT rows -> HBM: whole lines, write-through (the neighbours read rows 0, 1 / 4, 5 back)
#pragma unroll
    for (int i = 0; i < C1_SREGS; ++i) S[i] = *reinterpret_cast<const uint4*>(lds + sldx[i]);
#pragma unroll
    for (int i = 0; i < C1_SREGS; ++i)
      if (soff[i] != 0xffffffffu) ch_store16_sc1(blk.out + soff[i], S[i]);
    C1_STAMP(12);
  }
}

extern "C" int rumpy_res_chain1(const rumpy_res_chain_args* p, void* stream) {
  if (!p || !p->blocks || !p->work || !p->status || p->nblocks <= 0 || p->nblocks > 255) { rumpy_set_error("rumpy_res_chain1: bad argument"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->W > BSW) { rumpy_set_error("rumpy_res_chain1: needs 0 < W <= 48 (got %d)", p->W); return RUMPY_E_ARG; }
  if (p->fmt != RUMPY_FMT_BF16 && !(p->fmt == RUMPY_FMT_F16 && !p->backward)) { rumpy_set_error("rumpy_res_chain1: fmt %d is a forward-only format", p->fmt); return RUMPY_E_ARG; }
  if (p->edge_w) { rumpy_set_error("rumpy_res_chain1: the conv at the chain's outer end is not built into this form"); return RUMPY_E_ARG; }
  const int sy_n = (p->H + BSH - 1) / BSH;
  if (p->N * sy_n > rumpy_device_cus()) { rumpy_set_error("rumpy_res_chain1: %d strips do not fit %d CUs (all must be co-resident)", p->N * sy_n, rumpy_device_cus()); return RUMPY_E_ARG; }
  if (p->work_bytes < chain_work_bytes((int64_t)p->N * ((p->H + BSH - 1) / BSH))) { rumpy_set_error("rumpy_res_chain1: work buffer too small"); return RUMPY_E_ARG; }
  if (p->fake_xcc < 0) { rumpy_set_error("rumpy_res_chain1: fake_xcc"); return RUMPY_E_ARG; }
  C1Dev d;
  d.blk = reinterpret_cast<const C1Blk*>(p->blocks); d.nblk = p->nblocks; d.N = p->N; d.H = p->H; d.W = p->W; d.sy_n = sy_n;
  d.work = (unsigned*)p->work; d.status = (unsigned*)p->status;
  d.nxcd = rumpy_device_xcds(); d.fake_xcc = p->fake_xcc;
  if (d.fake_xcc > 0) d.nxcd = d.fake_xcc < CH_MAX_XCD ? d.fake_xcc : CH_MAX_XCD;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(p->N * sy_n);
  if (p->backward) RUMPY_LAUNCH_PROBED(5, (block_chain1_kernel<3>), grid, dim3(C1T), s, d);
  else if (p->fmt == RUMPY_FMT_F16) RUMPY_LAUNCH_PROBED(5, (block_chain1_kernel<1, RUMPY_FMT_F16>), grid, dim3(C1T), s, d);
  else RUMPY_LAUNCH_PROBED(5, (block_chain1_kernel<1>), grid, dim3(C1T), s, d);
  return rumpy_check_launch("rumpy_res_chain1");
}

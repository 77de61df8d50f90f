// Weight gradient of the 64-channel 3x3 convolutions, streaming version: LDS-DMA ring + K-split over 8 waves.
//
// Same math and slab format as wgrad_kernel<4> (wgrad_mfma.hip) - dW[co][tap][ci] = sum_p dy[p][co] * x[p+tap][ci],
// pixels are the MFMA K dimension, both operands come out of LDS through ds_read_b64_tr_b16 - but built as a
// streaming kernel, because this op moves 2 bytes per 288 flops less than the forward conv and is fed from HBM:
//   * 512 threads: wave (w4, kh) accumulates D[all 64 co][ci 16*w4..] for the 9 taps (36 f32 16x16 tiles) over the
//     k-steps {2kh, 2kh+1} of every 8x16-pixel tile; the two K halves are added through LDS once, at the end.
//   * tiles arrive by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write) into a ring of 3 (4 for the tail conv) slots, two (three)
//     tiles ahead of the MFMAs, ordered by a counted s_waitcnt vmcnt + one raw s_barrier per tile.
//   * LDS image: 128-byte pixels, unpadded (an LDS-DMA wave-instruction writes 1 KiB contiguously), with the 16-byte
//     chunk index XOR-swizzled by (pixel index & 7) - applied to the per-lane SOURCE address, so the DMA stays
//     lane-linear - which makes every transposed read (8 consecutive pixels x 32 B per 32-lane half) conflict free.
//   * out-of-image pixels are fetched from a zero page; the bias gradient is one extra MFMA against a ones fragment.
#include "common.hpp"
#ifndef WGRAD_PIPE
#define WGRAD_PIPE 1    // 0: A/B - leave the order of fragment reads and MFMAs inside a tile to the compiler (rounds 1-3)
#endif
#ifndef WGRAD_A1_SPREAD
#define WGRAD_A1_SPREAD 1    // 0: A/B - the second k-step's A fragments requested in one go at unit 5 (+0.2-0.4 % spread out, same box)
#endif
#ifndef WGRAD_ABL
#define WGRAD_ABL 0     // timing-only builds, results wrong by design (tests/tools/wgrad_time.py): 1 no DMA behind the prologue (MFMAs + fragment reads alone), 2 no MFMAs (the stream alone), 4 no slab stores
#endif
#ifndef WGRAD_NST
#define WGRAD_NST 3     // ring slots of the 64-channel kernel: 3 x 40 KB, two tiles in flight (round 6: 4 slots = all 160 KB of a CU's LDS, three in flight - the stream alone
                        // 168.6 against 167.9 us, the kernel in the step 228.9 against 226.1: depth is not what it waits for)
#endif
#ifndef WGRAD_RD
#define WGRAD_RD 4      // B fragments requested ahead of their MFMAs
#endif
#ifndef WGRAD_PIN
#define WGRAD_PIN 1     // (round 6) a unit's fragment reads pinned BETWEEN its MFMAs (sched_group_barrier: MFMA, read, MFMA, read ...) instead of in front of them:
                        // 245-249 -> 232-234 us per EDSR step on one box (0: A/B; 2: two MFMAs per read 239; 3: two reads per MFMA 234; read-ahead 3 / 5 / 6 under it: 232-233)
#endif
#ifndef WGRAD_STAGGER
#define WGRAD_STAGGER 1 // 0: A/B - every wave issues its DMA pieces right behind the tile's barrier (rounds 1-2)
#endif

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr2;
typedef __attribute__((address_space(3))) unsigned char* lds_u8;

__device__ __attribute__((aligned(256))) uint4 g_zero_page[16];   // zero-initialised; source of out-of-image pixels

constexpr int DX_PIX = 192;                        // 10x18 = 180 halo pixels, rounded up to 24 DMA pieces of 8 pixels
// MT = 4 (64 output channels): 24 x pieces + 16 dy pieces = 40 pieces, 5 per wave.
// MT = 1 (tail conv, dy = [N,H,W,4]): 24 x pieces + 1 dy piece (128 px x 8 B) + 7 pieces of zeros = 32, 4 per wave; the zero
// pieces double as the all-zero channels 4..15 of the transposed dy reads.
template <int MT> struct DmaCfg {
  static constexpr int PIECES = (MT == 4) ? 40 : 32;
  static constexpr int PER_WAVE = PIECES / 8;
  static constexpr int STAGE = PIECES * 1024;
  // ring depth: MT = 4 has room for 3 slots of 40 KB (two tiles in flight); the tail conv's 32 KB slots fit 4 (three tiles in
  // flight per CU) - that kernel only streams the largest activation of the network, bytes in flight are its throughput
  static constexpr int NSTAGE = (MT == 4) ? WGRAD_NST : 4;
  static constexpr int LDS = (NSTAGE * STAGE > 4 * 36 * 1024) ? NSTAGE * STAGE : 4 * 36 * 1024;   // >= the ring and >= the final K-half exchange
};

__device__ __forceinline__ short4v tr_read2(unsigned addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr2)(size_t)addr);
}
__device__ __forceinline__ bf16x8 join8b(short4v a, short4v b) {
  union { short8v s; bf16x8 h; } c;
  c.s = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return c.h;
}
// one LDS-DMA wave-instruction: lane l copies 16 B from its own global address to LDS byte (lds_dst + 16*l).
// M0 is compiler-reserved: save, set, use and restore it inside ONE statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct DmaJob { const uint16_t* x; const uint16_t* dy; int n0, H, W, x_cstride, x_coff, dy_mode, dy_cstride, dy_coff, tiles_x, tiles_y; };

// Per-lane geometry of this wave's DMA pieces: piece k of the wave = piece index wave + 8k; k < 3 are x halo pieces (8 pixels x 128 B of the 10x18 halo),
// k >= 3 dy pieces.  A lane's pixel inside a piece is sub = lane / 8 and its swizzled 16-byte chunk (lane & 7) ^ (pixel & 7) = (lane & 7) ^ sub for every
// piece (a piece starts at a multiple of 8 pixels) - the rest is a handful of 32-bit operations per piece and tile.  Round 6: recomputed per tile from
// an OPAQUE copy of `sub` instead of kept in registers per piece (5 pieces x row, column, offset, valid = 20 registers across the tile loop - the
// compiler hoists whatever is loop-invariant - that the cross-tile fragment pipeline needs for its sixth fragment and does not have).
struct LaneGeom { int sub, swz8; };

// issue this wave's pieces of one tile into ring slot `stage_addr` (LDS byte address, wave-uniform)
template <int MT>
__device__ __forceinline__ void tile_issue(const DmaJob& j, const LaneGeom& lg, int tile, unsigned stage_addr, int wave, int lane) {
  const TileCoord tc = decode_tile(tile, j.tiles_x, j.tiles_y);
  const int n = j.n0 + tc.n;
  const int y0 = tc.ty * TH, x0 = tc.tx * TW;
  const unsigned long long zero = (unsigned long long)(uintptr_t)g_zero_page;
  int sub = lg.sub;
  asm volatile("" : "+v"(sub));
#pragma unroll
  for (int k = 0; k < DmaCfg<MT>::PER_WAVE; ++k) {
    const int piece = wave + 8 * k;                       // wave-uniform
    int r, c, coff;
    bool valid = true;
    if (k < 3) {                                          // x halo: pixels 8*piece .. 8*piece+7 of the 10x18 tile
      const int pix = piece * 8 + sub;
      const int hr = (pix * 3641) >> 16;                  // pix / HALO_W for pix < 192
      r = hr - 1; c = pix - hr * HALO_W - 1;
      coff = j.x_coff + lg.swz8;
      valid = pix < HALO_PIX;
    } else if (MT == 1) {                                 // dy4: lane l carries pixels 2l, 2l+1 (8 B each) of piece 24; others: zeros
      const int pix = 2 * lane;
      r = pix >> 4; c = pix & 15; coff = 0; valid = piece == 24;
    } else {                                              // dy: pixels 8*(piece-24) .. of the 8x16 tile
      const int pix = (piece - 24) * 8 + sub;
      r = pix >> 4; c = pix & 15;
      coff = lg.swz8 + (j.dy_mode == 0 ? j.dy_coff : 0);
    }
    const int y = y0 + r, x = x0 + c;
    const bool ok = valid & ((unsigned)y < (unsigned)j.H) & ((unsigned)x < (unsigned)j.W);
    unsigned long long base;
    unsigned e;                                           // element offset (every tensor of the path has < 2^31 elements)
    if (k < 3) {
      base = (unsigned long long)(uintptr_t)j.x;
      e = (unsigned)(((n * j.H + y) * j.W + x) * j.x_cstride + coff);
    } else if (MT == 1) {
      base = (unsigned long long)(uintptr_t)j.dy;
      e = (unsigned)(((n * j.H + y) * j.W + x) * 4);      // W is even: x+1 is in the image too
    } else {
      base = (unsigned long long)(uintptr_t)j.dy;
      if (j.dy_mode == 0) e = (unsigned)(((n * j.H + y) * j.W + x) * j.dy_cstride + coff);
      else e = (unsigned)(((n * 2 * j.H + 2 * y + (j.dy_coff >> 1)) * (2 * j.W) + 2 * x + (j.dy_coff & 1)) * 64 + coff);
    }
    const unsigned long long src = ok ? base + 2ull * e : zero;
    dma16((const void*)(uintptr_t)src, __builtin_amdgcn_readfirstlane(stage_addr + piece * 1024));
  }
}

// one job (a tile range of one layer / cin chunk / cout tile) -> its slab; called by all 512 threads of a workgroup
template <int MT>
__device__ __forceinline__ void wgrad_dma_job(const rumpy_wgrad_job* __restrict__ jp) {
  constexpr int DSTAGE = DmaCfg<MT>::STAGE;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[DmaCfg<MT>::LDS];
  DmaJob j;
  j.x = (const uint16_t*)jp->x; j.dy = (const uint16_t*)jp->dy; j.n0 = jp->n0; j.H = jp->H; j.W = jp->W;
  j.x_cstride = jp->x_cstride; j.x_coff = jp->x_coff; j.dy_mode = jp->dy_mode; j.dy_cstride = jp->dy_cstride; j.dy_coff = jp->dy_coff;
  j.tiles_x = (j.W + TW - 1) / TW; j.tiles_y = (j.H + TH - 1) / TH;
  const int t0 = jp->t0, ntiles = jp->t1 - jp->t0;
  float* slab = jp->slab;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));       // opaque per job: the per-lane read offsets and piece geometry are recomputed, not kept across the job loop (93 spills otherwise)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w4 = wave & 3, kh = wave >> 2;
  const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  const unsigned lds0 = (unsigned)(size_t)(lds_u8)lds;

  // per-lane transposed-read offsets inside a stage, for k-step 0 of this wave (rows 4kh, 4kh+1):
  //   pixel (row, col) of the first read: row = 4kh + g/2, col = 4*(g%2) + q; second read: col + 8 (same swizzle)
  const int rsel = 4 * kh + (g >> 1), col0 = 4 * (g & 1) + q;
  unsigned offA[MT], offB[9];
  {
    const int idx = rsel * TW + col0;                    // dy tile pixel index
#pragma unroll
    for (int ct = 0; ct < MT; ++ct) {
      if (MT == 4) offA[ct] = DX_PIX * 128 + idx * 128 + (((2 * ct + (p4 >> 1)) ^ (idx & 7)) << 4) + (p4 & 1) * 8;
      else offA[ct] = (p4 == 0) ? DX_PIX * 128 + idx * 8 : DX_PIX * 128 + 1024;   // channels 0..3, else the zero piece
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int ix = (rsel + ky) * HALO_W + col0 + kx;   // x halo pixel index
      offB[tap] = ix * 128 + (((2 * w4 + (p4 >> 1)) ^ (ix & 7)) << 4) + (p4 & 1) * 8;
    }
  }
  // second k-step of this wave: rows +2 -> dy index +32 (same swizzle), x index +36 (swizzle ^ 4 -> byte 64 flips)

  f32x4 acc[MT][9];
#pragma unroll
  for (int ct = 0; ct < MT; ++ct)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[ct][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = dy against a ones fragment: output-channel tile w4 on the waves (w4, *) - one MFMA per k-step on every wave (round 4; all four
  // on the w4 = 0 waves until then: 40 MFMAs per k-step there against 36 elsewhere, and the tile barrier made everybody wait for them)
  f32x4 bacc = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  constexpr int NST = DmaCfg<MT>::NSTAGE, AHEAD = NST - 1, PW = DmaCfg<MT>::PER_WAVE;
  const LaneGeom pg = {lane >> 3, ((lane & 7) ^ (lane >> 3)) * 8};
#pragma unroll
  for (int k = 0; k < AHEAD; ++k)
    if (k < ntiles) tile_issue<MT>(j, pg, t0 + k, lds0 + k * DSTAGE, wave, lane);
  int slot = 0;
  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed once at most the pieces of the tiles issued after it are still in flight (per wave; counted
    // s_waitcnt immediates), then all waves meet
    const int younger = (WGRAD_ABL == 1) ? 0 : (ntiles - 1 - t < AHEAD - 1) ? ntiles - 1 - t : AHEAD - 1;
    static_assert(AHEAD - 1 <= 3, "counted waits below: up to three younger tiles");
    if (younger >= 3) { if (PW == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); }
    else if (younger == 2) { if (PW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
    else if (younger == 1) { if (PW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // The two waves of a SIMD are (w4, kh = 0) and (w4, kh = 1).  WGRAD_STAGGER: the kh = 0 wave issues its DMA pieces of tile t + AHEAD here,
    // the kh = 1 wave between its two k-steps - the barrier above lines all eight waves up once per tile, and a wave that feeds the
    // vector-memory queue (5 pieces, 100-185 issue cycles each next to LDS reads: MI355X_MICROARCH.md) cannot feed the matrix pipe meanwhile;
    // staggered, one wave of every SIMD issues MFMAs while its sibling issues DMA.  The counted waits are unchanged (a wave's pieces of
    // tile t + AHEAD are still its youngest at the top of step t + 1), and so is every accumulation order.
    const bool more = WGRAD_ABL != 1 && t + AHEAD < ntiles;
    const int s2 = (slot + AHEAD >= NST) ? slot + AHEAD - NST : slot + AHEAD;
    const unsigned sb = lds0 + slot * DSTAGE;
    // 18 units (k-step ks, tap) of one B fragment (two transposed reads) and MT MFMAs.  Round 4: the fragment of unit u + RD is requested BEFORE the
    // MFMAs of unit u are issued (RD + 1 fragments in rotation, the second k-step's A fragments in registers of their own, requested from unit 2
    // on): left to itself the compiler put every tap's two reads right behind the previous tap's MFMAs and an lgkmcnt(0) behind them - a full LDS
    // round trip per four MFMAs, hidden only as far as the sibling wave's MFMAs reach (WGRAD_PIPE=0 restores that order; same MFMA sequence
    // either way).  261 -> 233 us in the EDSR step.  Measured around it (same box): 2 / 3 / 4 / 5 / 6 fragments ahead - 4 is where it stops paying;
    // requests behind the unit's MFMAs instead of in front: -1 %; lgkmcnt counts 15 reads at most - with 16 outstanding (unit 5 when the four A
    // fragments are requested in one go) the compiler falls back to a full wait
    auto load_a = [&](int ks, bf16x8 (&dst)[MT]) {
#pragma unroll
      for (int ct = 0; ct < MT; ++ct) {
        if (MT == 4) {
          const unsigned pa = sb + offA[ct] + ks * (32 * 128);
          dst[ct] = join8b(tr_read2(pa), tr_read2(pa + 8 * 128));
        } else {   // dy4: 8 B per pixel; lanes p4 > 0 read the zero piece (same address for both reads)
          const unsigned pa = sb + offA[ct] + ((p4 == 0) ? ks * (32 * 8) : 0);
          dst[ct] = join8b(tr_read2(pa), tr_read2(pa + ((p4 == 0) ? 8 * 8 : 0)));
        }
      }
    };
    auto load_b = [&](int u) {
      const int ks = u / 9, tap = u - 9 * ks;
      const unsigned pb = sb + (ks ? (offB[tap] ^ 64u) + 36 * 128 : offB[tap]);
      return join8b(tr_read2(pb), tr_read2(pb + 8 * 128));
    };
    constexpr int RD = WGRAD_RD;
    bf16x8 A[2][MT], Bq[RD + 1];
    load_a(0, A[0]);
#pragma unroll
    for (int u = 0; u < RD; ++u) Bq[u] = load_b(u);
    // the kh = 0 waves' DMA pieces of tile t + AHEAD go out BEHIND the first fragment requests (their round trip runs under the DMA issue:
    // measured neutral against "in front"), the kh = 1 waves' at unit 9
    if (WGRAD_PIPE) __builtin_amdgcn_sched_barrier(0);
    if (more && (!WGRAD_STAGGER || kh == 0)) tile_issue<MT>(j, pg, t0 + t + AHEAD, lds0 + s2 * DSTAGE, wave, lane);
#pragma unroll
    for (int u = 0; u < 18; ++u) {
      const int ks = u / 9, tap = u - 9 * ks;
      if (u == 9 && WGRAD_STAGGER && more && kh == 1) tile_issue<MT>(j, pg, t0 + t + AHEAD, lds0 + s2 * DSTAGE, wave, lane);
      if (u + RD < 18) Bq[(u + RD) % (RD + 1)] = load_b(u + RD);
#if WGRAD_A1_SPREAD
      if (u >= 2 && u < 2 + MT) {        // the second k-step's A fragments, one dy tile per unit
        const int ct = u - 2;
        if (MT == 4) { const unsigned pa = sb + offA[ct] + (32 * 128); A[1][ct] = join8b(tr_read2(pa), tr_read2(pa + 8 * 128)); }
        else { const unsigned pa = sb + offA[ct] + ((p4 == 0) ? (32 * 8) : 0); A[1][ct] = join8b(tr_read2(pa), tr_read2(pa + ((p4 == 0) ? 8 * 8 : 0))); }
      }
#else
      if (u == 5) load_a(1, A[1]);
#endif
      if (WGRAD_PIPE && !WGRAD_PIN) __builtin_amdgcn_sched_barrier(0);
      if (tap == 0) {                      // bias: dy tile w4 (the compiler turns these wave-uniform selects into branches with an lgkmcnt(0) in each arm - twice
                                           // per tile, at units that wait for (nearly) everything anyway; picking the operand with masks instead cost more than it saved)
        bf16x8 Ab = A[ks][0];
        if (MT == 4) Ab = (w4 == 0) ? A[ks][0] : (w4 == 1) ? A[ks][1 % MT] : (w4 == 2) ? A[ks][2 % MT] : A[ks][3 % MT];
        const f32x4 b2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ab, ones, bacc, 0, 0, 0);
        bacc = (MT == 4 || w4 == 0) ? b2 : bacc;
      }
      if (WGRAD_ABL != 2) {
#pragma unroll
        for (int ct = 0; ct < MT; ++ct)
          acc[ct][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks][ct], Bq[u % (RD + 1)], acc[ct][tap], 0, 0, 0);
      } else {                             // the stream alone: the fragment stays requested, nothing multiplies it
        asm volatile("" :: "v"(Bq[u % (RD + 1)]));
#pragma unroll
        for (int ct = 0; ct < MT; ++ct) asm volatile("" :: "v"(A[ks][ct]));
      }
#if WGRAD_PIN
      if (MT == 4) {      // (1 MFMA, 1 transposed read) x 4: the unit's two to four reads travel between its four or five MFMAs
#pragma unroll
        for (int i = 0; i < (WGRAD_PIN == 2 ? 2 : 4); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, WGRAD_PIN == 2 ? 2 : 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, WGRAD_PIN == 3 ? 2 : 1, 0);
        }
      }
#endif
      if (WGRAD_PIPE) __builtin_amdgcn_sched_barrier(0);
    }
    slot = (slot + 1 >= NST) ? 0 : slot + 1;
  }
  // ---- add the two K halves through LDS, then write the slab: [co 64][tap 9][ci 64] + [64] bias sums ----
  __syncthreads();
  f32x4* xch = reinterpret_cast<f32x4*>(lds) + (size_t)w4 * (MT * 9 * 64) + lane;
  if (kh == 1) {
#pragma unroll
    for (int ct = 0; ct < MT; ++ct)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) xch[(ct * 9 + tap) * 64] = acc[ct][tap];
  }
  __syncthreads();
  if (kh == 0) {
    const int ci = 16 * w4 + (lane & 15);
#pragma unroll
    for (int ct = 0; ct < MT; ++ct)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const f32x4 o = acc[ct][tap] + xch[(ct * 9 + tap) * 64];
        if (WGRAD_ABL == 4) { bacc += o; continue; }
#pragma unroll
        for (int e = 0; e < 4; ++e) slab[((16 * ct + 4 * g + e) * 9 + tap) * 64 + ci] = o[e];
      }
  }
  // bias sums: waves (w4, kh = 0/1) hold D[co of tile w4][*] (all 16 columns equal; MT = 1: the w4 = 0 waves only); add the halves through LDS too
  __syncthreads();
  float* bx = reinterpret_cast<float*>(lds);
  const bool bias_wave = (MT == 4) || w4 == 0;
  const int bct = (MT == 4) ? w4 : 0;
  if (bias_wave && kh == 1 && (lane & 15) == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bx[16 * bct + 4 * g + e] = bacc[e];
  }
  __syncthreads();
  if (bias_wave && kh == 0 && (lane & 15) == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) slab[16 * MT * 576 + 16 * bct + 4 * g + e] = bacc[e] + bx[16 * bct + 4 * g + e];
  }
}

// first == NULL: one job per workgroup (grid = jobs).  first != NULL: workgroup w runs jobs first[w] .. first[w+1]-1 one after the other -
// the engine cuts the cost-weighted concatenated tile sequence of all layers into equal shares, one per CU; a share may end one layer and
// begin the next (a job = a share's part of one layer, with its own slab).
template <int MT>
__global__ void __launch_bounds__(512, 2) wgrad_dma_kernel(const rumpy_wgrad_job* __restrict__ jobs, const int* __restrict__ first) {
  const int j0 = first ? first[blockIdx.x] : (int)blockIdx.x, j1 = first ? first[blockIdx.x + 1] : (int)blockIdx.x + 1;
  for (int j = j0; j < j1; ++j) {
    if (j > j0) __syncthreads();          // the previous job's K-half / bias exchange used the ring's LDS
    wgrad_dma_job<MT>(jobs + j);
  }
}

int rumpy_wgrad_dma_launch(const rumpy_wgrad_job* jobs_device, int njobs, int mt, hipStream_t s) {
  if (mt == 4) hipLaunchKernelGGL(wgrad_dma_kernel<4>, dim3(njobs), dim3(512), 0, s, jobs_device, (const int*)nullptr);
  else hipLaunchKernelGGL(wgrad_dma_kernel<1>, dim3(njobs), dim3(512), 0, s, jobs_device, (const int*)nullptr);
  return 0;
}
int rumpy_wgrad_dma_launch_shares(const rumpy_wgrad_job* jobs_device, const int* first_device, int nshares, hipStream_t s) {
  hipLaunchKernelGGL(wgrad_dma_kernel<4>, dim3(nshares), dim3(512), 0, s, jobs_device, first_device);
  return 0;
}

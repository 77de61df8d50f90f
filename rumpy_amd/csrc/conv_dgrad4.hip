// Cin = 256 -> 64 3x3 convolution as a STREAMING kernel: the data gradient of the upsampler convs (common.py:30-33 Upsampler, backward of
// conv 64 -> 256 + PixelShuffle(2): input = the gradient in the shuffled layout [N,2H,2W,64] or a plain [N,H,W,256] tensor).
//
// History of this shape (tests/tools/abl_conv4.sh, kbench.py conv4; 32 x 96 x 96, where the MFMAs need 46 us at the 1.8 GHz the chip holds):
// conv3x3_kernel<4> (conv_mfma.hip: one wave per SIMD, register-staged input, workgroup barriers) 108 us; a K-split form with two waves per
// SIMD (round 2, removed) 96 us; a streaming K-split form with LDS-DMA rings and gates instead of barriers 97 us; this kernel 93-98 us.
// Four structures, one time - because what they share is what costs: in-kernel cycle counters of this kernel (a round-2 probe build) say sweeping
// 56 % (the MFMA loop itself runs at 84 % of the matrix pipe), ISSUING the DMA 26 % (213 cycles per global_load_lds: address arithmetic plus
// the wait for a slot in the vector-memory queue - the kernel with its MFMAs compiled out still takes 60 us, whatever the prefetch depth; the DMA
// issue, waits and hand-over ALONE run the gather in 31 us = 4.9 TB/s, tests/tools/overlap/gather_probe.hip, round 4: it is issue time, not bandwidth), waiting for a stage 10 %, epilogue 8 %.  A wave that feeds the memory pipe
// cannot feed the matrix pipe meanwhile, and every wave here must do both (the filter fills its registers: no spare wave).  Interleaving
// the DMA issue with the MFMAs made it worse (110 us).  Kept because it is the simplest of the four and 0.5 % faster on the step.
// Structure: ONE wave per SIMD (512 registers: the whole filter AND three fragment sets), but a wave that does not wait for itself:
//   * input by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write phase) into a ring of 6 chunk stages, FOUR stages (one whole
//     tile) ahead of the MFMAs; waits are counted (s_waitcnt vmcnt(N), N = DMA pieces issued after the awaited stage: loads return in
//     order, so the bound holds whatever the stores do);
//   * stage image = block_common.hpp's: 128-byte pixels, unpadded (a DMA wave-instruction writes 1 KiB contiguously), 16-byte chunk index
//     XOR-ed with (pixel & 7) on the SOURCE address; 10 x 18 halo pixels = 23 pieces of 8 pixels; out-of-image pixels come from a zero page;
//   * fragment reads of units j + 1, j + 2 (tap column, channel half, row half) are issued BEFORE the 12 MFMAs of unit j (block_sweep's pipelining);
//   * no workgroup barrier: the four waves meet on one LDS counter per stage (landed = every wave's pieces have arrived AND every wave is done
//     with the stage whose slot the next DMA overwrites); no K split, so no partial-sum exchange either: same MFMA order per accumulator as
//     conv3x3_kernel<4> (chunks 0 .. 3 in turn): bitwise that kernel (tests/test_kernels_gpu.py).
#include "block_common.hpp"
#include <cstdlib>
#include <type_traits>
static inline int d4_cdiv(int a, int b) { return (a + b - 1) / b; }

// Tile height as a parameter (round 4): 8 rows, or 6 where that fills the chip better - the 32 x 48 x 48 stage of the x4 upsampler is 576 tiles of
// 8 rows = three rounds of 192 workgroups for 2.25 rounds of work, and exactly three rounds of 256 six-row tiles (35.1 -> 27-28 us).  Every output
// element sees the same MFMA sequence either way (chunk, tap column, channel half, tap row): bitwise equal.
template <int TH_> struct D4Geo {
  static constexpr int TR = TH_;                         // output rows per tile
  static constexpr int HR = TH_ / 2;                     // rows per pass
  static constexpr int HPIX = (TH_ + 2) * HALO_W;        // halo pixels: 180 / 144
  static constexpr int PIECES = (HPIX + 7) / 8;          // DMA pieces of 8 pixels: 23 / 18
  static constexpr int STAGE = PIECES * 1024;
  static constexpr int PW = (PIECES + 3) / 4;            // DMA pieces per wave and stage (the surplus repeats the last piece): 6 / 5
};
constexpr int D4_NST = 6;                             // ring slots: the stage in use, four in flight, one being refilled
#ifndef D4_AHEAD
#define D4_AHEAD 4
#endif
#ifndef D4_XCD
#define D4_XCD 0        // 1: A/B - the tiles of a round dealt out in contiguous runs per XCD (common.hpp xcd_strip), so that neighbouring tiles, which
#endif                  // share 41 % of their halo pixels, meet in ONE L2.  Measured SLOWER (round 3, same box, kbench.py conv4: 32x96x96 114.0-115.1 against
                        // 101.7-102.5 us, 32x48x48 39.3-40.2 against 36.9-37.7): dealt round-robin, a round's 256 tiles load all eight L2s and their memory
                        // channels evenly at every moment; in runs, each XCD streams one compact region and the halo lines it saves were L2 hits of the
                        // Infinity Cache anyway (profiles/r03_conv4_xcd_ab.txt)
#ifndef D4_PIN
#define D4_PIN 1        // (round 6) a unit's LDS reads pinned BETWEEN its MFMAs (0: in front of them; same box: conv4dt 121.5 -> 117.5 us, conv4d unchanged; bitwise)
#endif
#ifndef D4_RD
#define D4_RD 2         // fragment sets read ahead of the MFMAs (units of HR + 2 reads / 3 HR MFMAs)
#endif

__device__ __attribute__((aligned(256))) uint4 g_zero_page4[16];
typedef __attribute__((address_space(3))) unsigned char* d4_lds_u8;

__device__ __forceinline__ void d4_dma16(const void* gsrc, unsigned lds_dst) {     // wgrad_dma.hip::dma16
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void d4_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }


template <int FMT, int TH_ = TH>      // element format of activations and filters (RUMPY_FMT_F16: evaluation plans of the wide nets); tile rows
__global__ void __launch_bounds__(256, 1) conv4d_kernel(ConvDev a) {
  typedef D4Geo<TH_> G;
  constexpr int D4_PIECES = G::PIECES, D4_STAGE = G::STAGE, D4_PW = G::PW, HR = G::HR;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[D4_NST * D4_STAGE];      // 141,312 / 110,592 B
  __shared__ unsigned landed;            // stages landed: 4 arrivals each
  const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, g = lane >> 4;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const int tile0 = blockIdx.x, tstride = (int)gridDim.x;
  if (tile0 >= ntiles) return;
  const int nt = (ntiles - tile0 + tstride - 1) / tstride;
  const int nstage = 4 * nt;
  // round `it` covers tiles [it * grid, (it + 1) * grid); workgroup b sits on XCD b % 8 (D4_XCD above)
  auto tile_of = [&](int it) {
    const int base = it * tstride, left = ntiles - base;
    return base + (D4_XCD ? xcd_strip(tile0, left < tstride ? left : tstride) : tile0);
  };
  const unsigned ring = (unsigned)(size_t)(d4_lds_u8)lds;
  if (tid == 0) landed = 0u;

  // output tile ct of cout_tiles (round 2: the 256 -> 256 convs of EDSR at the reference's shipped width run here too, with the epilogues of
  // conv3x3_kernel<4>: bias, ReLU, scale, ReLU mask, two residuals); filter image [ct][chunk][wave][18][lane]
  const int ct = blockIdx.y, cstr = 64 * a.cout_tiles;
  bf16x8 F[4][18];
  {
    const uint4* wp = a.w + (size_t)ct * 4 * (4 * 18 * 64);
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
      for (int s = 0; s < 18; ++s) F[ch][s] = as_bf16x8(wp[((ch * 4 + q) * 18 + s) * 64 + lane]);
  }

  // this wave's DMA pieces of a stage: piece q + 4k; lane = (pixel sub 0..7, 16-byte slot 0..7); source chunk = slot ^ (pixel & 7)
  const unsigned long long zero = (unsigned long long)(uintptr_t)g_zero_page4;
  // per-lane geometry of the wave's six pieces, fixed for the kernel: halo row / column of the lane's pixel and its source chunk offset
  int pr[D4_PW], pc[D4_PW];
#pragma unroll
  for (int k = 0; k < D4_PW; ++k) {
    const int piece = (q + 4 * k < D4_PIECES) ? q + 4 * k : D4_PIECES - 1;
    const int pix = piece * 8 + (lane >> 3);
    const int r = pix / HALO_W, c = pix - r * HALO_W;
    pr[k] = (pix < G::HPIX) ? r - 1 : -100000;                     // a pixel slot past the halo is out of the image for every tile
    pc[k] = ((c - 1) << 8) | (((lane & 7) ^ (pix & 7)) * 8);         // column (may be -1: arithmetic shift) and source chunk offset in elements
  }
  // piece k of stage u (tile tile0 + (u / 4) * tstride, input chunk u & 3): ~12 VALU + the DMA; called one piece at a time from inside the
  // MFMA stream, where a lone wave per SIMD has issue slots to spare (a stage's six pieces issued in one go at the top of a chunk cost
  // 190 VALU instructions there: a third of the chunk's MFMA time with nothing to overlap them)
  auto issue_piece = [&](int u, int k) {
    const TileCoord tc = decode_tile(tile_of(u >> 2), a.tiles_x, a.tiles_y);
    const int ch = u & 3;
    const unsigned dst = ring + (unsigned)(u % D4_NST) * D4_STAGE;
    const int piece = (q + 4 * k < D4_PIECES) ? q + 4 * k : D4_PIECES - 1;
    const int y = tc.ty * TH_ + pr[k], x = tc.tx * TW + (pc[k] >> 8);
    const int sch = pc[k] & 255;
    const bool ok = ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
    unsigned e;
    if (a.in_mode == 0) e = (unsigned)(((tc.n * a.H + y) * a.W + x) * 256 + ch * 64 + sch);
    else e = (unsigned)(((tc.n * 2 * a.H + 2 * y + (ch >> 1)) * (2 * a.W) + 2 * x + (ch & 1)) * 64 + sch);
    const unsigned long long src = ok ? (unsigned long long)(uintptr_t)a.x + 2ull * e : zero;
    d4_dma16((const void*)(uintptr_t)src, __builtin_amdgcn_readfirstlane(dst + piece * 1024));
  };
  auto issue = [&](int u) {
#pragma unroll
    for (int k = 0; k < D4_PW; ++k) issue_piece(u, k);
  };
  // fragment read bases (block_common.hpp::sweep_bases with an 18-pixel row): class d = (2 r + kx) & 7 of window row r, tap column kx
  unsigned off[8][2];
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int h = 0; h < 2; ++h) off[d][h] = (unsigned)(px * 128 + (((4 * h + g) ^ ((px + d) & 7)) << 4));

  __syncthreads();                        // counter zeroed
#pragma unroll
  for (int u = 0; u < D4_AHEAD; ++u)
    if (u < nstage) issue(u);

  const int c0 = 16 * q + 4 * g;
  float bj[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
    const float4 b4 = *reinterpret_cast<const float4*>(a.bias + ct * 64 + c0);
    bj[0] = b4.x; bj[1] = b4.y; bj[2] = b4.z; bj[3] = b4.w;
  }
  for (int it = 0; it < nt; ++it) {
    const TileCoord tc = decode_tile(tile_of(it), a.tiles_x, a.tiles_y);
    const int xx = tc.tx * TW + px;
    f32x4 acc[TH_];
#pragma unroll
    for (int r = 0; r < TH_; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const int u = 4 * it + ch;
      // stage u has landed once at most the pieces issued after it are outstanding (stages u+1 .. u+3, fewer at the end); then the waves meet
      const int younger = (nstage - 1 - u < D4_AHEAD - 1) ? nstage - 1 - u : D4_AHEAD - 1;
      if (younger >= 4) d4_wait<4 * D4_PW>(); else if (younger == 3) d4_wait<3 * D4_PW>(); else if (younger == 2) d4_wait<2 * D4_PW>(); else if (younger == 1) d4_wait<D4_PW>(); else d4_wait<0>();
      gate_arrive(&landed, lane);
      gate_wait(&landed, 4u * (unsigned)(u + 1));
      if (u + D4_AHEAD < nstage) issue(u + D4_AHEAD);   // into a slot every wave is past (stage u - 2's)
      const unsigned char* cur = lds + (u % D4_NST) * D4_STAGE;
      // 12 units (tap column, channel half, row half) of 6 fragment reads + 12 MFMAs; the reads of unit j + 1 travel under the MFMAs of unit j
      bf16x8 I[D4_RD + 1][HR + 2];
      auto load_unit = [&](int j, bf16x8 (&dst)[HR + 2]) {
        const int kx = j >> 2, half = (j >> 1) & 1, pass = j & 1;
#pragma unroll
        for (int r = 0; r < HR + 2; ++r)
          dst[r] = *reinterpret_cast<const bf16x8*>(cur + off[(2 * (HR * pass + r) + kx) & 7][half] + ((HR * pass + r) * HALO_W + kx) * 128);
      };
#pragma unroll
      for (int j = 0; j < D4_RD; ++j) load_unit(j, I[j]);
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        if (j + D4_RD < 12) load_unit(j + D4_RD, I[(j + D4_RD) % (D4_RD + 1)]);
#if !D4_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
        const int kx = j >> 2, half = (j >> 1) & 1, pass = j & 1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int r = 0; r < HR; ++r)
            acc[HR * pass + r] = mfma16<FMT>(F[ch][(ky * 3 + kx) * 2 + half], I[j % (D4_RD + 1)][r + ky], acc[HR * pass + r]);
#if D4_PIN
        // round 6: one wave per SIMD - nobody else feeds the matrix pipe while this wave issues the HR + 2 fragment reads of a later unit in front of its MFMAs;
        // pinned between them (MFMA x PER, read) they cost next to nothing (block_common.hpp::BLOCK_PIN)
        if (j + D4_RD < 12) {
          constexpr int PER = (3 * HR) / (HR + 2) > 0 ? (3 * HR) / (HR + 2) : 1;
#pragma unroll
          for (int i = 0; i < HR + 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
    // ---- epilogue: lane holds channels c0 .. c0+3 of pixel (row r, column px); the residual operand is read here (plain loads: the
    // compiler's wait for them drains the DMA queue - everything older - which costs one stage's slack once per tile) ----
#pragma unroll
    for (int r = 0; r < TH_; ++r) {
      const int y = tc.ty * TH_ + r;
      if (y < a.H && xx < a.W) {
        const size_t e = ((size_t)(tc.n * a.H + y) * a.W + xx) * cstr + ct * 64 + c0;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                     // conv3x3_kernel<4>'s epilogue, operation for operation
          v[j] = acc[r][j] + bj[j];
          if (a.relu) v[j] = relu_f32(v[j]);
          v[j] *= a.scale;
        }
        if (a.mask) {
          float m[4];
          unpack4<FMT>(*reinterpret_cast<const uint2*>(a.mask + e), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (m[j] > 0.f) ? v[j] : 0.f;
        }
        if (a.res1) {
          float m[4];
          unpack4<FMT>(*reinterpret_cast<const uint2*>(a.res1 + e), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        if (a.res2) {
          float m[4];
          unpack4<FMT>(*reinterpret_cast<const uint2*>(a.res2 + e), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        *reinterpret_cast<uint2*>(a.out + e) = pack4<FMT>(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

// grid_x <= 0: choose tile height and grid for `cap` resident workgroups per output tile (a workgroup is alone on its CU): fewest
// rounds x (rows + 2) - a tile costs its output rows plus the two halo rows it also fetches
int rumpy_conv4d_launch(ConvDev d, int grid_x, int cap, hipStream_t s, int fmt) {
  int th = TH;
  if (fmt != RUMPY_FMT_F16 && !getenv("RUMPY_CONV4_TH8")) {
    const long long t8 = (long long)d.N * d.tiles_x * d4_cdiv(d.H, 8), t6 = (long long)d.N * d.tiles_x * d4_cdiv(d.H, 6);
    if (d4_cdiv((int)t6, cap) * 8 < d4_cdiv((int)t8, cap) * 10) th = 6;
  }
  d.tiles_y = d4_cdiv(d.H, th);
  const int ntiles = d.N * d.tiles_x * d.tiles_y;
  int g2 = grid_x > 0 ? grid_x : cap;
  const int rounds = d4_cdiv(ntiles, g2);
  g2 = d4_cdiv(ntiles, rounds);
  // several output tiles: workgroup (x, ct) has linear id ct * g2 + x and lands on XCD id % 8 - with g2 a multiple of 8 the cout_tiles
  // workgroups that read the same input tiles share an XCD, i.e. an L2 (PMC: 118 MB fetched per 256 -> 256 launch at 16 x 48 x 48 with
  // g2 = 58, four times the input; profiles/r02_pmc_wide.md)
  if (grid_x <= 0 && d.cout_tiles > 1 && g2 >= 8) g2 = ((g2 + 7) & ~7) <= cap ? ((g2 + 7) & ~7) : (g2 & ~7);
  if (fmt == RUMPY_FMT_F16) hipLaunchKernelGGL(conv4d_kernel<RUMPY_FMT_F16>, dim3(g2, d.cout_tiles), dim3(256), 0, s, d);
  else if (th == 6) hipLaunchKernelGGL((conv4d_kernel<RUMPY_FMT_BF16, 6>), dim3(g2, d.cout_tiles), dim3(256), 0, s, d);
  else hipLaunchKernelGGL(conv4d_kernel<RUMPY_FMT_BF16>, dim3(g2, d.cout_tiles), dim3(256), 0, s, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Round 5: the tail conv's data gradient INSIDE this kernel (rumpy_conv4d_tail).  The last upsampler stage's data gradient used to read the
// 151 MB tensor dx = conv^T_tail(dy4) that rumpy_tail_dgrad had just written (64 us, write-bound, + this kernel's 26 % of DMA issue time for
// reading it back).  dx is a 3x3 conv over FOUR channels: a workgroup that holds the (2 TH + 6) x 40 pixel window of dy4 (7 KB by LDS-DMA,
// one tile ahead) makes each stage image - the 64 channels of one PixelShuffle phase over its 10 x 18 halo pixels - with 24 MFMAs per wave
// (tail_dgrad_kernel's two per 16 pixels x 16 channels, same operands, same order: the bf16 values are bitwise that kernel's) and writes them
// into the ring in the DMA's layout; the stage sweep below is conv4d_kernel's.  dx still goes to HBM - the weight gradient of the upsampler
// conv reads it - as whole 128-byte lines from the stage image, each pixel by the one tile that owns it.  No input DMA ring: a stage is
// produced one sweep ahead (4 slots), the waves meet on an LDS counter per stage as above.
// ------------------------------------------------------------------------------------------------------
typedef unsigned int tail_u32x4_d4 __attribute__((ext_vector_type(4)));
typedef unsigned int d4_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) d4_u32x2* d4_lds_u32x2;
struct Conv4TDev { const uint16_t* dy4; const uint4* wt; uint16_t* dx; const uint4* w; uint16_t* out; int N, H, W, tiles_x, tiles_y; };

template <int TH_>
__global__ void __launch_bounds__(256, 1) conv4dt_kernel(Conv4TDev a) {
  typedef D4Geo<TH_> G;
  constexpr int HR = G::HR;
  constexpr int STG = ((G::HPIX + 15) / 16) * 16 * 128;     // 24,576 / 18,432 B
  constexpr int DYR = 2 * (TH_ + 2) + 2, DYW = 40;          // dy4 window: rows 2 (ty TH - 1) - 1 .. , columns 32 tx - 4 .. (16-byte aligned pairs)
  constexpr int DYP = (DYR * (DYW / 2) + 63) / 64;          // DMA pieces of 64 x 16 bytes: 7 / 5
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * STG];
  __shared__ __attribute__((aligned(1024))) unsigned char sdy[2][DYP * 1024];
  __shared__ unsigned landed, dyl;
  const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, g = lane >> 4;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const int tile0 = blockIdx.x, tstride = (int)gridDim.x;
  if (tile0 >= ntiles) return;
  const int nt = (ntiles - tile0 + tstride - 1) / tstride;
  if (tid == 0) { landed = 0u; dyl = 0u; }

  bf16x8 F[4][18];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch)
#pragma unroll
    for (int s = 0; s < 18; ++s) F[ch][s] = as_bf16x8(a.w[((ch * 4 + q) * 18 + s) * 64 + lane]);
  const bf16x8 T0 = as_bf16x8(a.wt[(q * 2 + 0) * 64 + lane]), T1 = as_bf16x8(a.wt[(q * 2 + 1) * 64 + lane]);

  const unsigned long long zero = (unsigned long long)(uintptr_t)g_zero_page4;
  const unsigned dybase = (unsigned)(size_t)(d4_lds_u8)&sdy[0][0];
  // the dy4 window of tile t -> sdy[t & 1]: wave q moves pieces q and q + 4; lane = 16-byte unit (row, column pair) in row-major order
  auto dy_issue = [&](int t) {
    const TileCoord tc = decode_tile(tile0 + t * tstride, a.tiles_x, a.tiles_y);
    const int Y0 = 2 * (tc.ty * TH_ - 1) - 1, X0 = 32 * tc.tx - 4;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int piece = q + 4 * k;
      if (piece < DYP) {                                      // wave-uniform
        const int i = piece * 64 + lane, row = i / (DYW / 2), cp = i - row * (DYW / 2);
        const int Y = Y0 + row, X = X0 + 2 * cp;
        const bool ok = (row < DYR) & ((unsigned)Y < (unsigned)(2 * a.H)) & ((unsigned)X < (unsigned)(2 * a.W));
        const unsigned long long src = ok ? (unsigned long long)(uintptr_t)a.dy4 + 8ull * (unsigned)((tc.n * 2 * a.H + Y) * (2 * a.W) + X) : zero;
        d4_dma16((const void*)(uintptr_t)src, __builtin_amdgcn_readfirstlane(dybase + (unsigned)((t & 1) * DYP + piece) * 1024u));
      }
    }
  };
  // taps of this lane group in the tail filter's K order (tail_dgrad_kernel): MFMA 1 = taps 2 g, 2 g + 1; MFMA 2 = tap 8 (g = 0 only)
  const int ta = 2 * g, tb = 2 * g + 1;
  const int offa = ((ta / 3) * DYW + ta % 3) * 8, offb = ((tb / 3) * DYW + tb % 3) * 8, offc = (2 * DYW + 2) * 8;
  // Stage image of phase PH (= input chunk of the sweep: pixels (2 y + PH / 2, 2 x + PH % 2)) of tile t into ring slot PH, one group of 16 pixels at a
  // time and in three pieces - operand reads, the two MFMAs, the write - so that the sweep below can carry a group per unit: done in one go
  // (reads -> wait -> MFMA -> MFMA -> result -> write, twelve times) a stage image took longer than the sweep it feeds.
  // Groups: one per halo row (columns 0 .. 15: row and column known at compile time / per lane, so every LDS address is a lane constant
  // plus an immediate - groups cut out of the flattened pixel index cost 45 VALU instructions each in divisions and swizzles), then the two
  // columns 16, 17 of all rows in NEDGE more.
  constexpr int NROW = TH_ + 2, NEDGE = (2 * NROW + 15) / 16, NGRP = NROW + NEDGE;        // 10 + 2 / 8 + 1
  const unsigned sdy0 = (unsigned)(size_t)(d4_lds_u8)&sdy[0][0], ring0 = (unsigned)(size_t)(d4_lds_u8)lds;
  const unsigned chunkq = (unsigned)(2 * q + (g >> 1));
  const unsigned rd0 = sdy0 + (unsigned)((2 * px + 1) * 8);
  unsigned W4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) W4[k] = ring0 + (unsigned)(px * 128 + ((chunkq ^ (unsigned)((2 * k + px) & 7)) << 4) + (g & 1) * 8);
  unsigned erd[NEDGE], ewr[NEDGE];
  int er[NEDGE], ec[NEDGE];
#pragma unroll
  for (int e = 0; e < NEDGE; ++e) {
    const int pe = 16 * e + px;
    er[e] = pe >> 1; ec[e] = 16 + (pe & 1);
    // lanes past the last row (the second edge group's upper 12 pixel slots): operands from any valid address, result (zeros) into the image's spare
    // pixel slots behind the halo tile - NOT onto a real pixel, which a lane of the same write instruction fills
    const int rc = er[e] < NROW ? er[e] : NROW - 1, pE = er[e] < NROW ? rc * HALO_W + ec[e] : G::HPIX + (pe & 7);
    erd[e] = sdy0 + (unsigned)((2 * rc * DYW + 2 * ec[e] + 1) * 8);
    ewr[e] = ring0 + (unsigned)(pE * 128 + ((chunkq ^ (unsigned)(pE & 7)) << 4) + (g & 1) * 8);
  }
  struct PGroup { uint2 b0a, b0b, b1; };
  struct PTile { unsigned dyoff; int ty, tx; bool colok; bool eok[NEDGE]; };       // per produced tile: window slot, masks
  auto p_tile = [&](int t, PTile& T_) {
    const TileCoord tn = decode_tile(tile0 + t * tstride, a.tiles_x, a.tiles_y);
    T_.dyoff = (unsigned)((t & 1) * DYP * 1024); T_.ty = tn.ty; T_.tx = tn.tx;
    T_.colok = (unsigned)(tn.tx * TW + px - 1) < (unsigned)a.W;
#pragma unroll
    for (int e = 0; e < NEDGE; ++e)
      T_.eok[e] = (er[e] < NROW) & ((unsigned)(tn.ty * TH_ + er[e] - 1) < (unsigned)a.H) & ((unsigned)(tn.tx * TW + ec[e] - 1) < (unsigned)a.W);
  };
  auto lds_rd8 = [&](unsigned addr) { const d4_u32x2 v = *(d4_lds_u32x2)(size_t)addr; return make_uint2(v.x, v.y); };
  auto p_read = [&](const PTile& T_, auto PHc, auto Jc, PGroup& G_) {
    constexpr int PH = decltype(PHc)::value, j = decltype(Jc)::value;
    constexpr int rowimm = (j < NROW) ? ((2 * j + (PH >> 1)) * DYW + (PH & 1)) * 8 : ((PH >> 1) * DYW + (PH & 1)) * 8;
    const unsigned b = ((j < NROW) ? rd0 : erd[j < NROW ? 0 : j - NROW]) + T_.dyoff + rowimm;
    G_.b0a = lds_rd8(b + offa);
    G_.b0b = lds_rd8(b + offb);
    G_.b1 = lds_rd8(b + offc);          // every lane reads (no exec-masked read in the stream); lanes g > 0 drop it below
  };
  auto p_mfma = [&](const PGroup& G_, f32x4& acc) {
    union { uint2 u[2]; bf16x8 v; } B0, B1;
    B0.u[0] = G_.b0a; B0.u[1] = G_.b0b;
    B1.u[0] = (g == 0) ? G_.b1 : make_uint2(0, 0);
    B1.u[1] = make_uint2(0, 0);
    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T0, B0.v, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T1, B1.v, acc, 0, 0, 0);
  };
  auto p_write = [&](const PTile& T_, auto PHc, auto Jc, const f32x4& acc) {
    constexpr int PH = decltype(PHc)::value, j = decltype(Jc)::value;
    bool inside;
    unsigned dst;
    if (j < NROW) {
      inside = T_.colok & ((unsigned)(T_.ty * TH_ + j - 1) < (unsigned)a.H);
      dst = W4[j & 3] + (unsigned)(PH * STG + j * HALO_W * 128);
    } else {
      inside = T_.eok[j < NROW ? 0 : j - NROW];
      dst = ewr[j < NROW ? 0 : j - NROW] + (unsigned)(PH * STG);
    }
    uint2 v = pack4_bf16(acc[0], acc[1], acc[2], acc[3]);
    v.x = inside ? v.x : 0u; v.y = inside ? v.y : 0u;      // the conv's zero padding (NOT the tail gradient of a pixel outside the image)
    *(d4_lds_u32x2)(size_t)dst = (d4_u32x2){v.x, v.y};
  };
  auto produce = [&](int t, auto PHc) {      // a whole stage image in one go: the first one of the kernel only
    PTile T_;
    p_tile(t, T_);
    PGroup G_; f32x4 pa;
#define D4T_ONE(J) if (J < NGRP) { p_read(T_, PHc, std::integral_constant<int, (J < NGRP ? J : 0)>(), G_); p_mfma(G_, pa); p_write(T_, PHc, std::integral_constant<int, (J < NGRP ? J : 0)>(), pa); }
    D4T_ONE(0) D4T_ONE(1) D4T_ONE(2) D4T_ONE(3) D4T_ONE(4) D4T_ONE(5) D4T_ONE(6) D4T_ONE(7) D4T_ONE(8) D4T_ONE(9) D4T_ONE(10) D4T_ONE(11)
#undef D4T_ONE
  };
  // dx of the pixels this tile owns, phase PH: whole 128-byte lines out of the stage image, TH_ / 2 pieces of 16 bytes per thread - piece i is read in
  // unit i of the sweep and leaves in unit i + 1
  constexpr int NDX = TH_ * TW * 8 / 256;
  auto dx_read = [&](auto PHc, int i, uint4& v) {
    constexpr int PH = decltype(PHc)::value;
    int tido = tid;
    asm volatile("" : "+v"(tido));
    const int e = tido + 256 * i, pl = e >> 3, c = e & 7, rr = pl >> 4, cc = pl & 15;
    const int p = (rr + 1) * HALO_W + cc + 1;
    v = *reinterpret_cast<const uint4*>(lds + PH * STG + p * 128 + ((c ^ (p & 7)) << 4));
  };
  auto dx_store = [&](const TileCoord& tc, auto PHc, int i, const uint4& v) {
    constexpr int PH = decltype(PHc)::value;
    int tido = tid;
    asm volatile("" : "+v"(tido));
    const int e = tido + 256 * i, pl = e >> 3, c = e & 7, rr = pl >> 4, cc = pl & 15;
    const int y = tc.ty * TH_ + rr, x = tc.tx * TW + cc;
    if (y < a.H && x < a.W)      // (plain instead of non-temporal stores: the same time; without the stores at all: -5 us of 118)
      __builtin_nontemporal_store((tail_u32x4_d4){v.x, v.y, v.z, v.w},
          reinterpret_cast<tail_u32x4_d4*>(a.dx + ((size_t)((tc.n * 2 * a.H + 2 * y + (PH >> 1)) * (2 * a.W) + 2 * x + (PH & 1))) * 64 + c * 8));
  };
  unsigned off[8][2];
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int h = 0; h < 2; ++h) off[d][h] = (unsigned)(px * 128 + (((4 * h + g) ^ ((px + d) & 7)) << 4));

  __syncthreads();                        // counters zeroed
  dy_issue(0);
  d4_wait<0>();
  gate_arrive(&dyl, lane);
  gate_wait(&dyl, 4u);
  produce(0, std::integral_constant<int, 0>());
  gate_arrive(&landed, lane);

  const int c0 = 16 * q + 4 * g;
  for (int it = 0; it < nt; ++it) {
    const TileCoord tc = decode_tile(tile0 + it * tstride, a.tiles_x, a.tiles_y);
    const int xx = tc.tx * TW + px;
    if (it + 1 < nt) dy_issue(it + 1);    // into the window of tile it - 1: every wave is past its last stage image (it arrived for it)
    f32x4 acc[TH_];
#pragma unroll
    for (int r = 0; r < TH_; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto stage = [&](auto CHc) {
      constexpr int ch = decltype(CHc)::value;
      typedef std::integral_constant<int, (ch + 1) & 3> NPc;      // the phase whose stage image this sweep carries along
      const int u = 4 * it + ch;
      const bool more = ch < 3 || it + 1 < nt;               // wave-uniform
      PTile PT;
      p_tile(more && ch == 3 ? it + 1 : it, PT);
      if (ch == 3 && more) {
        d4_wait<0>();                     // the next tile's dy4 window (issued four sweeps ago); the dx stores in flight are a sweep old
        gate_arrive(&dyl, lane);
        gate_wait(&dyl, 4u * (unsigned)(it + 2));
      }
      gate_wait(&landed, 4u * (unsigned)(u + 1));
      const unsigned char* cur = lds + ch * STG;
      bf16x8 I[D4_RD + 1][HR + 2];
      auto load_unit = [&](int j, bf16x8 (&dst)[HR + 2]) {
        const int kx = j >> 2, half = (j >> 1) & 1, pass = j & 1;
#pragma unroll
        for (int r = 0; r < HR + 2; ++r)
          dst[r] = *reinterpret_cast<const bf16x8*>(cur + off[(2 * (HR * pass + r) + kx) & 7][half] + ((HR * pass + r) * HALO_W + kx) * 128);
      };
      PGroup PG[2];
      f32x4 pacc[2];
      uint4 dxv;
#pragma unroll
      for (int j = 0; j < D4_RD; ++j) load_unit(j, I[j]);
      // unit j: the fragment reads of unit j + 2, the operand reads of group j of the next stage image, the write of group j - 1 and piece j of this
      // image's dx rows travel under the 12 MFMAs of unit j; the group's own two MFMAs follow them (their operands have landed by then)
      auto unit = [&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        if (j + D4_RD < 12) load_unit(j + D4_RD, I[(j + D4_RD) % (D4_RD + 1)]);
        if (more && j < NGRP) p_read(PT, NPc(), std::integral_constant<int, (j < NGRP ? j : 0)>(), PG[j & 1]);
        if (more && j >= 1 && j - 1 < NGRP) p_write(PT, NPc(), std::integral_constant<int, (j >= 1 && j - 1 < NGRP ? j - 1 : 0)>(), pacc[(j - 1) & 1]);
        if (j >= 1 && j - 1 < NDX) dx_store(tc, CHc, j - 1, dxv);
        if (j < NDX) dx_read(CHc, j, dxv);
#if !D4_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
        const int kx = j >> 2, half = (j >> 1) & 1, pass = j & 1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int r = 0; r < HR; ++r)
            acc[HR * pass + r] = mfma16<RUMPY_FMT_BF16>(F[ch][(ky * 3 + kx) * 2 + half], I[j % (D4_RD + 1)][r + ky], acc[HR * pass + r]);
#if D4_PIN
        {      // (the unit's LDS reads - fragments of unit j + 2, the stage image's operands, a dx piece - between its 3 HR MFMAs: MFMA x 2, read, ...)
#pragma unroll
          for (int i = 0; i < (3 * HR) / 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (more && j < NGRP) p_mfma(PG[j & 1], pacc[j & 1]);
      };
      unit(std::integral_constant<int, 0>()); unit(std::integral_constant<int, 1>()); unit(std::integral_constant<int, 2>());
      unit(std::integral_constant<int, 3>()); unit(std::integral_constant<int, 4>()); unit(std::integral_constant<int, 5>());
      unit(std::integral_constant<int, 6>()); unit(std::integral_constant<int, 7>()); unit(std::integral_constant<int, 8>());
      unit(std::integral_constant<int, 9>()); unit(std::integral_constant<int, 10>()); unit(std::integral_constant<int, 11>());
      if (more) {
        if (NGRP == 12) p_write(PT, NPc(), std::integral_constant<int, NGRP - 1>(), pacc[1]);
        gate_arrive(&landed, lane);
      }
    };
    stage(std::integral_constant<int, 0>());
    stage(std::integral_constant<int, 1>());
    stage(std::integral_constant<int, 2>());
    stage(std::integral_constant<int, 3>());
#pragma unroll
    for (int r = 0; r < TH_; ++r) {
      const int y = tc.ty * TH_ + r;
      if (y < a.H && xx < a.W)
        *reinterpret_cast<uint2*>(a.out + ((size_t)(tc.n * a.H + y) * a.W + xx) * 64 + c0) = pack4<RUMPY_FMT_BF16>(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    }
  }
}

extern "C" int rumpy_conv4d_tail(const rumpy_conv4d_tail_args* p, void* stream) {
  if (!p || !p->dy4 || !p->w_tail || !p->dx || !p->w || !p->out || p->N <= 0 || p->H <= 0 || p->W <= 0) {
    rumpy_set_error("rumpy_conv4d_tail: bad argument");
    return RUMPY_E_ARG;
  }
  if ((long long)p->N * 2 * p->H * 2 * p->W * 64 >= (1ll << 31)) { rumpy_set_error("rumpy_conv4d_tail: tensor beyond 2^31 elements"); return RUMPY_E_ARG; }
  Conv4TDev d;
  d.dy4 = (const uint16_t*)p->dy4; d.wt = (const uint4*)p->w_tail; d.dx = (uint16_t*)p->dx; d.w = (const uint4*)p->w; d.out = (uint16_t*)p->out;
  d.N = p->N; d.H = p->H; d.W = p->W; d.tiles_x = d4_cdiv(p->W, TW);
  const int cap = rumpy_device_cus() > 0 ? rumpy_device_cus() : 1;
  int th = 8;
  {
    const long long t8 = (long long)d.N * d.tiles_x * d4_cdiv(d.H, 8), t6 = (long long)d.N * d.tiles_x * d4_cdiv(d.H, 6);
    if (d4_cdiv((int)t6, cap) * 8 < d4_cdiv((int)t8, cap) * 10) th = 6;
  }
  d.tiles_y = d4_cdiv(d.H, th);
  const int ntiles = d.N * d.tiles_x * d.tiles_y;
  int g2 = p->grid_x > 0 ? p->grid_x : cap;
  g2 = d4_cdiv(ntiles, d4_cdiv(ntiles, g2));
  if (th == 6) hipLaunchKernelGGL(conv4dt_kernel<6>, dim3(g2), dim3(256), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(conv4dt_kernel<8>, dim3(g2), dim3(256), 0, (hipStream_t)stream, d);
  return rumpy_check_launch("rumpy_conv4d_tail");
}

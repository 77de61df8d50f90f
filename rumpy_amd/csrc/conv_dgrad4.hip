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
        __builtin_amdgcn_sched_barrier(0);
        const int kx = j >> 2, half = (j >> 1) & 1, pass = j & 1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int r = 0; r < HR; ++r)
            acc[HR * pass + r] = mfma16<FMT>(F[ch][(ky * 3 + kx) * 2 + half], I[j % (D4_RD + 1)][r + ky], acc[HR * pass + r]);
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

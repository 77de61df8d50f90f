// Shared by conv_block.hip (one residual block per launch) and conv_block_chain.hip (a chain of blocks per launch):
// strip geometry, the XOR-swizzled unpadded LDS image, the software-pipelined MFMA sweep and the paired-tile helpers.
#pragma once
#include "common.hpp"

#ifndef BLOCK_PRIO
#define BLOCK_PRIO 1       // (round 6) a wave raises its issue priority (s_setprio) for the length of a sweep: when its sibling on the SIMD is in an epilogue or a
                           // hand-off, the sweep's MFMAs and fragment reads win the arbitration.  0: A/B.  Same box: forward chain 224.6 -> 217.6 us, data-gradient
                           // chain 234.9 -> 232.2, EDSR step +0.6 %; priority 3 = priority 1.  No arithmetic changes.
#endif
#ifndef BLOCK_PIN
#define BLOCK_PIN 1        // (round 6) block_sweep's fragment reads pinned BETWEEN the MFMAs of the group before (0: in front of them, rounds 1-5; A/B same box:
                           // EDSR step 29.60 -> 29.75 k patches/s, conv_up 53.2 -> 51.7 us, forward chain 235.2 -> 232.7 us, RCAN +0.3 %; bitwise)
#endif
constexpr int BSH = 6, BSW = 48, BCOLS = BSW + 2;
constexpr int BXROWS = BSH + 4, BTROWS = BSH + 2;
constexpr int BXBYTES = BXROWS * BCOLS * 128;       // 64000
constexpr int BTBYTES = BTROWS * BCOLS * 128;       // 51200
constexpr int BTHREADS = 512;
constexpr int BPIECES = BXROWS * BCOLS * 8;         // 4000 16-byte pieces of the input tile
constexpr int BREGS = (BPIECES + BTHREADS - 1) / BTHREADS;   // 8

// ---- geometry of a workgroup's strip (round 3: the one-launch kernels take images wider than one strip) ----
// NC = 16-column MFMA tiles per output row.  CT = false: the strip spans the image (W <= 16 NC): one zero halo column per side in both LDS
// images (the round-1/2 geometry = BlockGeo<3, false>).  CT = true: the image is cut into column tiles of OW = 16 NC output columns; the
// second conv then needs T on one REAL halo column per side, so the input image carries two halo columns per side and every wave computes
// one more (half-filled) MFMA tile in the first phase: its 4 T rows x the 2 halo columns (block_common.hpp::halo_sweep).
// SH = rows of a strip (round 4): 6 everywhere until round 3.  With 32-column tiles (NC = 2) a strip may also be 4 or 8 rows high, so that the
// number of workgroups of a launch lands on a multiple of the CUs: 16 crops of 64 x 64 are 16 x 11 x 2 = 352 workgroups of 6 rows (1.4 rounds on
// 256 CUs) and 16 x 8 x 2 = 256 of 8 rows (one round).  A row half owns OR = SH / 2 output rows and TR = OR + 1 rows of the intermediate image.
template <int NC_, bool CT_, int SH_ = BSH> struct BlockGeo {
  static constexpr int NC = NC_, OW = 16 * NC_;
  static constexpr bool CT = CT_;
  static constexpr int SH = SH_, OR = SH_ / 2, TR = SH_ / 2 + 1;
  static constexpr int XROWS = SH_ + 4, TROWS = SH_ + 2;
  static constexpr int XH = CT_ ? 2 : 1;                 // halo columns per side of the input image
  static constexpr int XC = OW + 2 * XH;                 // columns of the input image in LDS
  static constexpr int TC = OW + 2;                      // columns of the T image in LDS
  static constexpr int XBYTES = XROWS * XC * 128, TBYTES = TROWS * TC * 128;
  static constexpr int XPIECES = XROWS * XC * 8;         // 16-byte pieces of the input tile
  static constexpr int XREGS = (XPIECES + BTHREADS - 1) / BTHREADS;
  static constexpr int GPIECES = OR * OW * 8;            // a row half's strip rows as 16-byte pieces
  static constexpr int GREGS = (GPIECES + 255) / 256;
  static constexpr int SPIECES = SH_ * OW * 8;           // the whole strip
  static constexpr int SREGS = (SPIECES + BTHREADS - 1) / BTHREADS;
  static_assert(SH_ == BSH || (NC_ == 2 && CT_) || (NC_ == 4 && !CT_ && SH_ == 4), "strips of 4 or 8 rows are built for 32-column tiles (and 4 rows x 64 columns for the chain)");
};
typedef BlockGeo<3, false> GeoL;       // W <= 48: BXBYTES / BTBYTES / BCOLS above

// column tiles of a W-pixel-wide image: 48- or 32-column tiles, whichever costs less with a workgroup priced at (its 16-column units + 1):
// the halo tile, the two extra input columns per side and what a workgroup pays once (prologue, epilogue tails) are about one unit's time.
// 64 columns -> 2 x 32 (the reference's training crops), 100 -> 3 x 48, 128 -> 3 x 48, 510 -> 11 x 48.
static inline void block_col_tiles(int W, int* nc, int* ct_n) {
  if (W <= BSW) { *nc = 3; *ct_n = 1; return; }
  const int units = (W + 15) / 16;
  const int t3 = (units + 2) / 3, t2 = (units + 1) / 2;
  if (4 * t3 <= 3 * t2) { *nc = 3; *ct_n = t3; } else { *nc = 2; *ct_n = t2; }
}

struct BlockDev {
  const uint16_t* x; const uint4* w1; const float* b1; const uint4* w2; const float* b2;
  const uint16_t* mask; const uint16_t* res2; uint16_t* t; uint16_t* out;
  int N, H, W, sy_n, relu1; float scale1, scale2;
  int res_mode; const uint16_t* res1; float* pool;
  int ct_n;                  // column tiles per strip row (1 when the strip spans the image)
  unsigned char* mbits;      // ReLU mask as one byte per 8 channels: written by the forward form, read by the data-gradient form (or NULL)
};

__device__ __forceinline__ unsigned swz(int p, int chunk) { return (unsigned)(p * 128 + ((chunk ^ (p & 7)) << 4)); }

// MFMA sweep over the 18 (channel half, tap column, column tile) groups for ROWS output rows per wave (window = ROWS + 2
// input rows).  off[d][half]: per-lane byte address of (window row 0, column px, chunk 4*half + g) for XOR class d.
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
// hook(grp) runs after the MFMAs of group grp have been issued: the place for work that should travel under the matrix pipe
// (the HBM stores of the previous phase's tile, block_common.hpp::strip_store_piece)
// AHEAD: groups whose fragment reads are in flight in front of the MFMAs (1: the next group's travel under this group's MFMAs; 2 costs ROWS + 2 more
// fragment registers - measured in the chain kernel, round 5)
template <int ROWS, int FMT = RUMPY_FMT_BF16, class Hook = NoHook, int NC = 3, int COLS = BCOLS, int AHEAD = 1>
__device__ __forceinline__ void block_sweep(f32x4 (&acc)[ROWS][NC], const bf16x8 (&F)[18], const unsigned char* lds, const unsigned (&off)[8][2],
                                            Hook hook = Hook()) {
  constexpr int NG = 6 * NC;            // (channel half, tap column, column tile) groups
  if (BLOCK_PRIO) __builtin_amdgcn_s_setprio(BLOCK_PRIO);
  bf16x8 I[AHEAD + 1][ROWS + 2];
  auto load_group = [&](int grp, bf16x8 (&dst)[ROWS + 2]) {
    const int half = grp / (3 * NC), kx = (grp % (3 * NC)) / NC, c = grp % NC;
#pragma unroll
    for (int r = 0; r < ROWS + 2; ++r)
      dst[r] = *reinterpret_cast<const bf16x8*>(lds + off[(r * COLS + kx) & 7][half] + (r * COLS + 16 * c + kx) * 128);
  };
#pragma unroll
  for (int g0 = 0; g0 < AHEAD; ++g0) load_group(g0, I[g0]);
#pragma unroll
  for (int grp = 0; grp < NG; ++grp) {
    if (grp + AHEAD < NG) load_group(grp + AHEAD, I[(grp + AHEAD) % (AHEAD + 1)]);
#if !BLOCK_PIN
    __builtin_amdgcn_sched_barrier(0);   // keep the next group's reads ahead of this group's MFMAs
#endif
    const int half = grp / (3 * NC), kx = (grp % (3 * NC)) / NC, c = grp % NC;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        acc[r][c] = mfma16<FMT>(F[(ky * 3 + kx) * 2 + half], I[grp % (AHEAD + 1)][r + ky], acc[r][c]);
      }
    hook(grp);
#if BLOCK_PIN
    // round 6: the next group's ROWS + 2 fragment reads pinned BETWEEN this group's 3 ROWS MFMAs (sched_group_barrier) instead of in front of them
    if (grp + AHEAD < NG) {
      constexpr int PER = (3 * ROWS) / (ROWS + 2) > 0 ? (3 * ROWS) / (ROWS + 2) : 1;
#pragma unroll
      for (int i = 0; i < ROWS + 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // a group's instructions stay inside the group
#endif
  }
  if (BLOCK_PRIO) __builtin_amdgcn_s_setprio(0);
}

// per-lane read bases of a phase: window row 0 = row `row0` of the image (COLS columns) at byte `buffer` of the LDS allocation, window
// column 0 = image column c0 (folded in here so that every read is one register + a 16-bit immediate)
template <int COLS = BCOLS>
__device__ __forceinline__ void sweep_bases(unsigned (&off)[8][2], unsigned buffer, int row0, int px, int g, int c0 = 0) {
  const int p0 = row0 * COLS + px + c0;
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int h = 0; h < 2; ++h) off[d][h] = buffer + (unsigned)(p0 * 128 + (((4 * h + g) ^ ((p0 + d) & 7)) << 4));
}

// ---- column-tiled strips: T on the two halo columns (block_common.hpp::BlockGeo) ----
// One MFMA tile per wave whose 16 "pixels" are lane px -> (T row 4 rh + ((px >> 1) & 3), halo side px & 1); lanes px >= 8 repeat lanes
// px - 8 (their results are not used).  Same accumulation order per output element as block_sweep: (channel half, tap column) groups,
// tap row innermost - the values are bitwise those of a tile that holds the same pixels in any other arrangement.
// hoff: sweep_bases-style bases of the lane's own input window (row = its T row, column = its T column in input-image columns).
template <int FMT, int COLS>
__device__ __forceinline__ f32x4 halo_sweep(f32x4 acc, const bf16x8 (&F)[18], const unsigned char* lds, const unsigned (&hoff)[8][2]) {
  bf16x8 I[18];
#pragma unroll
  for (int t = 0; t < 18; ++t) {
    const int half = t / 9, kx = (t % 9) / 3, ky = t % 3;
    I[t] = *reinterpret_cast<const bf16x8*>(lds + hoff[(ky * COLS + kx) & 7][half] + (ky * COLS + kx) * 128);
  }
#pragma unroll
  for (int t = 0; t < 18; ++t) {
    const int half = t / 9, kx = (t % 9) / 3, ky = t % 3;
    acc = mfma16<FMT>(F[(ky * 3 + kx) * 2 + half], I[t], acc);
  }
  return acc;
}

// exchange between lane g and g ^ 1 so that even-g lanes end up with 8 consecutive channels of tile X's pixel and odd-g lanes
// with 8 channels of tile Y's pixel (conv_strip.hip)
__device__ __forceinline__ void pair_up(const f32x4& tx, const f32x4& ty, int g, float (&v)[8]) {
  // v_permlane16_swap_b32 (new in gfx950) swaps the odd 16-lane rows of its first operand with the even rows of its second:
  // even-g lanes get (own X values, the X values of lane g+1), odd-g lanes (the Y values of lane g-1, own Y values) - the whole
  // exchange in one VALU instruction per dword (the portable form, a select + ds_bpermute + two selects, was a third of the
  // epilogues' instructions).  Checked on the GPU against lane ids; `g` is implied by the lane's row.
  (void)g;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(tx[j]), __float_as_uint(ty[j]), false, false);
    v[j] = __uint_as_float(r[0]);
    v[4 + j] = __uint_as_float(r[1]);
  }
}
// ReLU-mask on PACKED bf16 pairs: keeps the halves of v whose mask half is > 0, i.e. sign bit clear and any other bit set (a NaN with
// a clear sign bit counts as > 0; the mask operand of the product path is a stored post-ReLU activation).  (t + 0x7fff) sets bit 15 of
// a half exactly when t != 0 and cannot carry into the next half.  7 integer operations per pair instead of unpack / compare /
// select / repack per element.
__device__ __forceinline__ unsigned relu_keep(unsigned m) {
  const unsigned t = m & 0x7fff7fffu;
  return ((((t + 0x7fff7fffu) & ~m) & 0x80008000u) >> 15) * 0xffffu;
}
__device__ __forceinline__ uint4 relu_mask_packed(uint4 v, uint4 m) {
  return make_uint4(v.x & relu_keep(m.x), v.y & relu_keep(m.y), v.z & relu_keep(m.z), v.w & relu_keep(m.w));
}
// The same mask as ONE BYTE per 8 channels: the forward launch stores which halves of its packed post-ReLU vector are non-zero
// (ReLU leaves +x or +0), the data-gradient launch expands bit j to the keep mask of half j.  The mask operand of a data-gradient
// launch shrinks from a full bf16 tensor (12.6 MB per launch with the halo rows) to 1/16 of it.
__device__ __forceinline__ unsigned relu_bits(uint4 o) {
  auto two = [](unsigned w) { return ((w & 0xffffu) ? 1u : 0u) | ((w >> 16) ? 2u : 0u); };
  return two(o.x) | (two(o.y) << 2) | (two(o.z) << 4) | (two(o.w) << 6);
}
__device__ __forceinline__ uint4 relu_mask_bits(uint4 v, unsigned b) {
  auto keep = [](unsigned b2) { return ((b2 & 1u) * 0xffffu) | (((b2 >> 1) & 1u) * 0xffff0000u); };
  return make_uint4(v.x & keep(b), v.y & keep(b >> 2), v.z & keep(b >> 4), v.w & keep(b >> 6));
}
template <int FMT = RUMPY_FMT_BF16>
__device__ __forceinline__ void unpack8(uint4 u, float (&m)[8]) {
  unpack4<FMT>(make_uint2(u.x, u.y), *reinterpret_cast<float(*)[4]>(&m[0]));
  unpack4<FMT>(make_uint2(u.z, u.w), *reinterpret_cast<float(*)[4]>(&m[4]));
}



// ---- HBM stores of a strip's own 6 x 48 pixels FROM THE LDS IMAGE, as whole 128-byte lines ----
// The accumulator layout of an MFMA epilogue gives a lane 8 channels of a pixel: a wave's store instruction then covers 32-byte pieces
// of 32 different lines, the lines are assembled in L2 from four waves' stores and stay dirty until the end-of-kernel write-back
// (which is what the "kernel boundary" of a dependent launch mostly waits for: B / 6 TB/s).  Stored from the LDS image instead - 8 lanes
// per pixel, 1 KB contiguous per wave-instruction - and non-temporal, the lines leave L2 while the kernel still computes:
// measured -2.1 us of 16.7 per residual-block launch (round-2 ablation builds, tests/tools/patches/abl_r03.patch).
// Piece p = tid + 512 i (i < 5) = 16-byte chunk p & 7 of strip pixel p >> 3 (row-major over 6 x 48).
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_nt(uint16_t* p, uint4 v) {
  __builtin_nontemporal_store((u32x4v){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4v*>(p));
}
constexpr int STRIP_PIECES = BSH * BSW * 8;         // 2304
constexpr int STRIP_REGS = (STRIP_PIECES + BTHREADS - 1) / BTHREADS;   // 5
// LDS pixel row of strip row 0: ROW0 = 1 in the T image, 2 in the input image (which also selects the image's column count and the
// column of strip column 0: geometry G)
template <int ROW0, class G = GeoL>
__device__ __forceinline__ void strip_stage(uint4 (&S)[G::SREGS], const unsigned char* img, int tid) {
  constexpr int COLS = ROW0 == 2 ? G::XC : G::TC, CO = ROW0 == 2 ? G::XH : 1;
#pragma unroll
  for (int i = 0; i < G::SREGS; ++i) {
    const int p = tid + BTHREADS * i, pix = (p < G::SPIECES ? p : 0) >> 3, r = pix / G::OW, col = pix - r * G::OW;
    S[i] = *reinterpret_cast<const uint4*>(img + swz((r + ROW0) * COLS + col + CO, p & 7));
  }
}
// element offset of piece i in an [N,H,W,64] tensor, or 0xffffffff when the pixel lies outside the image (x0 = image column of strip column 0)
template <class G = GeoL>
__device__ __forceinline__ unsigned strip_piece_off(int i, int tid, int n, int sy, int H, int W, int x0 = 0) {
  const int p = tid + BTHREADS * i, pix = p >> 3, r = pix / G::OW, col = pix - r * G::OW, y = sy * G::SH + r;
  return (p < G::SPIECES && y < H && x0 + col < W) ? (unsigned)(((n * H + y) * W + x0 + col) * 64 + (p & 7) * 8) : 0xffffffffu;
}


// ---- row-half groups: waves 0..3 (rh = 0) and 4..7 (rh = 1) synchronise among themselves through LDS counters ----
// The workgroup barrier between the two phases made the row half that finishes its sweep first (the older waves win the matrix pipe)
// idle through the other half's epilogue with the pipe empty (0.96 us of a 14.9 us residual-block launch, stamps of tests/tools/kbench.py
// block).  A group now counts its waves in (gate_arrive) when its rows of an LDS image are written and whoever needs those rows waits
// for the count (gate_wait): the first group starts its second sweep on the output rows that depend on its own T rows only.
__device__ __forceinline__ void gate_arrive(unsigned* cnt, int lane) {
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);     // behind this wave's LDS writes
}
__device__ __forceinline__ void gate_wait(unsigned* cnt, unsigned target) {
  while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}
// a group's 3 strip rows as 16-byte pieces: p = tg + 256 i (tg = thread within the group, i < 5, p < 1152) = chunk p & 7 of pixel p >> 3
constexpr int GROUP_PIECES = 3 * BSW * 8;            // 1152
constexpr int GROUP_REGS = (GROUP_PIECES + 255) / 256;   // 5 (the last one half used)
template <int ROW0, class G = GeoL>
__device__ __forceinline__ void group_stage(uint4 (&S)[G::GREGS], const unsigned char* img, int tg, int rh) {
  constexpr int COLS = ROW0 == 2 ? G::XC : G::TC, CO = ROW0 == 2 ? G::XH : 1;
#pragma unroll
  for (int i = 0; i < G::GREGS; ++i) {
    const int p = tg + 256 * i, pix = (p < G::GPIECES ? p : 0) >> 3, r = pix / G::OW, col = pix - r * G::OW;
    S[i] = *reinterpret_cast<const uint4*>(img + swz((G::OR * rh + r + ROW0) * COLS + col + CO, p & 7));
  }
}
template <class G = GeoL>
__device__ __forceinline__ unsigned group_piece_off(int i, int tg, int rh, int n, int sy, int H, int W, int x0 = 0) {
  const int p = tg + 256 * i, pix = p >> 3, r = pix / G::OW, col = pix - r * G::OW, y = sy * G::SH + G::OR * rh + r;
  return (p < G::GPIECES && y < H && x0 + col < W) ? (unsigned)(((n * H + y) * W + x0 + col) * 64 + (p & 7) * 8) : 0xffffffffu;
}

// Shared by conv_block_fp8.hip and conv_rcab_fp8.hip (precision 'fp8'): LDS layout of the fp8 kernels, conversions, the block-scaled MFMA sweep.
#pragma once
#include "block_common.hpp"

typedef int f8_v8i __attribute__((ext_vector_type(8)));
typedef int f8_v4i __attribute__((ext_vector_type(4)));
typedef short f8_v2s __attribute__((ext_vector_type(2)));

constexpr int F8_C16 = BSH * BSW * 128;            // 36864: a strip's own pixels as a bf16 image
constexpr int F8_X8 = BXROWS * BCOLS * 64;         // 32000
constexpr int F8_T8 = BTROWS * BCOLS * 64;         // 25600
constexpr int F8_OFF_X8 = F8_C16, F8_OFF_T8 = F8_C16 + F8_X8, F8_OFF_T16 = F8_C16 + F8_X8 + F8_T8;
constexpr int F8_LDS = F8_OFF_T16 + F8_C16;        // 131328

// MODE.FP16_OVFL (bit 23 of the wave's MODE register) = 1: conversions to fp16 / fp8 clamp an out-of-range result to the largest finite value
// instead of producing inf / NaN.  Per wave, for the lifetime of the kernel.
__device__ __forceinline__ void f8_saturating_mode() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }

__device__ __forceinline__ unsigned f8_swz(int p, int quarter) { return (unsigned)(p * 64 + ((quarter ^ (((p >> 2) & 1) << 1)) << 4)); }

// 8 fp32 -> 8 fp8 bytes of value / scale (E5M2 = false: OCP e4m3, true: e5m2); round to nearest even, saturating under f8_saturating_mode
template <bool E5M2>
__device__ __forceinline__ uint2 f8_pack8(const float (&f)[8], float scale) {
  // Left to themselves the conversion instructions round a slight overflow down to the largest finite value but turn a large one into NaN,
  // and delayed scaling means a value CAN outgrow last step's scale.  The kernels therefore run with MODE.FP16_OVFL set (f8_saturating_mode,
  // first statement of each): the conversions then saturate - measured, tests/test_fp8_gpu.py::test_fp8_conversions...  (A v_med3_f32 per
  // value in front of every conversion did the same for 0.25 us per RCAB launch.)
  f8_v2s a = {0, 0}, b = {0, 0};
  if (E5M2) {
    a = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(a, f[0], f[1], scale, false);
    a = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(a, f[2], f[3], scale, true);
    b = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(b, f[4], f[5], scale, false);
    b = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(b, f[6], f[7], scale, true);
  } else {
    a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(a, f[0], f[1], scale, false);
    a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(a, f[2], f[3], scale, true);
    b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(b, f[4], f[5], scale, false);
    b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(b, f[6], f[7], scale, true);
  }
  return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}

// the same from 8 bf16 values as they come out of HBM (a 16-byte piece of a tile): v_cvt_scalef32_pk_{fp8,bf8}_bf16 converts the packed pairs
// directly - no unpacking to fp32 (the staging pass of a tile is VALU time in front of the first MFMA)
typedef __bf16 f8_v2bf __attribute__((ext_vector_type(2)));
template <bool E5M2>
__device__ __forceinline__ uint2 f8_pack8_bf16(uint4 v, float scale) {
  f8_v2s a = {0, 0}, b = {0, 0};
  if (E5M2) {
    a = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(a, __builtin_bit_cast(f8_v2bf, v.x), scale, false);
    a = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(a, __builtin_bit_cast(f8_v2bf, v.y), scale, true);
    b = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(b, __builtin_bit_cast(f8_v2bf, v.z), scale, false);
    b = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(b, __builtin_bit_cast(f8_v2bf, v.w), scale, true);
  } else {
    a = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(a, __builtin_bit_cast(f8_v2bf, v.x), scale, false);
    a = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(a, __builtin_bit_cast(f8_v2bf, v.y), scale, true);
    b = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(b, __builtin_bit_cast(f8_v2bf, v.z), scale, false);
    b = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(b, __builtin_bit_cast(f8_v2bf, v.w), scale, true);
  }
  return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}
// running amax of packed bf16 pairs as integers: |x| = bits & 0x7fff, and non-negative floats order like their bit patterns (v_pk_max_u16);
// f8_amax_bf16_value turns the two halves into the fp32 amax
typedef unsigned short f8_v2u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned f8_amax_bf16(unsigned m, uint4 v) {
  auto mx = [](unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f8_v2u16, a), __builtin_bit_cast(f8_v2u16, b & 0x7fff7fffu)));
  };
  return mx(mx(mx(mx(m, v.x), v.y), v.z), v.w);
}
__device__ __forceinline__ float f8_amax_bf16_value(unsigned m) {
  const unsigned h = (m >> 16) > (m & 0xffffu) ? (m >> 16) : (m & 0xffffu);
  return __uint_as_float(h << 16);
}

// A = filter fragment (always e4m3), B = image fragment (e4m3 or e5m2); sa / sb = e8m0 exponents, uniform over the lanes
template <bool E5M2>
__device__ __forceinline__ f32x4 f8_mfma(f8_v8i a, f8_v8i b, f32x4 c, int sa, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, E5M2 ? 1 : 0, 0, sa, 0, sb);
}

// max over the wave, in every lane (values >= 0): the butterflies of common.hpp::wave64_sum with max
__device__ __forceinline__ float f8_wave_max(float t, int lane) {
  t = fmaxf(t, dpp_quad1(t)); t = fmaxf(t, dpp_quad2(t)); t = fmaxf(t, dpp_half_mirror(t)); t = fmaxf(t, dpp_row_mirror(t));
  t = fmaxf(t, lane_xor16(t, lane >> 4));
  t = fmaxf(t, lane_xor32(t, lane));
  return t;
}

// per-lane read bases for window row 0 = image row `row0` of the fp8 image at byte `buffer`: fb[d] = chunk g of pixel (row0, px) for XOR class d
__device__ __forceinline__ void f8_bases(unsigned (&fb)[8], unsigned buffer, int row0, int px, int g) {
  const int p0 = row0 * BCOLS + px;
#pragma unroll
  for (int d = 0; d < 8; ++d) fb[d] = buffer + (unsigned)(p0 * 64 + ((g ^ ((((p0 + d) >> 2) & 1) << 1)) << 4));
}

// The sweep.  Operands of the five MFMAs of an output tile (window rows r .. r + 2 of the image, lane = (pixel px, channel group g)):
//   P[ky] (3x): taps (ky, kx 0 | kx 1): bytes 0-15 = channels 16 g .. of the pixel at column px, bytes 16-31 = the same channels at px + 1 - one
//               row fragment R[r + ky] (two 16-byte reads), shared by the three output rows it feeds;
//   Q01       : taps (ky 0 | ky 1, kx 2): bytes 0-15 = channels 16 g .. of the pixel at (row r, px + 2), bytes 16-31 = at (row r + 1, px + 2);
//   Q2        : taps (nothing | ky 2, kx 2): bytes 0-15 meet zeros in the filter image, bytes 16-31 = channels 16 g .. at (row r + 2, px + 2).
// Round 4, second form: the column-2 pieces C[r] = (row r, px + 2) of ALL window rows sit in ONE run of registers, so that Q01's operand is the
// window C[r .. r + 1] and Q2's C[r + 1 .. r + 2] - sub-tuples of the run, no moves, no zero registers - and every 16-byte piece of the image is
// read from LDS once per column tile: 18 reads where the first form (its Q01 mixed two rows ACROSS lane groups and needed its own reads) took
// 24, and 72 fragment registers instead of 112.  With 24 reads the fp8 sweep was LDS-bound (8 waves x 72 KB per first phase = 4608 clocks of
// the CU's 128 B / clk against 3840 of matrix pipe per SIMD); with 18 it is 3456.
// hook(i), i = 0 .. 8, runs after the MFMAs of step i have been issued.
#ifndef F8_PRIO
#define F8_PRIO 1          // (round 6) 0: A/B
#endif
typedef int f8_v32i __attribute__((ext_vector_type(32)));
template <int ROWS, bool E5M2, class Hook = NoHook>
__device__ __forceinline__ void f8_sweep(f32x4 (&acc)[ROWS][3], const f8_v8i (&A)[5], const unsigned char* lds, const unsigned (&fb)[8],
                                         int sa, int sb, Hook hook = Hook()) {
  f8_v8i R[ROWS + 2];
  f8_v32i C;                              // pieces C[r] at elements 4 r .. 4 r + 3 (r < ROWS + 2 <= 6)
#pragma unroll
  for (int i = 0; i < 32; ++i) C[i] = 0;
  auto ld16 = [&](unsigned addr) { return *reinterpret_cast<const f8_v4i*>(lds + addr); };
  auto load_r = [&](int c) {
#pragma unroll
    for (int r = 0; r < ROWS + 2; ++r) {
      const int k0 = r * BCOLS + 16 * c, k1 = k0 + 1;
      const f8_v4i lo = ld16(fb[k0 & 7] + k0 * 64), hi = ld16(fb[k1 & 7] + k1 * 64);
      R[r] = (f8_v8i){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
  };
  auto load_c = [&](int c) {
#pragma unroll
    for (int r = 0; r < ROWS + 2; ++r) {
      const int k = r * BCOLS + 16 * c + 2;
      const f8_v4i v = ld16(fb[k & 7] + k * 64);
      C[4 * r] = v.x; C[4 * r + 1] = v.y; C[4 * r + 2] = v.z; C[4 * r + 3] = v.w;
    }
  };
  auto win = [&](int r) {                 // C[r .. r + 1] as one operand
    return (f8_v8i){C[4 * r], C[4 * r + 1], C[4 * r + 2], C[4 * r + 3], C[4 * r + 4], C[4 * r + 5], C[4 * r + 6], C[4 * r + 7]};
  };
  if (F8_PRIO) __builtin_amdgcn_s_setprio(F8_PRIO);      // the sweeping wave wins the SIMD's arbitration against its sibling's epilogue (block_common.hpp BLOCK_PRIO)
  load_r(0);
  load_c(0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[r][c] = f8_mfma<E5M2>(A[ky], R[r + ky], acc[r][c], sa, sb);
    hook(3 * c);
    if (c + 1 < 3) load_r(c + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r][c] = f8_mfma<E5M2>(A[3], win(r), acc[r][c], sa, sb);
    hook(3 * c + 1);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r][c] = f8_mfma<E5M2>(A[4], win(r + 1), acc[r][c], sa, sb);
    hook(3 * c + 2);
    if (c + 1 < 3) load_c(c + 1);
  }
  if (F8_PRIO) __builtin_amdgcn_s_setprio(0);
}

// a row half's 3 strip rows of a 6 x 48-pixel bf16 image as 16-byte pieces (block_common.hpp::group_stage for an image without halo)
__device__ __forceinline__ void f8_stage48(uint4 (&S)[GROUP_REGS], const unsigned char* img, int tg, int rh) {
#pragma unroll
  for (int i = 0; i < GROUP_REGS; ++i) {
    const int p = tg + 256 * i, pix = (p < GROUP_PIECES ? p : 0) >> 3;
    S[i] = *reinterpret_cast<const uint4*>(img + swz(3 * rh * BSW + pix, p & 7));
  }
}

__device__ __forceinline__ int f8_exp(unsigned word) { const int e = (int)(word & 255u); return e ? e : 127; }


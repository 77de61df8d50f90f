// Shared device helpers for the gfx950 kernels (CDNA4: wave64, MFMA 16x16x32 bf16, 160 KiB LDS/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/rumpy_amd.h"
#include "../../include/rumpy_amd_debug.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Tile geometry shared by conv / wgrad: 8 rows x 16 columns of output pixels per stage, 1-pixel halo.
constexpr int TH = RUMPY_TILE_H;       // 8
constexpr int TW = RUMPY_TILE_W;       // 16
constexpr int HALO_H = TH + 2;         // 10
constexpr int HALO_W = TW + 2;         // 18
constexpr int HALO_PIX = HALO_H * HALO_W;  // 180
// One pixel = 64 bf16 = 128 B, stored at a 160-B stride in LDS: an odd multiple of 32 B makes both the
// 16-lane ds_read_b128 fragment reads (conv) and the 8-consecutive-pixel ds_read_b64_tr_b16 reads (wgrad)
// bank-conflict free (64 banks x 4 B; checked by brute force over the gfx950 lane groups).
constexpr int PIX_STRIDE = 160;
constexpr int X_STAGE_BYTES = HALO_PIX * PIX_STRIDE;  // 28800
constexpr int DY_STAGE_BYTES = TH * TW * PIX_STRIDE;  // 20480

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) {
  union { uint4 u; bf16x8 b; } c; c.u = v; return c.b;
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) { return __uint_as_float(hi16 << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  union { __bf16 b; uint16_t u; } c; c.b = b; return c.u;
}
// two fp32 -> one dword of two bf16 (a in the low half): ONE v_cvt_pk_bf16_f32.  (Converting element by element and OR-ing the halves
// together lets the compiler pair the conversions its own way and then re-shuffle the halves: 8 more instructions per 8 values.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
  union { bf16x2v v; uint32_t u; } c;
  c.v = __builtin_convertvector((f32x2){a, b}, bf16x2v);
  return c.u;
}
__device__ __forceinline__ uint2 pack4_bf16(float a, float b, float c, float d) {
  return make_uint2(pack2_bf16(a, b), pack2_bf16(c, d));
}
// ReLU as a signed-integer max on the bit pattern: one v_max_i32 (fmaxf(x, 0) costs two instructions: IEEE mode canonicalises x first).
// Negative values, -0 and NaNs with the sign bit set become +0; everything else is unchanged.
__device__ __forceinline__ float relu_f32(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ void unpack4_bf16(uint2 v, float (&o)[4]) {
  o[0] = bf16_bits_to_f32(v.x & 0xffffu); o[1] = bf16_bits_to_f32(v.x >> 16);
  o[2] = bf16_bits_to_f32(v.y & 0xffffu); o[3] = bf16_bits_to_f32(v.y >> 16);
}

// A loaded 16-byte vector, zeroed unless `ok` - as four ANDs with an all-ones / zero mask.  Written as `if (!ok) v = make_uint4(0, 0, 0, 0)` the
// compiler is free to build an exec-masked block that holds an s_waitcnt vmcnt(0) right behind the load; in the tile loaders of the one-launch RCAB
// kernels it did (round 5, found in the ISA: eight to eighteen loads per wave, each followed by a full wait - and the block is entered by every
// strip of a W <= 48 image, whose halo columns always lie outside it): the tile then arrives one load latency at a time.
__device__ __forceinline__ uint4 keep_if(uint4 v, bool ok) {
  const unsigned m = 0u - (unsigned)ok;
  return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}

// ---- element format of the stored activations / packed filters (rumpy_amd.h: RUMPY_FMT_*) ----
// bf16 (training, the default) or IEEE fp16 (evaluation plans: same MFMA rate and bytes, 11 instead of 8 significant bits - the
// forward pass of a trained SR network stays far inside fp16's range, and rumpy_tail_fwd reports a non-finite output).  The kernels
// that an evaluation plan launches take the format as a template parameter; everything else (training) is bf16 only.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
template <int FMT> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  if (FMT == RUMPY_FMT_F16) {
    union { f16x2v v; uint32_t u; } c;
    c.v = __builtin_convertvector((f32x2){a, b}, f16x2v);      // v_cvt_pk_f16_f32 (round to nearest even; overflow -> inf)
    return c.u;
  }
  return pack2_bf16(a, b);
}
template <int FMT> __device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  return make_uint2(pack2<FMT>(a, b), pack2<FMT>(c, d));
}
template <int FMT> __device__ __forceinline__ void unpack4(uint2 v, float (&o)[4]) {
  if (FMT == RUMPY_FMT_F16) {
    union { uint2 u; _Float16 h[4]; } c; c.u = v;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (float)c.h[j];
  } else {
    unpack4_bf16(v, o);
  }
}
template <int FMT> __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if (FMT == RUMPY_FMT_F16) {
    union { bf16x8 b; f16x8 h; } ua, ub; ua.b = a; ub.b = b;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ua.h, ub.h, c, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// XCD-aware work order (cdna_hip_programming.md T1): workgroup ids are dealt round-robin over the 8 XCDs, so consecutive ids - e.g. the
// vertically adjacent strips of an image, which share 4 of their 10 input rows, or neighbouring tiles with their common halo - would sit
// behind 8 different L2s and every shared line would come from HBM twice.  Strip = (id % 8) * ceil(n / 8) + id / 8 (bijective form) gives each XCD a contiguous run of strips.
// A speed choice only: any placement is correct.
__device__ __forceinline__ int xcd_strip(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

// A pointer that a kernel reads from device memory (the x_ind / target_ind tables of rumpy_set_pointers) is a GENERIC pointer to the compiler:
// every load through it becomes a flat_load, which counts on both wait counters and is unordered, so the counted s_waitcnt vmcnt(N) of the
// surrounding code degrade to full drains (tail_fwd_kernel: +5 us per launch, round 3).  Read as a global-address-space pointer, the loads
// stay global_load.
typedef const float __attribute__((address_space(1)))* rumpy_global_cfloat;
__device__ __forceinline__ const float* load_global_ptr(const float* const* slot) {
  const unsigned long long v = *reinterpret_cast<const unsigned long long*>(slot);
  return (const float*)reinterpret_cast<rumpy_global_cfloat>(v);
}

// ---- cross-lane sums without the LDS crossbar ----
// __shfl_xor is ds_bpermute_b32: an LDS-latency operation per stage, five or six dependent ones per sum.  The same butterflies on the VALU
// (round 3): xor 1 / 2 = DPP quad permutes, xor 4 / 8 = DPP row_half_mirror / row_mirror (lane i <-> 7 - i / 15 - i: after the previous stages
// all lanes of a quad / of an 8-lane half hold the same value, so the mirrored partner carries what the xor partner carries), xor 16 / 32 =
// v_permlane16_swap / v_permlane32_swap (gfx950).  Every add has the operands of the shuffle version (in either order): same bits -
// tests/tools/overlap/dpp_check.hip compares them on the GPU.
__device__ __forceinline__ float dpp_quad1(float v) { const int i = __float_as_int(v); return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false)); }
__device__ __forceinline__ float dpp_quad2(float v) { const int i = __float_as_int(v); return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x4E, 0xF, 0xF, false)); }
__device__ __forceinline__ float dpp_half_mirror(float v) { const int i = __float_as_int(v); return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x141, 0xF, 0xF, false)); }
__device__ __forceinline__ float dpp_row_mirror(float v) { const int i = __float_as_int(v); return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x140, 0xF, 0xF, false)); }
// sum over the 16 lanes of a row (the 16 pixels of an MFMA tile), in every lane: the xor 1, 2, 4, 8 butterfly
__device__ __forceinline__ float row16_sum(float t) {
  t += dpp_quad1(t); t += dpp_quad2(t); t += dpp_half_mirror(t); t += dpp_row_mirror(t);
  return t;
}
// the value of lane ^ 16 (g = lane >> 4)
__device__ __forceinline__ float lane_xor16(float t, int g) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return __uint_as_float((g & 1) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float t, int lane) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return __uint_as_float((lane & 32) ? r[0] : r[1]);
}
// sum over the wave, in every lane (the xor 1 .. 32 butterfly)
__device__ __forceinline__ float wave64_sum(float t, int lane) {
  t = row16_sum(t);
  t += lane_xor16(t, lane >> 4);
  t += lane_xor32(t, lane);
  return t;
}

struct TileCoord { int n, ty, tx; };
__device__ __forceinline__ TileCoord decode_tile(int tile, int tiles_x, int tiles_y) {
  TileCoord t;
  t.tx = tile % tiles_x;
  const int r = tile / tiles_x;
  t.ty = r % tiles_y;
  t.n = r / tiles_y;
  return t;
}

// Issue the global loads of one 10x18-pixel x 64-channel halo tile into registers (6 x 16 B per thread).
// mode 0: src is [N,H,W,cstride] bf16, channels coff..coff+63.  mode 1: src is [N,2H,2W,64], sub-pixel q = coff
// (PixelShuffle^T gather: logical pixel (y,x) lives at (2y + q/2, 2x + q%2)).  Out-of-image pixels read as zero.
__device__ __forceinline__ void halo_issue(uint4 (&R)[6], const uint16_t* __restrict__ src, int mode, int cstride,
                                           int coff, int n, int ty, int tx, int H, int W, int tid) {
  // Branch-free: every lane loads from a clamped in-bounds address and the result is zeroed afterwards, so the six loads
  // issue back to back on every path (no exec-mask regions) and callers' s_waitcnt values stay counted.
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int p = tid + 256 * i;
    const int pix = p >> 3, part = p & 7;
    const int r = pix / HALO_W, c = pix - r * HALO_W;
    const int y = ty * TH + r - 1, x = tx * TW + c - 1;
    const bool ok = (p < HALO_PIX * 8) & ((unsigned)y < (unsigned)H) & ((unsigned)x < (unsigned)W);
    size_t e = 0;
    if (mode == 0) e = ((size_t)(n * H + y) * W + x) * cstride + coff + part * 8;
    else e = ((size_t)(n * 2 * H + 2 * y + (coff >> 1)) * (2 * W) + 2 * x + (coff & 1)) * 64 + part * 8;
    uint4 v = *reinterpret_cast<const uint4*>(src + (ok ? e : (size_t)0));
    if (!ok) v = make_uint4(0, 0, 0, 0);
    R[i] = v;
  }
}
__device__ __forceinline__ void halo_write(const uint4 (&R)[6], unsigned char* lds, int tid) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int p = tid + 256 * i;
    if (p < HALO_PIX * 8) {
      const int pix = p >> 3, part = p & 7;
      *reinterpret_cast<uint4*>(lds + pix * PIX_STRIDE + part * 16) = R[i];
    }
  }
}

// device-side argument block of the tile kernels (conv_mfma.hip, conv_dgrad4.hip)
struct ConvDev {
  const uint16_t* x; const uint4* w; const float* bias; uint16_t* out;
  const uint16_t* mask; const uint16_t* res1; const uint16_t* res2; float* pool;
  int N, H, W, cout_tiles, in_mode, out_mode, relu; float scale; int tiles_x, tiles_y;
};

// host side
void rumpy_set_error(const char* fmt, ...);
int rumpy_check_launch(const char* what);
void rumpy_probe_pre(int kernel_id, hipStream_t s);
void rumpy_probe_post(int kernel_id, hipStream_t s);
bool rumpy_probe_slot(int kernel_id, hipEvent_t* start, hipEvent_t* stop);
// launch with the timing probe's events attached to the dispatch itself when kernel id `id` is being probed (api.hip)
#define RUMPY_LAUNCH_PROBED(id, kernel, grid, block, stream, ...)                                                       \
  do {                                                                                                                  \
    hipEvent_t e0_ = nullptr, e1_ = nullptr;                                                                            \
    if (rumpy_probe_slot(id, &e0_, &e1_)) hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, e0_, e1_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                               \
  } while (0)

// BatchNorm statistics of the degradation encoder (enc_conv.hip forward, enc_train.hip backward): number of pixel blocks = partial-sum rows.
// 256 workgroups in all (blocks x C/64 channel groups) when the map is large enough: one per CU, each with 8 loads in flight per thread.
static inline int rumpy_bn_blocks(int P, int C) {
  int cap = 256 / (C / 64 > 0 ? C / 64 : 1);
  if (cap < 64) cap = 64;
  int b = (P + 255) / 256;
  return b > cap ? cap : (b < 1 ? 1 : b);
}


// fp8 go / no-go (round 3): the residual-block kernel's two sweeps on the block-scaled fp8 MFMA
// (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands, e8m0 scales), measured on the hardware next to conv_block_kernel.
// A measurement tool with a real result: forward (inference) form of conv_block.hip for W <= 48,
//     T = relu(convA(X) + b1) ;  OUT = X + scale2 * (convB(T) + b2)
// with X and T quantised to OCP e4m3 for the matrix pipe (per-tensor power-of-two scales carried by the instruction's scale operands;
// the residual operand, the accumulation and OUT stay as in the bf16 kernel).  Checked against a torch emulation of exactly this
// arithmetic (tests/tools/fp8/fp8_block.py); never loaded by the product.
//
// What differs from the bf16 kernel:
//  * LDS holds the bf16 input image (residual operand, OUT staging) plus fp8 images of X and T: 64 bytes per pixel, 16-byte chunk index
//    XOR-ed with 2 * ((pixel >> 2) & 1) (conflict-free for a wave's ds_read_b128 of chunk g of 16 consecutive pixels; brute force over the
//    gfx950 lane groups).  The conversion happens where the values are produced: X while its tile is staged (registers -> both images), T
//    in the first phase's epilogue (v_cvt_scalef32_pk_fp8_f32) - not in the sweeps.
//  * One MFMA has K = 128 = two taps x 64 channels; lane (pixel px, group g) supplies 32 bytes.  The 9 taps of an output tile are 5 MFMAs:
//      P[ky]  (3x): taps (ky, kx 0 | kx 1): lane bytes 0-15 = channels 16g.. of the pixel at kx 0, bytes 16-31 = the same channels at kx 1 -
//                   a fragment is an image ROW property, shared by the three output rows it feeds (as in the bf16 sweep);
//      Q01        : taps (ky 0 | ky 1, kx 2): lanes g < 2 read the pixel of row r, lanes g >= 2 the pixel of row r + 1, channels 32 (g & 1)..;
//      Q2         : tap (ky 2, kx 2) | nothing: bytes 0-15 = channels 16g.., bytes 16-31 meet zeros in the filter image.
//    (probe: tests/tools/fp8/fp8_probe.hip - any lane/byte -> k assignment works as long as both operands use the same one; the e8m0
//    scale of k block b is taken from lane group b, a uniform scale is what this kernel passes).
//    5 MFMAs of 32 cycles against 18 of 16: 0.56 of the matrix-pipe time.
#include "../../../rumpy_amd/csrc/block_common.hpp"

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef short v2s __attribute__((ext_vector_type(2)));

constexpr int X8BYTES = BXROWS * BCOLS * 64;   // 32000
constexpr int T8BYTES = BTROWS * BCOLS * 64;   // 25600

struct Fp8Dev {
  const uint16_t* x; const v8i* w1; const float* b1; const v8i* w2; const float* b2; uint16_t* out;
  int N, H, W, sy_n; float scale2;
  float x_scale, t_scale;      // powers of two: fp8 value = real value / scale
  int sa1, sa2, sbx, sbt;      // e8m0 scale bytes of the two filter images and of the X / T images (127 + log2 scale)
};

__device__ __forceinline__ unsigned swz8(int p, int quarter) { return (unsigned)(p * 64 + ((quarter ^ (((p >> 2) & 1) << 1)) << 4)); }

__device__ __forceinline__ uint2 to_fp8x8(const float (&f)[8], float scale) {
  v2s a = {0, 0}, b = {0, 0};
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(a, f[0], f[1], scale, false);
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(a, f[2], f[3], scale, true);
  b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(b, f[4], f[5], scale, false);
  b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(b, f[6], f[7], scale, true);
  return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}

__device__ __forceinline__ f32x4 mfma8(v8i a, v8i b, f32x4 c, int sa, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
}

// bases of the three fragment kinds for window row 0 = image row `row0` of the fp8 image at byte `buffer`:
//   fb[d]: chunk g of pixel (row0, px) for XOR class d ; hb[d]: chunk 2 (g & 1) of pixel (row0 + (g >> 1), px + 2)
__device__ __forceinline__ void bases8(unsigned (&fb)[8], unsigned (&hb)[8], unsigned buffer, int row0, int px, int g) {
  const int p0 = row0 * BCOLS + px, ph = (row0 + (g >> 1)) * BCOLS + px + 2;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    fb[d] = buffer + (unsigned)(p0 * 64 + ((g ^ ((((p0 + d) >> 2) & 1) << 1)) << 4));
    hb[d] = buffer + (unsigned)(ph * 64 + (((2 * (g & 1)) ^ ((((ph + d) >> 2) & 1) << 1)) << 4));
  }
}

template <int ROWS, class Hook = NoHook>
__device__ __forceinline__ void sweep8(f32x4 (&acc)[ROWS][3], const v8i (&A)[5], const unsigned char* lds, const unsigned (&fb)[8],
                                       const unsigned (&hb)[8], int sa, int sb, Hook hook = Hook()) {
  v8i F[ROWS + 2], Hh[ROWS], G[ROWS];
  auto ld16 = [&](unsigned addr) { return *reinterpret_cast<const v4i*>(lds + addr); };
  auto load_f = [&](int c) {
#pragma unroll
    for (int r = 0; r < ROWS + 2; ++r) {
      const int k0 = r * BCOLS + 16 * c, k1 = k0 + 1;
      const v4i lo = ld16(fb[k0 & 7] + k0 * 64), hi = ld16(fb[k1 & 7] + k1 * 64);
      F[r] = (v8i){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
  };
  auto load_h = [&](int c) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int k = r * BCOLS + 16 * c;                    // (the + 2 columns and the lane's row are in hb)
      const v4i lo = ld16(hb[k & 7] + k * 64), hi = ld16(hb[k & 7] + k * 64 + 16);
      Hh[r] = (v8i){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
  };
  auto load_g = [&](int c) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int k = (r + 2) * BCOLS + 16 * c + 2;
      const v4i lo = ld16(fb[k & 7] + k * 64);
      G[r] = (v8i){lo.x, lo.y, lo.z, lo.w, 0, 0, 0, 0};     // the filter image's second half is zero for this MFMA
    }
  };
  load_f(0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    load_h(c);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[r][c] = mfma8(A[ky], F[r + ky], acc[r][c], sa, sb);
    hook(3 * c);
    load_g(c);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r][c] = mfma8(A[3], Hh[r], acc[r][c], sa, sb);
    hook(3 * c + 1);
    if (c + 1 < 3) load_f(c + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r][c] = mfma8(A[4], G[r], acc[r][c], sa, sb);
    hook(3 * c + 2);
  }
}

__global__ void __launch_bounds__(BTHREADS, 2) conv_block_fp8_kernel(Fp8Dev a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[BXBYTES + X8BYTES + T8BYTES];
  __shared__ unsigned gate[4];
  unsigned char* const ldx = lds;
  unsigned char* const lx8 = lds + BXBYTES;
  unsigned char* const lt8 = lds + BXBYTES + X8BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = __builtin_amdgcn_readfirstlane(wave >> 2), tg = tid & 255;
  const int strip = xcd_strip(blockIdx.x, gridDim.x);
  const int n = strip / a.sy_n, sy = strip - n * a.sy_n;

  // ---- phase 0: input tile -> bf16 image (residual operand) and fp8 image (matrix operand) ----
  {
    uint4 R[BREGS];
    const int y0 = sy * BSH - 2;
#pragma unroll
    for (int i = 0; i < BREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      const int lr = pix / BCOLS, lc = pix - lr * BCOLS;
      const int y = y0 + lr, x = lc - 1;
      const bool ok = (p < BPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
      const int e = ok ? ((n * a.H + y) * a.W + x) * 64 + part * 8 : 0;
      uint4 v = *reinterpret_cast<const uint4*>(a.x + (unsigned)e);
      if (!ok) v = make_uint4(0, 0, 0, 0);
      R[i] = v;
    }
    if (tid < 4) gate[tid] = 0u;
    if (tid < BTROWS * 2 * 4) {           // border columns of the T image: convB's zero padding
      const int row = tid >> 3, side = (tid >> 2) & 1, quarter = tid & 3;
      *reinterpret_cast<uint4*>(lt8 + swz8(row * BCOLS + side * (BCOLS - 1), quarter)) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      if (p < BPIECES) {
        *reinterpret_cast<uint4*>(ldx + swz(pix, part)) = R[i];
        float f[8];
        unpack8(R[i], f);
        *reinterpret_cast<uint2*>(lx8 + swz8(pix, part >> 1) + (part & 1) * 8) = to_fp8x8(f, a.x_scale);
      }
    }
  }
  v8i A[5];
  {
    const v8i* wp = a.w1 + (size_t)q * 5 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 5; ++t) A[t] = wp[t * 64];
  }
  const int c0 = 16 * q + 4 * g;
  const int gpair = 4 * (g & ~1);
  const int chunk8 = 2 * q + (gpair >> 3);
  __syncthreads();

  // ---- phase 1: T rows 4rh .. 4rh+3 ----
  {
    f32x4 acc[4][3];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.b1) { const float4 t = *reinterpret_cast<const float4*>(a.b1 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    unsigned fb[8], hb[8];
    bases8(fb, hb, (unsigned)BXBYTES, 4 * rh, px, g);
    sweep8<4>(acc, A, lds, fb, hb, a.sa1, a.sbx);
    {
      const v8i* wp = a.w2 + (size_t)q * 5 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 5; ++t) A[t] = wp[t * 64];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const f32x4 tx = (k < 4) ? acc[k < 4 ? k : 0][0] : acc[2 * (k < 4 ? 0 : k - 4)][2];
      const f32x4 ty = (k < 4) ? acc[k < 4 ? k : 0][1] : acc[2 * (k < 4 ? 0 : k - 4) + 1][2];
      float v[8];
      pair_up(tx, ty, g, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = relu_f32(v[j]);
      const int jr = (k < 4) ? k : (2 * (k - 4) + (g & 1)), c = (k < 4) ? (g & 1) : 2;
      const int j = 4 * rh + jr, xx = 16 * c + px;
      const int y = sy * BSH - 1 + j;
      uint2 o = make_uint2(0, 0);
      if (((unsigned)y < (unsigned)a.H) & (xx < a.W)) o = to_fp8x8(v, a.t_scale);
      *reinterpret_cast<uint2*>(lt8 + swz8(j * BCOLS + xx + 1, q) + gpair) = o;      // this wave's 16 channels = quarter q; 8 of them per lane
    }
    gate_arrive(&gate[rh], lane);
  }
  unsigned soff[GROUP_REGS];
#pragma unroll
  for (int i = 0; i < GROUP_REGS; ++i) soff[i] = group_piece_off(i, tg, rh, n, sy, a.H, a.W);
  gate_wait(&gate[rh], 4u);
  if (rh == 1) gate_wait(&gate[0], 4u);

  // ---- phase 2: output rows 3rh .. 3rh+2 ; OUT = X + scale2 * (convB(T) + b2) ----
  {
    f32x4 acc[3][3];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.b2) { const float4 t = *reinterpret_cast<const float4*>(a.b2 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    unsigned fb[8], hb[8];
    const unsigned tbuf = (unsigned)(BXBYTES + X8BYTES);
    if (rh == 0) {
      bases8(fb, hb, tbuf, 0, px, g);
      sweep8<2>(*reinterpret_cast<f32x4(*)[2][3]>(&acc[0]), A, lds, fb, hb, a.sa2, a.sbt);
      gate_wait(&gate[1], 4u);
      bases8(fb, hb, tbuf, 2, px, g);
      sweep8<1>(*reinterpret_cast<f32x4(*)[1][3]>(&acc[2]), A, lds, fb, hb, a.sa2, a.sbt);
    } else {
      bases8(fb, hb, tbuf, 3, px, g);
      sweep8<3>(acc, A, lds, fb, hb, a.sa2, a.sbt);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 tx = (k < 3) ? acc[k < 3 ? k : 0][0] : acc[0][2];
      const f32x4 ty = (k < 3) ? acc[k < 3 ? k : 0][1] : acc[1][2];
      float v[8], m[8];
      pair_up(tx, ty, g, v);
      const int r = (k < 3) ? k : (g & 1), c = (k < 3) ? (g & 1) : 2;
      const int srow = 3 * rh + r, y = sy * BSH + srow, xx = 16 * c + px;
      if (y < a.H && xx < a.W) {
        unpack8(*reinterpret_cast<const uint4*>(ldx + swz((srow + 2) * BCOLS + xx + 1, chunk8)), m);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a.scale2, m[j]);
        const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<uint4*>(ldx + swz((srow + 2) * BCOLS + xx + 1, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    {
      const int srow = 3 * rh + 2, y = sy * BSH + srow, xx = 32 + px;
      if (y < a.H && xx < a.W) {
        float v[4] = {acc[2][2][0], acc[2][2][1], acc[2][2][2], acc[2][2][3]};
        float m[4];
        unpack4_bf16(*reinterpret_cast<const uint2*>(ldx + swz((srow + 2) * BCOLS + xx + 1, 2 * q + (g >> 1)) + (g & 1) * 8), m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], a.scale2, m[j]);
        *reinterpret_cast<uint2*>(ldx + swz((srow + 2) * BCOLS + xx + 1, 2 * q + (g >> 1)) + (g & 1) * 8) = pack4_bf16(v[0], v[1], v[2], v[3]);
      }
    }
  }
  gate_arrive(&gate[2 + rh], lane);
  gate_wait(&gate[2 + rh], 4u);
  {
    uint4 S[GROUP_REGS];
    group_stage<2>(S, ldx, tg, rh);
#pragma unroll
    for (int i = 0; i < GROUP_REGS; ++i)
      if (soff[i] != 0xffffffffu) st16_nt(a.out + soff[i], S[i]);
  }
}

// C ABI of the tool (ctypes: tests/tools/fp8/fp8_block.py)
struct fp8_block_args {
  const void* x; const void* w1; const float* b1; const void* w2; const float* b2; void* out;
  int32_t N, H, W; float scale2; float x_scale, t_scale; int32_t sa1, sa2;
};
extern "C" int fp8_block(const fp8_block_args* p, void* stream) {
  if (!p || !p->x || !p->w1 || !p->w2 || !p->out || p->W > BSW || p->W <= 0 || p->H <= 0 || p->N <= 0) return -1;
  Fp8Dev d;
  d.x = (const uint16_t*)p->x; d.w1 = (const v8i*)p->w1; d.b1 = p->b1; d.w2 = (const v8i*)p->w2; d.b2 = p->b2; d.out = (uint16_t*)p->out;
  d.N = p->N; d.H = p->H; d.W = p->W; d.sy_n = (p->H + BSH - 1) / BSH; d.scale2 = p->scale2;
  d.x_scale = p->x_scale; d.t_scale = p->t_scale; d.sa1 = p->sa1; d.sa2 = p->sa2;
  int ex, et;
  frexpf(p->x_scale, &ex); frexpf(p->t_scale, &et);          // scale = 2^(e - 1)
  d.sbx = 127 + ex - 1; d.sbt = 127 + et - 1;
  hipLaunchKernelGGL(conv_block_fp8_kernel, dim3(d.N * d.sy_n), dim3(BTHREADS), 0, (hipStream_t)stream, d);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

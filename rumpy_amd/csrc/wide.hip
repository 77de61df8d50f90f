// Pieces that make the SR engine independent of "64 features, PixelShuffle(2)": what EDSR at the reference's own shipped width
// (Documentation/sample_config_files/div2k/edsr.toml:43-45: 256 features x 32 blocks) and the x3 upsampler
// (rumpy/SISR/models/advanced/common.py:39-44: conv F -> 9F + PixelShuffle(3)) need beyond the kernels of the 64-feature path.
//
//  * rumpy_pixel_shuffle: nn.PixelShuffle(r) / its inverse on NHWC 16-bit maps, out[n, r h + i, r w + j, c] = in[n, h, w, c r^2 + i r + j]
//    (common.py:33,42).  The 64-feature x2 path fuses this permutation into the conv's store and into the data gradient's gather; here the
//    conv writes its natural channel order (any Cout = r^2 F), this pass permutes, and the backward pass applies the inverse to the incoming
//    gradient - after which data and weight gradient are those of a plain conv.  Pure HBM traffic: one read + one write of the map.
//  * rumpy_tail_fwd_wide / rumpy_tail_dgrad_wide: the F -> 3 tail conv (architectures.py:229) and its data gradient for F = 64 k > 64, on the
//    fp32 VALU straight from the fp32 master filter (no packed image): 2 * 27 F flops per pixel and 2 F bytes of activation - the layer
//    is bandwidth-bound at any width, and the 64-feature kernels' MFMA layout (16 rows for 3 real channels) has nothing to gain here.
#include "common.hpp"

// ---- pixel shuffle ------------------------------------------------------------------------------------------------------------------
// one thread = 8 feature channels of one LOW-resolution pixel = 8 r^2 consecutive 16-bit values there (r^2 16-byte vectors), which are the
// 8-channel vectors of the r^2 high-resolution pixels under it: a register transpose between two sets of whole 16-byte accesses.
// inverse = 0: lo -> hi ; 1: hi -> lo.
template <int R>
__global__ void __launch_bounds__(256) pixel_shuffle_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int N, int H, int W, int F,
                                                            int inverse) {
  constexpr int R2 = R * R;
  const int fv = F / 8;
  const size_t total = (size_t)N * H * W * fv;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (size_t)gridDim.x * 256) {
    const int c8 = (int)(v % fv);
    size_t t = v / fv;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const size_t lo = (((size_t)n * H + h) * W + w) * ((size_t)F * R2) + (size_t)c8 * 8 * R2;     // channels (8 c8 + k) r^2 + q, k < 8, q < r^2
    uint16_t e[8 * R2];
    if (!inverse) {
#pragma unroll
      for (int j = 0; j < R2; ++j) *reinterpret_cast<uint4*>(e + 8 * j) = *reinterpret_cast<const uint4*>(src + lo + 8 * j);
#pragma unroll
      for (int q = 0; q < R2; ++q) {
        uint16_t o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = e[k * R2 + q];
        const size_t hi = ((((size_t)n * H * R + (size_t)h * R + q / R) * W * R) + (size_t)w * R + q % R) * F + (size_t)c8 * 8;
        *reinterpret_cast<uint4*>(dst + hi) = *reinterpret_cast<const uint4*>(o);
      }
    } else {
#pragma unroll
      for (int q = 0; q < R2; ++q) {
        const size_t hi = ((((size_t)n * H * R + (size_t)h * R + q / R) * W * R) + (size_t)w * R + q % R) * F + (size_t)c8 * 8;
        uint16_t o[8];
        *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(src + hi);
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k * R2 + q] = o[k];
      }
#pragma unroll
      for (int j = 0; j < R2; ++j) *reinterpret_cast<uint4*>(dst + lo + 8 * j) = *reinterpret_cast<const uint4*>(e + 8 * j);
    }
  }
}

extern "C" int rumpy_pixel_shuffle(const rumpy_pixel_shuffle_args* p, void* stream) {
  if (!p || !p->src || !p->dst) { rumpy_set_error("rumpy_pixel_shuffle: null pointer"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->F <= 0 || p->F % 8 || p->r < 2 || p->r > 4) {
    rumpy_set_error("rumpy_pixel_shuffle: unsupported shape (N=%d H=%d W=%d F=%d r=%d)", p->N, p->H, p->W, p->F, p->r); return RUMPY_E_ARG; }
  const size_t total = (size_t)p->N * p->H * p->W * (p->F / 8);
  size_t blocks = (total + 255) / 256;
  const size_t cap = (size_t)rumpy_device_cus() * 16;
  if (blocks > cap) blocks = cap;
  const dim3 grid((unsigned)blocks), blk(256);
  hipStream_t s = (hipStream_t)stream;
  const uint16_t* src = (const uint16_t*)p->src;
  uint16_t* dst = (uint16_t*)p->dst;
  if (p->r == 2) hipLaunchKernelGGL(pixel_shuffle_kernel<2>, grid, blk, 0, s, src, dst, p->N, p->H, p->W, p->F, p->inverse);
  else if (p->r == 3) hipLaunchKernelGGL(pixel_shuffle_kernel<3>, grid, blk, 0, s, src, dst, p->N, p->H, p->W, p->F, p->inverse);
  else hipLaunchKernelGGL(pixel_shuffle_kernel<4>, grid, blk, 0, s, src, dst, p->N, p->H, p->W, p->F, p->inverse);
  return rumpy_check_launch("rumpy_pixel_shuffle");
}

// ---- tail conv F -> C (C <= 4), fp32 VALU ------------------------------------------------------------------------------------------
// workgroup = 64 consecutive output pixels of one image row x 4 channel slices (thread = pixel, slice of F/4 channels); the filter sits in
// LDS as [tap][ci][4] floats (one 16-byte broadcast read per (tap, ci): the C <= 4 output channels together); the four slices' partial sums
// meet in LDS in a fixed order.  out: fp32 NCHW, like the 64-feature tail kernel writes it.
constexpr int TW_MAXF = 512;
template <int FMT>      // element format of x (RUMPY_FMT_F16: evaluation plans)
__global__ void __launch_bounds__(256) tail_fwd_wide_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ out, int N, int H, int W, int F, int C, int* __restrict__ nonfinite) {
  extern __shared__ float4 swide[];                       // [9 * F] float4 filter, then [4][64] float4 partial sums
  float4* sw = swide;
  float4* part = swide + 9 * F;
  const int tid = threadIdx.x, px = tid & 63, sl = tid >> 6;
  for (int i = tid; i < 9 * F; i += 256) {                // i = tap * F + ci  <-  w[c][ci][tap]
    const int tap = i / F, ci = i - tap * F;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) v[c] = w[((size_t)c * F + ci) * 9 + tap];
    sw[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();
  const int segs = (W + 63) / 64;
  int t = blockIdx.x;
  const int seg = t % segs; t /= segs;
  const int y = t % H, n = t / H;
  const int xx = seg * 64 + px;
  const int cs = F / 4, c_lo = sl * cs;                   // this thread's channel slice
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (xx < W) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xc = xx + kx - 1;
        if (xc < 0 || xc >= W) continue;
        const uint4* xp = reinterpret_cast<const uint4*>(x + (((size_t)n * H + yy) * W + xc) * F + c_lo);
        const float4* wp = sw + (ky * 3 + kx) * F + c_lo;
        for (int c8 = 0; c8 < cs / 8; ++c8) {
          const uint4 v = xp[c8];
          float f[8];
          { float lo[4], hi[4]; unpack4<FMT>(make_uint2(v.x, v.y), lo); unpack4<FMT>(make_uint2(v.z, v.w), hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[i] = lo[i]; f[4 + i] = hi[i]; } }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float4 ww = wp[c8 * 8 + i];
            acc.x = fmaf(f[i], ww.x, acc.x); acc.y = fmaf(f[i], ww.y, acc.y); acc.z = fmaf(f[i], ww.z, acc.z); acc.w = fmaf(f[i], ww.w, acc.w);
          }
        }
      }
    }
  }
  part[sl * 64 + px] = acc;
  __syncthreads();
  if (sl == 0 && xx < W) {
    const float4 a = part[px], b = part[64 + px], c = part[128 + px], d = part[192 + px];
    const float r[4] = {((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z, ((a.w + b.w) + c.w) + d.w};
    bool bad = false;
    for (int ch = 0; ch < C; ++ch) {
      const float o = r[ch] + bias[ch];
      bad |= !(fabsf(o) <= 3.0e38f);
      out[(((size_t)n * C + ch) * H + y) * W + xx] = o;
    }
    if (bad && nonfinite) atomicOr(nonfinite, 1);
  }
}

// data gradient: dx[n, y, x, ci] = sum_{tap, c} dy4[n, y + 1 - ky, x + 1 - kx, c] * w[c][ci][ky][kx]; thread = one pixel x 8 channels; the
// filter in LDS as [tap][ci] float4 again.  dy4: [N, H, W, 4] bf16 (the upstream gradient as rumpy_nchw_to_nhwc4 leaves it).
__global__ void __launch_bounds__(256) tail_dgrad_wide_kernel(const uint2* __restrict__ dy4, const float* __restrict__ w, uint16_t* __restrict__ dx,
                                                              int N, int H, int W, int F, int C) {
  extern __shared__ float4 swide[];
  const int tid = threadIdx.x;
  const int fv = F / 8;
  for (int i = tid; i < 9 * F; i += 256) {              // stored as [tap][ci & 7][ci >> 3]: the lanes of a wave (consecutive 8-channel
    const int tap = i / F, ci = i - tap * F;            // groups) read consecutive 16-byte entries
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) v[c] = w[((size_t)c * F + ci) * 9 + tap];
    swide[(tap * 8 + (ci & 7)) * fv + (ci >> 3)] = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();
  const size_t total = (size_t)N * H * W * fv;
  for (size_t v = (size_t)blockIdx.x * 256 + tid; v < total; v += (size_t)gridDim.x * 256) {
    const int c8 = (int)(v % fv);
    size_t t = v / fv;
    const int xx = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + 1 - ky;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xc = xx + 1 - kx;
        if (xc < 0 || xc >= W) continue;
        float g[4];
        unpack4_bf16(dy4[((size_t)n * H + yy) * W + xc], g);
        const float4* wp = swide + (size_t)(ky * 3 + kx) * 8 * fv + c8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 ww = wp[i * fv];
          acc[i] = fmaf(g[0], ww.x, fmaf(g[1], ww.y, fmaf(g[2], ww.z, fmaf(g[3], ww.w, acc[i]))));
        }
      }
    }
    const uint2 lo = pack4_bf16(acc[0], acc[1], acc[2], acc[3]), hi = pack4_bf16(acc[4], acc[5], acc[6], acc[7]);
    *reinterpret_cast<uint4*>(dx + v * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
}

static bool tail_wide_ok(const rumpy_tail_wide_args* p, const char* who) {
  if (!p || !p->x || !p->w || !p->out) { rumpy_set_error("%s: null pointer", who); return false; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->F < 64 || p->F % 64 || p->F > TW_MAXF || p->C <= 0 || p->C > 4) {
    rumpy_set_error("%s: unsupported shape (N=%d H=%d W=%d F=%d C=%d)", who, p->N, p->H, p->W, p->F, p->C); return false; }
  return true;
}
template <typename K> static bool wide_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

extern "C" int rumpy_tail_fwd_wide(const rumpy_tail_wide_args* p, void* stream) {
  if (!tail_wide_ok(p, "rumpy_tail_fwd_wide")) return RUMPY_E_ARG;
  if (!p->bias) { rumpy_set_error("rumpy_tail_fwd_wide: null bias"); return RUMPY_E_ARG; }
  const size_t lds = ((size_t)9 * p->F + 256) * sizeof(float4);
  if (p->fmt != RUMPY_FMT_BF16 && p->fmt != RUMPY_FMT_F16) { rumpy_set_error("rumpy_tail_fwd_wide: unknown fmt %d", p->fmt); return RUMPY_E_ARG; }
  if (!wide_lds(tail_fwd_wide_kernel<RUMPY_FMT_BF16>, lds) || !wide_lds(tail_fwd_wide_kernel<RUMPY_FMT_F16>, lds)) {
    rumpy_set_error("rumpy_tail_fwd_wide: cannot reserve %zu bytes of LDS", lds); return RUMPY_E_ARG; }
  const int segs = (p->W + 63) / 64;
  const dim3 grid((unsigned)((size_t)p->N * p->H * segs));
  if (p->fmt == RUMPY_FMT_F16)
    hipLaunchKernelGGL(tail_fwd_wide_kernel<RUMPY_FMT_F16>, grid, dim3(256), lds, (hipStream_t)stream, (const uint16_t*)p->x, p->w, p->bias, (float*)p->out,
                       p->N, p->H, p->W, p->F, p->C, p->nonfinite);
  else
    hipLaunchKernelGGL(tail_fwd_wide_kernel<RUMPY_FMT_BF16>, grid, dim3(256), lds, (hipStream_t)stream, (const uint16_t*)p->x, p->w, p->bias, (float*)p->out,
                       p->N, p->H, p->W, p->F, p->C, p->nonfinite);
  return rumpy_check_launch("rumpy_tail_fwd_wide");
}

extern "C" int rumpy_tail_dgrad_wide(const rumpy_tail_wide_args* p, void* stream) {
  if (!tail_wide_ok(p, "rumpy_tail_dgrad_wide")) return RUMPY_E_ARG;
  const size_t lds = (size_t)9 * p->F * sizeof(float4);
  if (!wide_lds(tail_dgrad_wide_kernel, lds)) { rumpy_set_error("rumpy_tail_dgrad_wide: cannot reserve %zu bytes of LDS", lds); return RUMPY_E_ARG; }
  const size_t total = (size_t)p->N * p->H * p->W * (p->F / 8);
  size_t blocks = (total + 255) / 256;
  const size_t cap = (size_t)rumpy_device_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(tail_dgrad_wide_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, (const uint2*)p->x, p->w, (uint16_t*)p->out,
                     p->N, p->H, p->W, p->F, p->C);
  return rumpy_check_launch("rumpy_tail_dgrad_wide");
}

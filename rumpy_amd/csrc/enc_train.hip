// Backward pass of the degradation encoder's BatchNorm2d + LeakyReLU stages, and the momentum update of MoCo's key encoder - what training
// the `Encoder` (rumpy/regression/models/contrastive_learning/encoding_models.py:5-55) under MoCo / SupMoCo
// (moco.py:66-72,132-187, supmoco.py:52-128) adds to the inference kernels of enc_conv.hip.  The convolutions of that backward pass are the
// kernels the SR path already has: the data gradient is rumpy_enc_conv on the filter's dgrad image, the weight gradient rumpy_wgrad_grouped
// + rumpy_wgrad_reduce (rumpy_head_wgrad for the 3-channel first layer).  A stride-2 convolution needs no kernel of its own in this
// direction: the BatchNorm backward below writes its gradient at every second pixel of a zeroed stride-1 grid (`up` = 2), on which data and
// weight gradient are the stride-1 ones.
//
//   forward   y = z * sc + sh  (sc = gamma * invstd, sh = beta - mean * sc),  a = y > 0 ? y : slope * y
//   backward  dy = da * (y > 0 ? 1 : slope),  xh = (z - mean) * invstd
//             dbeta = sum dy,  dgamma = sum dy * xh,  dz = gamma * invstd * (dy - dbeta / P - xh * dgamma / P)
// Three launches per stage: per-block partial sums in a fixed order, an fp64 combine, the apply pass.  HBM-bound: z and da are read twice
// (2 x 2 x 2 B per element), dz written once.
#include "common.hpp"


struct BnBwd {
  const void* z; const uint4* da; const float* dpool; const float* scale_shift; const float* saved;
  int P, C, HW; float inv_hw, neg_slope;
};

// per-thread constants of its 8 channels: forward scale / shift, batch mean, 1 / sigma
struct BnbCh { float sc[8], sh[8], mu[8], is[8]; };
__device__ __forceinline__ BnbCh bnb_channels(const BnBwd& a, int c0) {
  BnbCh k;
#pragma unroll
  for (int i = 0; i < 8; ++i) { k.sc[i] = a.scale_shift[c0 + i]; k.sh[i] = a.scale_shift[a.C + c0 + i]; k.mu[i] = a.saved[c0 + i]; k.is[i] = a.saved[a.C + c0 + i]; }
  return k;
}
// raw operands of pixel p, channels 8 cv..: the conv output and (unless the pool's gradient stands in) the stage-output gradient
struct BnbRaw { uint4 z, z2, d; };      // (z2: the second half of an fp32 z vector)
template <int FMT>
__device__ __forceinline__ BnbRaw bnb_fetch(const BnBwd& a, int p, int cv, int cvec) {
  BnbRaw r;
  if (FMT == RUMPY_FMT_F32) { const uint4* q = reinterpret_cast<const uint4*>(a.z) + 2 * ((size_t)p * cvec + cv); r.z = q[0]; r.z2 = q[1]; }
  else { r.z = reinterpret_cast<const uint4*>(a.z)[(size_t)p * cvec + cv]; r.z2 = make_uint4(0, 0, 0, 0); }
  r.d = a.da ? a.da[(size_t)p * cvec + cv] : make_uint4(0, 0, 0, 0);
  return r;
}
// -> dy and xh of those 8 channels (FMT: format of the stored conv output z; the gradient da is bf16)
template <int FMT>
__device__ __forceinline__ void bnb_decode(const BnBwd& a, const BnbCh& k, const BnbRaw& r, int p, int c0, float (&dy)[8], float (&xh)[8]) {
  float z[8], d[8];
  if (FMT == RUMPY_FMT_F32) {
    z[0] = __uint_as_float(r.z.x); z[1] = __uint_as_float(r.z.y); z[2] = __uint_as_float(r.z.z); z[3] = __uint_as_float(r.z.w);
    z[4] = __uint_as_float(r.z2.x); z[5] = __uint_as_float(r.z2.y); z[6] = __uint_as_float(r.z2.z); z[7] = __uint_as_float(r.z2.w);
  } else {
    constexpr int ZF = FMT == RUMPY_FMT_F32 ? RUMPY_FMT_BF16 : FMT;
    float lo[4], hi[4]; unpack4<ZF>(make_uint2(r.z.x, r.z.y), lo); unpack4<ZF>(make_uint2(r.z.z, r.z.w), hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) { z[i] = lo[i]; z[4 + i] = hi[i]; }
  }
  if (a.da) {
    float lo[4], hi[4]; unpack4_bf16(make_uint2(r.d.x, r.d.y), lo); unpack4_bf16(make_uint2(r.d.z, r.d.w), hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) { d[i] = lo[i]; d[4 + i] = hi[i]; }
  } else {                                                  // AdaptiveAvgPool2d(1) backward: every pixel of image n gets dpool[n] / HW
    const float* dp = a.dpool + (size_t)(p / a.HW) * a.C + c0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = dp[i] * a.inv_hw;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float y = fmaf(z[i], k.sc[i], k.sh[i]);           // the forward pass's own arithmetic: same sign
    dy[i] = y > 0.f ? d[i] : d[i] * a.neg_slope;
    xh[i] = (z[i] - k.mu[i]) * k.is[i];
  }
}

template <int FMT>
__global__ void __launch_bounds__(256) enc_bnb_stats_kernel(BnBwd a, float* __restrict__ partial, int chunk) {
  __shared__ float red[32][129];
  const int b = blockIdx.x, cg = blockIdx.y, tid = threadIdx.x, c8 = tid & 7, pr = tid >> 3;
  const int cvec = a.C / 8, cv = cg * 8 + c8, c0 = cv * 8;
  const int p0 = b * chunk, p1 = min(a.P, p0 + chunk);
  const BnbCh k = bnb_channels(a, c0);
  float s[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
  auto add = [&](const BnbRaw& r, int p) {
    float dy[8], xh[8];
    bnb_decode<FMT>(a, k, r, p, c0, dy, xh);
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] += dy[i]; q[i] = fmaf(dy[i], xh[i], q[i]); }
  };
  int p = p0 + pr;
  for (; p + 3 * 32 < p1; p += 4 * 32) {               // four pixels (eight loads) in flight per thread; summed in pixel order all the same
    BnbRaw r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = bnb_fetch<FMT>(a, p + 32 * j, cv, cvec);
#pragma unroll
    for (int j = 0; j < 4; ++j) add(r[j], p + 32 * j);
  }
  for (; p < p1; p += 32) add(bnb_fetch<FMT>(a, p, cv, cvec), p);
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[pr][c8 * 8 + i] = s[i]; red[pr][64 + c8 * 8 + i] = q[i]; }
  __syncthreads();
  if (tid < 128) {
    float tot = 0.f;
    for (int j = 0; j < 32; ++j) tot += red[j][tid];
    const int which = tid >> 6, c = cg * 64 + (tid & 63);
    partial[((size_t)b * 2 + which) * a.C + c] = tot;
  }
}

// -> dgamma, dbeta (x scale) and the three per-channel coefficients of the apply pass, coef[3][C]: gamma * invstd, dbeta / P, dgamma / P
__global__ void __launch_bounds__(256) enc_bnb_finalize_kernel(const float* __restrict__ partial, int nblk, int P, int C, const float* __restrict__ gamma,
                                                               const float* __restrict__ saved, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ coef, float scale) {
  __shared__ double red[4][64][2];
  const int part = threadIdx.x >> 6, c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s = 0.0, q = 0.0;                           // four threads per channel, blocks part, part + 4, ..: fixed order
  if (c < C) {
#pragma unroll 8
    for (int b = part; b < nblk; b += 4) { s += (double)partial[((size_t)b * 2) * C + c]; q += (double)partial[((size_t)b * 2 + 1) * C + c]; }
  }
  red[part][threadIdx.x & 63][0] = s; red[part][threadIdx.x & 63][1] = q;
  __syncthreads();
  if (part != 0 || c >= C) return;
  s = ((red[0][threadIdx.x][0] + red[1][threadIdx.x][0]) + red[2][threadIdx.x][0]) + red[3][threadIdx.x][0];
  q = ((red[0][threadIdx.x][1] + red[1][threadIdx.x][1]) + red[2][threadIdx.x][1]) + red[3][threadIdx.x][1];
  if (dbeta) dbeta[c] = (float)s * scale;
  if (dgamma) dgamma[c] = (float)q * scale;
  coef[c] = gamma[c] * saved[C + c];
  coef[C + c] = (float)(s / P);
  coef[2 * C + c] = (float)(q / P);
}

template <int FMT>
__global__ void __launch_bounds__(256) enc_bnb_apply_kernel(BnBwd a, const float* __restrict__ coef, uint4* __restrict__ dz, int Wo, int up, int Hz, int Wz) {
  const int cvec = a.C / 8;
  const size_t total = (size_t)a.P * cvec;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (size_t)gridDim.x * 256) {
    const int cv = (int)(v % cvec), p = (int)(v / cvec), c0 = cv * 8;
    float dy[8], xh[8], r[8];
    bnb_decode<FMT>(a, bnb_channels(a, c0), bnb_fetch<FMT>(a, p, cv, cvec), p, c0, dy, xh);
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = coef[c0 + i] * (dy[i] - coef[a.C + c0 + i] - xh[i] * coef[2 * a.C + c0 + i]);
    const uint2 lo = pack4_bf16(r[0], r[1], r[2], r[3]), hi = pack4_bf16(r[4], r[5], r[6], r[7]);
    size_t q = (size_t)p;
    if (up != 1) {                                         // pixel (n, oy, ox) of the conv output -> (n, up * oy, up * ox) of the stride-1 grid
      const int ox = p % Wo, t = p / Wo, oy = t % (a.HW / Wo), n = t / (a.HW / Wo);
      q = ((size_t)n * Hz + (size_t)up * oy) * Wz + (size_t)up * ox;
    }
    dz[q * cvec + cv] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
}

extern "C" int rumpy_enc_bn_bwd(const rumpy_enc_bn_bwd_args* p, void* stream) {
  if (!p || !p->z || (!p->da && !p->dpool) || !p->scale_shift || !p->saved || !p->gamma || !p->dz || !p->partial || !p->coef) {
    rumpy_set_error("rumpy_enc_bn_bwd: null pointer"); return RUMPY_E_ARG; }
  const long long P = (long long)p->N * p->Ho * p->Wo;
  if (p->fmt != RUMPY_FMT_BF16 && p->fmt != RUMPY_FMT_F16 && p->fmt != RUMPY_FMT_F32) { rumpy_set_error("rumpy_enc_bn_bwd: fmt %d", p->fmt); return RUMPY_E_ARG; }
  const bool h16 = p->fmt == RUMPY_FMT_F16, z32 = p->fmt == RUMPY_FMT_F32;
  if (p->N <= 0 || p->Ho <= 0 || p->Wo <= 0 || P < 2 || P > 0x7fffffffLL || p->C <= 0 || p->C % 64 || (p->up != 1 && p->up != 2) ||
      p->Hz < p->up * (p->Ho - 1) + 1 || p->Wz < p->up * (p->Wo - 1) + 1) {
    rumpy_set_error("rumpy_enc_bn_bwd: unsupported shape (N=%d Ho=%d Wo=%d C=%d up=%d Hz=%d Wz=%d)", p->N, p->Ho, p->Wo, p->C, p->up, p->Hz, p->Wz);
    return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  BnBwd a;
  a.z = p->z; a.da = (const uint4*)p->da; a.dpool = p->dpool; a.scale_shift = p->scale_shift; a.saved = p->saved;
  a.P = (int)P; a.C = p->C; a.HW = p->Ho * p->Wo; a.inv_hw = 1.f / (float)a.HW; a.neg_slope = p->neg_slope;
  const int nblk = rumpy_bn_blocks(a.P, a.C), chunk = (a.P + nblk - 1) / nblk;
  if (z32) hipLaunchKernelGGL(enc_bnb_stats_kernel<RUMPY_FMT_F32>, dim3(nblk, a.C / 64), dim3(256), 0, s, a, p->partial, chunk);
  else if (h16) hipLaunchKernelGGL(enc_bnb_stats_kernel<RUMPY_FMT_F16>, dim3(nblk, a.C / 64), dim3(256), 0, s, a, p->partial, chunk);
  else hipLaunchKernelGGL(enc_bnb_stats_kernel<RUMPY_FMT_BF16>, dim3(nblk, a.C / 64), dim3(256), 0, s, a, p->partial, chunk);
  hipLaunchKernelGGL(enc_bnb_finalize_kernel, dim3(a.C / 64), dim3(256), 0, s, p->partial, nblk, a.P, a.C, p->gamma, p->saved, p->dgamma, p->dbeta,
                     p->coef, p->scale);
  const size_t tv = (size_t)a.P * (a.C / 8);
  size_t blocks = (tv + 255) / 256;
  const size_t cap = (size_t)rumpy_device_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (z32) hipLaunchKernelGGL(enc_bnb_apply_kernel<RUMPY_FMT_F32>, dim3((unsigned)blocks), dim3(256), 0, s, a, p->coef, (uint4*)p->dz, p->Wo, p->up, p->Hz, p->Wz);
  else if (h16) hipLaunchKernelGGL(enc_bnb_apply_kernel<RUMPY_FMT_F16>, dim3((unsigned)blocks), dim3(256), 0, s, a, p->coef, (uint4*)p->dz, p->Wo, p->up, p->Hz, p->Wz);
  else hipLaunchKernelGGL(enc_bnb_apply_kernel<RUMPY_FMT_BF16>, dim3((unsigned)blocks), dim3(256), 0, s, a, p->coef, (uint4*)p->dz, p->Wo, p->up, p->Hz, p->Wz);
  return rumpy_check_launch("rumpy_enc_bn_bwd");
}

// MoCo's momentum update of the key encoder (moco.py:66-72): k = k * m + q * (1 - m), each product rounded on its own like the torch
// expression, over the flat parameter buffers of the two encoders.  The caller passes 1 - m as the reference forms it (in double, then
// rounded to fp32 with the multiplication): fp32(1 - 0.999) is not 1.f - 0.999f.
__global__ void __launch_bounds__(256) ema_kernel(float* __restrict__ k, const float* __restrict__ q, size_t n, float m, float one_minus_m) {
#pragma clang fp contract(off)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float a = k[i] * m, b = q[i] * one_minus_m;
    k[i] = a + b;
  }
}
extern "C" int rumpy_ema(float* k, const float* q, int64_t n, float m, float one_minus_m, void* stream) {
  if (!k || !q || n <= 0) { rumpy_set_error("rumpy_ema: bad argument"); return RUMPY_E_ARG; }
  size_t blocks = ((size_t)n + 255) / 256;
  const size_t cap = (size_t)rumpy_device_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k, q, (size_t)n, m, one_minus_m);
  return rumpy_check_launch("rumpy_ema");
}

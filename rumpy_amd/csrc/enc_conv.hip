// Degradation-encoder convolutions (frozen, inference only): the DASR-style `Encoder` the blind-SR pipeline runs in front of QRCAN
// (rumpy/regression/models/contrastive_learning/encoding_models.py:5-55; ContrastiveBlindSRPipeline.forward,
// rumpy/SISR/models/blur_kernel_blind_sr/contrastive_blind_sr.py:241-329).  Five of its six 3x3 convs have 64-multiple channel counts
// (64->64, 64->128 /2, 128->128, 128->256 /2, 256->256), each followed by an eval-mode BatchNorm (folded into filter and bias on the
// host, once - the encoder is frozen) and LeakyReLU(0.1); the first conv (3->64) goes through rumpy_head_fwd.  0.3 % of a blind-QRCAN
// step's FLOPs, so the kernel is the plain form of the implicit GEMM the SR body uses:
//   D[co 16][px 16] += A[co][k = 32 input channels of one tap] * B[k][px]          (v_mfma_f32_16x16x32_bf16)
// workgroup = R = 4 output rows x 16 output columns x 64 output channels; wave w owns channels 16w.. of all R rows; per 64-channel
// input chunk the halo tile is staged in LDS (unpadded 128-B pixels, 16-B chunk index XOR (pixel & 7)) and the wave's 18 filter
// fragments are read straight from the packed image (kind-0 layout of rumpy_pack_weights, 1 KB coalesced per fragment).
// (R = 8 - half the filter-fragment traffic per pixel - was measured SLOWER: 125 vs 95 us for 64 -> 64 at 256 x 48 x 48, 256 VGPRs.  The SR
// path's strip kernel does that shape in 43 us but takes cin = 64 or 256 only; tests/tools/enc_conv_ab.py.)
#include "common.hpp"

struct EncConv {
  const uint4* x; const uint4* w; const float* bias; uint16_t* out;
  const uint4* w_lo;        // NULL, or the filter's rounding-residual image: swept into the same accumulators (LO builds)
  float* out32;             // O32 builds: the accumulators go out as fp32
  int N, H, W, Ho, Wo, chn, ctn, tiles_x, tiles_y;
  float neg_slope;
};

// LO / O32 (round 4, the encoder's TRAINING forward pass): second filter image = w - fp16(w) and fp32 output.  The gradient of this BatchNorm +
// LeakyReLU network is dominated by pre-activations that change sign under forward rounding; of the three 11-bit roundings of the fp16 pass
// (filter, conv output z, stage output a) the filter's and z's together carry most of it: with both removed the worst parameter tensor is 3-4e-2
// from the fp32 reference on five seeded cases where the all-fp16 pass gives 5-8e-2 (CPU simulation of the rounding points, tests/tools/enc_storage_sim.py).
// CW = 16-column tiles per workgroup.  CW = 2 for the LO builds (a filter fragment then feeds twice the MFMAs) was measured SLOWER - 26.3 against
// 20.3 us for 64 -> 64 at 64 x 48 x 48: 256 registers, one wave per SIMD less - and is not instantiated.
template <int S, int R, int FMT = RUMPY_FMT_BF16, bool LO = false, bool O32 = false, int CW = 1>
__global__ void __launch_bounds__(256) enc_conv_kernel(EncConv a) {
  constexpr int IR = (R - 1) * S + 3, IC = (16 * CW - 1) * S + 3;    // input rows / columns under an R x 16 CW output tile
  __shared__ uint4 lds[IR * IC * 8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int ct = blockIdx.y;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int n = t / a.tiles_y;
  const int oy0 = ty * R, ox0 = tx * 16 * CW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  const int cin8 = a.chn * 8;                              // 16-byte vectors per input pixel
  f32x4 acc[R][CW];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ch = 0; ch < a.chn; ++ch) {
    __syncthreads();
    for (int v = tid; v < IR * IC * 8; v += 256) {
      const int p = v >> 3, c8 = v & 7;
      const int r = p / IC, c = p - r * IC;
      const int y = iy0 + r, xx = ix0 + c;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (y >= 0 && y < a.H && xx >= 0 && xx < a.W) val = a.x[((size_t)(n * a.H + y) * a.W + xx) * cin8 + ch * 8 + c8];
      lds[p * 8 + (c8 ^ (p & 7))] = val;
    }
    const uint4* wp = a.w + ((size_t)((ct * a.chn + ch) * 4 + wave) * 18) * 64 + lane;
    uint4 F[18], Fl[LO ? 18 : 1];
#pragma unroll
    for (int s = 0; s < 18; ++s) F[s] = wp[s * 64];
    if (LO) {
      const uint4* wl = a.w_lo + ((size_t)((ct * a.chn + ch) * 4 + wave) * 18) * 64 + lane;
#pragma unroll
      for (int s = 0; s < 18; ++s) Fl[LO ? s : 0] = wl[s * 64];
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int chunk = half * 4 + g;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int c = 0; c < CW; ++c) {
            const int p = (r * S + ky) * IC + (j + 16 * c) * S + kx;
            const uint4 b = lds[p * 8 + (chunk ^ (p & 7))];
            if (LO) acc[r][c] = mfma16<FMT>(as_bf16x8(Fl[LO ? tap * 2 + half : 0]), as_bf16x8(b), acc[r][c]);      // the small terms first
            acc[r][c] = mfma16<FMT>(as_bf16x8(F[tap * 2 + half]), as_bf16x8(b), acc[r][c]);
          }
      }
    }
  }
  const int co = ct * 64 + wave * 16 + 4 * g;              // D[m = 4g + i][n = j]
  const float4 bv = *reinterpret_cast<const float4*>(a.bias + co);
  const int cout = a.ctn * 64;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int oy = oy0 + r, ox = ox0 + 16 * c + j;
      if (oy < a.Ho && ox < a.Wo) {
        float v0 = acc[r][c][0] + bv.x, v1 = acc[r][c][1] + bv.y, v2 = acc[r][c][2] + bv.z, v3 = acc[r][c][3] + bv.w;
        v0 = v0 > 0.f ? v0 : v0 * a.neg_slope; v1 = v1 > 0.f ? v1 : v1 * a.neg_slope;
        v2 = v2 > 0.f ? v2 : v2 * a.neg_slope; v3 = v3 > 0.f ? v3 : v3 * a.neg_slope;
        if (O32) *reinterpret_cast<float4*>(a.out32 + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * cout + co) = make_float4(v0, v1, v2, v3);
        else *reinterpret_cast<uint2*>(a.out + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * cout + co) = pack4<FMT>(v0, v1, v2, v3);
      }
    }
}

// 8 consecutive stored elements (vector index i of a [.., C] tensor in 8-element vectors) as raw registers, FMT = RUMPY_FMT_F32: two 16-byte loads
struct EncRaw8 { uint4 a, b; };
template <int FMT> __device__ __forceinline__ EncRaw8 enc_fetch8(const void* base, size_t i) {
  EncRaw8 r;
  if (FMT == RUMPY_FMT_F32) { const uint4* p = reinterpret_cast<const uint4*>(base) + 2 * i; r.a = p[0]; r.b = p[1]; }
  else { r.a = reinterpret_cast<const uint4*>(base)[i]; r.b = make_uint4(0, 0, 0, 0); }
  return r;
}
template <int FMT> __device__ __forceinline__ void enc_decode8(const EncRaw8& r, float (&f)[8]);
// 8 stored elements -> fp32
template <int FMT> __device__ __forceinline__ void enc_unpack8(const uint4 v, float (&f)[8]) {
  float lo[4], hi[4];
  unpack4<FMT>(make_uint2(v.x, v.y), lo);
  unpack4<FMT>(make_uint2(v.z, v.w), hi);
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[i] = lo[i]; f[4 + i] = hi[i]; }
}

template <int FMT> __device__ __forceinline__ void enc_decode8(const EncRaw8& r, float (&f)[8]) {
  if (FMT == RUMPY_FMT_F32) {
    f[0] = __uint_as_float(r.a.x); f[1] = __uint_as_float(r.a.y); f[2] = __uint_as_float(r.a.z); f[3] = __uint_as_float(r.a.w);
    f[4] = __uint_as_float(r.b.x); f[5] = __uint_as_float(r.b.y); f[6] = __uint_as_float(r.b.z); f[7] = __uint_as_float(r.b.w);
  } else {
    enc_unpack8<FMT == RUMPY_FMT_F32 ? RUMPY_FMT_BF16 : FMT>(r.a, f);
  }
}

// AdaptiveAvgPool2d(1) over an NHWC bf16 / fp16 map -> fp32 [N, C]; grid (N, C/64), fixed summation order (deterministic).
template <int FMT>
__global__ void __launch_bounds__(256) enc_pool_kernel(const uint4* __restrict__ x, float* __restrict__ out, int HW, int C) {
  __shared__ float red[32][65];
  const int n = blockIdx.x, cg = blockIdx.y, tid = threadIdx.x, c8 = tid & 7, pr = tid >> 3;
  const int cvec = C / 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int p = pr; p < HW; p += 32) {
    float f[8];
    enc_unpack8<FMT>(x[((size_t)n * HW + p) * cvec + cg * 8 + c8], f);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] += f[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[pr][c8 * 8 + i] = s[i];
  __syncthreads();
  if (tid < 64) {
    float tot = 0.f;
    for (int k = 0; k < 32; ++k) tot += red[k][tid];
    out[(size_t)n * C + cg * 64 + tid] = tot / (float)HW;
  }
}

// ---- training-mode BatchNorm2d of the (frozen) encoder.  The reference runs the encoder under net.train() inside run_train
// (base_architecture.py:472 after handlers.py:517), so its BatchNorms normalise with BATCH statistics and update their running
// statistics even though no parameter of the encoder is trained.  Three launches on the conv's bf16 output [P = N*H*W, C]:
// per-block partial sums (fixed order), finalize (fp64 combine -> scale/shift, running statistics, counter), apply + LeakyReLU in place.

template <int FMT>
__global__ void __launch_bounds__(256) enc_bn_stats_kernel(const void* __restrict__ x, float* __restrict__ partial, int P, int C, int chunk) {
  __shared__ float red[32][129];
  const int b = blockIdx.x, cg = blockIdx.y, tid = threadIdx.x, c8 = tid & 7, pr = tid >> 3;
  const int cvec = C / 8;
  const int p0 = b * chunk, p1 = min(P, p0 + chunk);
  float s[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
  auto add = [&](const EncRaw8& v) {
    float f[8];
    enc_decode8<FMT>(v, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] += f[i]; q[i] = fmaf(f[i], f[i], q[i]); }
  };
  int p = p0 + pr;
  for (; p + 7 * 32 < p1; p += 8 * 32) {           // eight loads in flight per thread; summed in pixel order all the same
    EncRaw8 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = enc_fetch8<FMT>(x, (size_t)(p + 32 * k) * cvec + cg * 8 + c8);
#pragma unroll
    for (int k = 0; k < 8; ++k) add(v[k]);
  }
  for (; p < p1; p += 32) add(enc_fetch8<FMT>(x, (size_t)p * cvec + cg * 8 + c8));
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[pr][c8 * 8 + i] = s[i]; red[pr][64 + c8 * 8 + i] = q[i]; }
  __syncthreads();
  if (tid < 128) {
    float tot = 0.f;
    for (int k = 0; k < 32; ++k) tot += red[k][tid];
    const int which = tid >> 6, c = cg * 64 + (tid & 63);
    partial[((size_t)b * 2 + which) * C + c] = tot;
  }
}

__global__ void __launch_bounds__(256) enc_bn_finalize_kernel(const float* __restrict__ partial, int nblk, int P, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt,
                                                              float* __restrict__ scale_shift, float* __restrict__ saved, float eps, float momentum) {
  __shared__ double red[4][64][2];
  const int part = threadIdx.x >> 6, c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s = 0.0, q = 0.0;                           // four threads per channel, blocks part, part + 4, ..: fixed order
  if (c < C) {
#pragma unroll 8
    for (int b = part; b < nblk; b += 4) { s += (double)partial[((size_t)b * 2) * C + c]; q += (double)partial[((size_t)b * 2 + 1) * C + c]; }
  }
  red[part][threadIdx.x & 63][0] = s; red[part][threadIdx.x & 63][1] = q;
  __syncthreads();
  if (part == 0 && c < C) {
    s = ((red[0][threadIdx.x][0] + red[1][threadIdx.x][0]) + red[2][threadIdx.x][0]) + red[3][threadIdx.x][0];
    q = ((red[0][threadIdx.x][1] + red[1][threadIdx.x][1]) + red[2][threadIdx.x][1]) + red[3][threadIdx.x][1];
    const double mean = s / P;
    double var = q / P - mean * mean;                      // biased variance: what the batch is normalised with
    if (var < 0.0) var = 0.0;
    const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
    scale_shift[c] = sc;
    scale_shift[C + c] = beta[c] - (float)mean * sc;
    if (saved) { saved[c] = (float)mean; saved[C + c] = (float)(1.0 / sqrt(var + (double)eps)); }   // for rumpy_enc_bn_bwd
    if (rmean) {                                           // running statistics: unbiased variance (torch.nn.BatchNorm2d)
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * ((double)P / (double)(P - 1)));
    }
  }
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

template <int FMT, int XF = FMT>
__global__ void __launch_bounds__(256) enc_bn_apply_kernel(const void* x, uint4* out, uint4* out_bf16, const float* __restrict__ scale_shift, size_t total_vec,
                                                           int C, float neg_slope) {
  const int cvec = C / 8;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < total_vec; v += (size_t)gridDim.x * 256) {
    const int c0 = (int)(v % cvec) * 8;
    float f[8];
    enc_decode8<XF>(enc_fetch8<XF>(x, v), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float y = fmaf(f[i], scale_shift[c0 + i], scale_shift[C + c0 + i]);
      f[i] = y > 0.f ? y : y * neg_slope;
    }
    const uint2 lo = pack4<FMT>(f[0], f[1], f[2], f[3]), hi = pack4<FMT>(f[4], f[5], f[6], f[7]);
    out[v] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    if (out_bf16) {                                        // the weight-gradient kernels' operand: the same values as bf16
      const uint2 l2 = pack4_bf16(f[0], f[1], f[2], f[3]), h2 = pack4_bf16(f[4], f[5], f[6], f[7]);
      out_bf16[v] = make_uint4(l2.x, l2.y, h2.x, h2.y);
    }
  }
}

extern "C" int64_t rumpy_enc_bn_partial_floats(int32_t P, int32_t C) { return (int64_t)rumpy_bn_blocks(P, C) * 2 * C; }
static int enc_bn_train(const rumpy_enc_bn_args* p, void* out, void* out_bf16, float* saved, void* stream, const char* who) {
  if (!p || !p->x || !p->gamma || !p->beta || !p->partial || !p->scale_shift) { rumpy_set_error("%s: null pointer", who); return RUMPY_E_ARG; }
  if ((p->running_mean == nullptr) != (p->running_var == nullptr)) { rumpy_set_error("%s: running_mean and running_var go together", who); return RUMPY_E_ARG; }
  if (p->fmt != RUMPY_FMT_BF16 && p->fmt != RUMPY_FMT_F16) { rumpy_set_error("%s: fmt %d", who, p->fmt); return RUMPY_E_ARG; }
  const bool h16 = p->fmt == RUMPY_FMT_F16;
  const bool x32 = p->x_fmt == RUMPY_FMT_F32;
  if (p->x_fmt != 0 && !x32) { rumpy_set_error("%s: x_fmt %d", who, p->x_fmt); return RUMPY_E_ARG; }
  if (x32 && (out == (void*)p->x || !h16)) { rumpy_set_error("%s: an fp32 x goes with the out-of-place form and fp16 outputs", who); return RUMPY_E_ARG; }
  if (p->P < 2 || p->C <= 0 || p->C % 64) {   // torch refuses a single value per channel in training mode, too
    rumpy_set_error("%s: needs more than one value per channel and C %% 64 == 0 (P=%d C=%d)", who, p->P, p->C); return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const int nblk = rumpy_bn_blocks(p->P, p->C), chunk = (p->P + nblk - 1) / nblk;
  if (x32) hipLaunchKernelGGL(enc_bn_stats_kernel<RUMPY_FMT_F32>, dim3(nblk, p->C / 64), dim3(256), 0, s, (const void*)p->x, p->partial, p->P, p->C, chunk);
  else if (h16) hipLaunchKernelGGL(enc_bn_stats_kernel<RUMPY_FMT_F16>, dim3(nblk, p->C / 64), dim3(256), 0, s, (const void*)p->x, p->partial, p->P, p->C, chunk);
  else hipLaunchKernelGGL(enc_bn_stats_kernel<RUMPY_FMT_BF16>, dim3(nblk, p->C / 64), dim3(256), 0, s, (const void*)p->x, p->partial, p->P, p->C, chunk);
  hipLaunchKernelGGL(enc_bn_finalize_kernel, dim3(p->C / 64), dim3(256), 0, s, p->partial, nblk, p->P, p->C, p->gamma, p->beta, p->running_mean,
                     p->running_var, (long long*)p->num_batches_tracked, p->scale_shift, saved, p->eps, p->momentum);
  const size_t tv = (size_t)p->P * (p->C / 8);
  size_t blocks = (tv + 255) / 256;
  const size_t cap = (size_t)rumpy_device_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (x32) hipLaunchKernelGGL((enc_bn_apply_kernel<RUMPY_FMT_F16, RUMPY_FMT_F32>), dim3((unsigned)blocks), dim3(256), 0, s, (const void*)p->x, (uint4*)out, (uint4*)out_bf16, p->scale_shift, tv, p->C, p->neg_slope);
  else if (h16) hipLaunchKernelGGL(enc_bn_apply_kernel<RUMPY_FMT_F16>, dim3((unsigned)blocks), dim3(256), 0, s, (const void*)p->x, (uint4*)out, (uint4*)out_bf16, p->scale_shift, tv, p->C, p->neg_slope);
  else hipLaunchKernelGGL(enc_bn_apply_kernel<RUMPY_FMT_BF16>, dim3((unsigned)blocks), dim3(256), 0, s, (const void*)p->x, (uint4*)out, (uint4*)out_bf16, p->scale_shift, tv, p->C, p->neg_slope);
  return rumpy_check_launch(who);
}
extern "C" int rumpy_enc_bn_train(const rumpy_enc_bn_args* p, void* stream) { return enc_bn_train(p, p ? p->x : nullptr, nullptr, nullptr, stream, "rumpy_enc_bn_train"); }
extern "C" int rumpy_enc_bn_train_keep(const rumpy_enc_bn_args* p, void* out, void* out_bf16, float* saved, void* stream) {
  if (!out || !saved) { rumpy_set_error("rumpy_enc_bn_train_keep: null pointer"); return RUMPY_E_ARG; }
  return enc_bn_train(p, out, out_bf16, saved, stream, "rumpy_enc_bn_train_keep");
}

extern "C" int rumpy_enc_conv(const rumpy_enc_conv_args* p, void* stream) {
  if (!p || !p->x || !p->w || !p->bias || !p->out) { rumpy_set_error("rumpy_enc_conv: null pointer"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->cin <= 0 || p->cin % 64 || p->cout <= 0 || p->cout % 64 || (p->stride != 1 && p->stride != 2)) {
    rumpy_set_error("rumpy_enc_conv: unsupported shape (cin=%d cout=%d stride=%d)", p->cin, p->cout, p->stride); return RUMPY_E_ARG; }
  EncConv a;
  a.x = (const uint4*)p->x; a.w = (const uint4*)p->w; a.bias = p->bias; a.out = (uint16_t*)p->out;
  a.w_lo = (const uint4*)p->w_lo; a.out32 = (float*)p->out;
  const bool o32 = p->out_fmt == RUMPY_FMT_F32;
  if ((p->out_fmt != 0 && !o32) || ((o32 || p->w_lo) && !(p->fmt == RUMPY_FMT_F16 && o32 && p->w_lo))) {
    rumpy_set_error("rumpy_enc_conv: out_fmt F32 and w_lo go together, with fmt F16 (the training forward pass of the encoder)"); return RUMPY_E_ARG; }
  a.N = p->N; a.H = p->H; a.W = p->W;
  a.Ho = (p->H - 1) / p->stride + 1; a.Wo = (p->W - 1) / p->stride + 1;      // 3x3, padding 1
  a.chn = p->cin / 64; a.ctn = p->cout / 64;
  a.tiles_x = (a.Wo + 15) / 16; a.tiles_y = (a.Ho + 3) / 4;
  a.neg_slope = p->neg_slope;
  const dim3 grid((unsigned)(p->N * a.tiles_x * a.tiles_y), (unsigned)a.ctn);
  if (o32) {
    if (p->stride == 1) hipLaunchKernelGGL((enc_conv_kernel<1, 4, RUMPY_FMT_F16, true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((enc_conv_kernel<2, 4, RUMPY_FMT_F16, true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else if (p->fmt == RUMPY_FMT_F16) {
    if (p->stride == 1) hipLaunchKernelGGL((enc_conv_kernel<1, 4, RUMPY_FMT_F16>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((enc_conv_kernel<2, 4, RUMPY_FMT_F16>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else if (p->fmt != RUMPY_FMT_BF16) { rumpy_set_error("rumpy_enc_conv: fmt %d", p->fmt); return RUMPY_E_ARG; }
  else if (p->stride == 1) hipLaunchKernelGGL((enc_conv_kernel<1, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((enc_conv_kernel<2, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return rumpy_check_launch("rumpy_enc_conv");
}

extern "C" int rumpy_enc_pool(const void* x, float* out, int32_t N, int32_t HW, int32_t C, int32_t fmt, void* stream) {
  if (!x || !out || N <= 0 || HW <= 0 || C <= 0 || C % 64 || (fmt != RUMPY_FMT_BF16 && fmt != RUMPY_FMT_F16)) { rumpy_set_error("rumpy_enc_pool: bad argument"); return RUMPY_E_ARG; }
  if (fmt == RUMPY_FMT_F16) hipLaunchKernelGGL(enc_pool_kernel<RUMPY_FMT_F16>, dim3(N, C / 64), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, out, HW, C);
  else hipLaunchKernelGGL(enc_pool_kernel<RUMPY_FMT_BF16>, dim3(N, C / 64), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, out, HW, C);
  return rumpy_check_launch("rumpy_enc_pool");
}

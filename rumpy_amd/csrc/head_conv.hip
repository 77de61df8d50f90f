// Head convolution (image, C <= 4 channels, fp32 NCHW -> 64*k channels NHWC bf16) and its weight gradient.
// K = 9*C <= 36 is far too small for MFMA to matter (0.09 % of the network's FLOPs), so both run on the fp32
// VALU and are exact fp32 like the reference; the image is read straight from the caller's NCHW tensor.
#include "common.hpp"
#include <cstdlib>

constexpr int HEAD_MAXC = 4;
// forward: one wave per workgroup - the headline shape has 73,728 pixels = 1152 waves = 4.5 per CU, which 256-thread
// workgroups would hand out as 288 workgroups on 256 CUs (two rounds, the second almost empty)
constexpr int HEAD_FWD_THREADS = 64;

// one thread = one pixel x 64 output channels; weights transposed to [k = (c,ky,kx)][co] in LDS (broadcast reads)
template <int C, int FMT = RUMPY_FMT_BF16>
__global__ void __launch_bounds__(HEAD_FWD_THREADS) head_fwd_kernel(const float* x, const float* const* x_ind, const float* __restrict__ w,
                                                       const float* __restrict__ b, uint16_t* __restrict__ out,
                                                       int N, int H, int W, int cout, float slope_m1) {
  if (x_ind) x = load_global_ptr(x_ind);          // the batch of this replay (rumpy_set_pointers)
  __shared__ __attribute__((aligned(16))) float sw[9 * HEAD_MAXC * 64];
  __shared__ float sb[64];
  const int ct = blockIdx.y;
  constexpr int K = 9 * C;
  for (int i = threadIdx.x; i < K * 64; i += HEAD_FWD_THREADS) {
    const int k = i >> 6, co = i & 63;
    sw[i] = w[(size_t)(ct * 64 + co) * K + k];  // OIHW: [co][c][ky][kx] -> k = c*9 + ky*3 + kx
  }
  if (threadIdx.x < 64) sb[threadIdx.x] = b[ct * 64 + threadIdx.x];
  __syncthreads();
  const size_t total = (size_t)N * H * W;
  const size_t p = (size_t)blockIdx.x * HEAD_FWD_THREADS + threadIdx.x;
  if (p >= total) return;
  const int xx = (int)(p % W);
  const int y = (int)((p / W) % H);
  const int n = (int)(p / ((size_t)W * H));
  float in[K];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xc = xx + kx - 1;
        in[c * 9 + ky * 3 + kx] = (yy >= 0 && yy < H && xc >= 0 && xc < W) ? x[((size_t)(n * C + c) * H + yy) * W + xc] : 0.f;
      }
  uint16_t* op = out + p * cout + ct * 64;
#pragma unroll
  for (int cb = 0; cb < 64; cb += 8) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = sb[cb + j];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float4 w0 = *reinterpret_cast<const float4*>(&sw[k * 64 + cb]);
      const float4 w1 = *reinterpret_cast<const float4*>(&sw[k * 64 + cb + 4]);
      const float v = in[k];
      acc[0] = fmaf(v, w0.x, acc[0]); acc[1] = fmaf(v, w0.y, acc[1]); acc[2] = fmaf(v, w0.z, acc[2]); acc[3] = fmaf(v, w0.w, acc[3]);
      acc[4] = fmaf(v, w1.x, acc[4]); acc[5] = fmaf(v, w1.y, acc[5]); acc[6] = fmaf(v, w1.z, acc[6]); acc[7] = fmaf(v, w1.w, acc[7]);
    }
    if (slope_m1 != 0.f) {                                 // v + (slope - 1) * min(v, 0)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(slope_m1, fminf(acc[j], 0.f), acc[j]);
    }
    if (FMT == RUMPY_FMT_F32) {       // the encoder's training forward pass: its first conv output as it is (this kernel is exact fp32)
      float* of = reinterpret_cast<float*>(out) + p * cout + ct * 64 + cb;
      *reinterpret_cast<float4*>(of) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *reinterpret_cast<float4*>(of + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    } else {
      constexpr int PF = FMT == RUMPY_FMT_F32 ? RUMPY_FMT_BF16 : FMT;
      const uint2 lo = pack4<PF>(acc[0], acc[1], acc[2], acc[3]);
      const uint2 hi = pack4<PF>(acc[4], acc[5], acc[6], acc[7]);
      *reinterpret_cast<uint4*>(op + cb) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  }
}

// weight gradient: persistent workgroups over 8x16-pixel tiles.  Thread = (pixel quarter q = tid >> 6,
// channel quad c4 = (tid >> 2) & 15 -> co 4*c4..4*c4+3, tap part kp = tid & 3 -> k = kp, kp+4, ... < 9*C):
// per pixel one 8-byte LDS read of dy feeds up to 36 FMAs.  The four pixel quarters are added through LDS in a
// fixed order; slab per workgroup: [cout_tiles][64][9*C + 1] (last column = bias sum).
__global__ void __launch_bounds__(256) head_wgrad_kernel(const float* x, const float* const* x_ind, const uint16_t* __restrict__ dy,
                                                         float* __restrict__ slab, int N, int C, int H, int W, int cout) {
  if (x_ind) x = load_global_ptr(x_ind);
  __shared__ float sx[HEAD_MAXC * HALO_PIX];
  __shared__ __attribute__((aligned(16))) uint16_t sdy[TH * TW * 64];
  __shared__ float red[4 * 64 * (9 * HEAD_MAXC + 1)];
  const int tid = threadIdx.x, q = tid >> 6, c4 = (tid >> 2) & 15, kp = tid & 3;
  const int ct = blockIdx.y;
  const int K = 9 * C;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int ntiles = N * tiles_y * tiles_x;
  float acc[9][4];
  int xoff[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
    const int k = kp + 4 * i;
    const int c = k / 9, t = k - 9 * c, ky = t / 3, kx = t - 3 * ky;
    xoff[i] = (k < K) ? c * HALO_PIX + ky * HALO_W + kx : -1;
  }
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const TileCoord tc = decode_tile(tile, tiles_x, tiles_y);
    __syncthreads();
    for (int i = tid; i < C * HALO_PIX; i += 256) {
      const int c = i / HALO_PIX, pix = i - c * HALO_PIX;
      const int r = pix / HALO_W, cc = pix - r * HALO_W;
      const int y = tc.ty * TH + r - 1, xx = tc.tx * TW + cc - 1;
      sx[i] = (y >= 0 && y < H && xx >= 0 && xx < W) ? x[((size_t)(tc.n * C + c) * H + y) * W + xx] : 0.f;
    }
    for (int i = tid; i < TH * TW * 8; i += 256) {
      const int pix = i >> 3, part8 = i & 7;
      const int y = tc.ty * TH + (pix >> 4), xx = tc.tx * TW + (pix & 15);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (y < H && xx < W) v = *reinterpret_cast<const uint4*>(dy + ((size_t)(tc.n * H + y) * W + xx) * cout + ct * 64 + part8 * 8);
      *reinterpret_cast<uint4*>(&sdy[pix * 64 + part8 * 8]) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int p = 0; p < 32; ++p) {
      const int pix = q * 32 + p;
      float d[4];
      unpack4_bf16(*reinterpret_cast<const uint2*>(&sdy[pix * 64 + 4 * c4]), d);
      const int base = (pix >> 4) * HALO_W + (pix & 15);
      if (kp == 0) { bs[0] += d[0]; bs[1] += d[1]; bs[2] += d[2]; bs[3] += d[3]; }
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        if (xoff[i] >= 0) {
          const float xv = sx[xoff[i] + base];
          acc[i][0] = fmaf(d[0], xv, acc[i][0]); acc[i][1] = fmaf(d[1], xv, acc[i][1]);
          acc[i][2] = fmaf(d[2], xv, acc[i][2]); acc[i][3] = fmaf(d[3], xv, acc[i][3]);
        }
      }
    }
  }
  // cross-quarter reduction in a fixed order
  const int KS = K + 1;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int k = kp + 4 * i;
    if (k < K) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(q * 64 + 4 * c4 + j) * KS + k] = acc[i][j];
    }
  }
  if (kp == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(q * 64 + 4 * c4 + j) * KS + K] = bs[j];
  }
  __syncthreads();
  float* s = slab + ((size_t)blockIdx.x * gridDim.y + ct) * 64 * KS;
  for (int e = tid; e < 64 * KS; e += 256)
    s[e] = (red[e] + red[64 * KS + e]) + (red[2 * 64 * KS + e] + red[3 * 64 * KS + e]);
}

// The same weight gradient on the matrix cores: dW[co][k] = sum over pixels of dy[px][co] * xpatch[px][k] (k = (c, ky, kx), 9C <= 36)
// is a GEMM with the PIXEL index as K, like the 64-channel weight gradients (wgrad_mfma.hip): A = dy^T through the transposing LDS
// read, B = the image patches, gathered from the fp32 halo tile and split into a bf16 high and low part (two MFMAs per fragment:
// 16 mantissa bits, |error| < 2^-16 relative - the reference runs this conv in fp32, and the VALU version above took 35 us, 2.6 % of
// the EDSR step, as an LDS-read-bound loop).  Wave w owns output channels 16w.. of the 64-channel tile; the four waves build the same
// B fragments.  Slab layout = head_wgrad_kernel's: [workgroup][cout tile][64][9C + 1] (last column = bias sum, from an MFMA against a
// ones fragment).  K order inside a 32-pixel step as in wgrad_mfma.hip: lane group g, element e -> tile row 2t + g/2, column
// 4*(g%2) + e (e < 4) or 8 + 4*(g%2) + e - 4.
typedef __attribute__((address_space(3))) short4v* head_lds_s4;
__device__ __forceinline__ bf16x8 head_tr_pair(const unsigned char* p) {
  const short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((head_lds_s4)(p));
  const short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((head_lds_s4)(p + 8 * PIX_STRIDE));
  union { short8v s; bf16x8 h; } c;
  c.s = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return c.h;
}
__global__ void __launch_bounds__(256) head_wgrad_mfma_kernel(const float* x, const float* const* x_ind, const uint16_t* __restrict__ dy,
                                                              float* __restrict__ slab, int N, int C, int H, int W, int cout) {
  if (x_ind) x = load_global_ptr(x_ind);
  __shared__ float sx[HEAD_MAXC * HALO_PIX];
  __shared__ __attribute__((aligned(16))) unsigned char sdy[TH * TW * PIX_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3, n16 = lane & 15;
  const int ct = blockIdx.y;
  const int K = 9 * C, KT = (K + 15) / 16;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int ntiles = N * tiles_y * tiles_x;
  f32x4 acc[3], bacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) acc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  // halo offset of this lane's k index in k tile kt (k = 16 kt + n16 -> c, ky, kx), -1 for the padding columns
  int koff[3];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    const int k = 16 * kt + n16;
    const int c = k / 9, t = k - 9 * c, ky = t / 3, kx = t - 3 * ky;
    koff[kt] = (k < K) ? c * HALO_PIX + ky * HALO_W + kx : -1;
  }
  const int rsel = g >> 1, colA = 4 * (g & 1) + q, colB = 4 * (g & 1);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const TileCoord tc = decode_tile(tile, tiles_x, tiles_y);
    __syncthreads();
    for (int i = tid; i < C * HALO_PIX; i += 256) {
      const int c = i / HALO_PIX, pix = i - c * HALO_PIX;
      const int r = pix / HALO_W, cc = pix - r * HALO_W;
      const int y = tc.ty * TH + r - 1, xx = tc.tx * TW + cc - 1;
      sx[i] = (y >= 0 && y < H && xx >= 0 && xx < W) ? x[((size_t)(tc.n * C + c) * H + y) * W + xx] : 0.f;
    }
    for (int i = tid; i < TH * TW * 8; i += 256) {
      const int pix = i >> 3, part8 = i & 7;
      const int y = tc.ty * TH + (pix >> 4), xx = tc.tx * TW + (pix & 15);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (y < H && xx < W) v = *reinterpret_cast<const uint4*>(dy + ((size_t)(tc.n * H + y) * W + xx) * cout + ct * 64 + part8 * 8);
      *reinterpret_cast<uint4*>(sdy + pix * PIX_STRIDE + part8 * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TH / 2; ++t) {
      const int row = 2 * t + rsel;
      const bf16x8 A = head_tr_pair(sdy + (row * TW + colA) * PIX_STRIDE + wave * 32 + p4 * 8);
      bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, ones, bacc, 0, 0, 0);
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        if (kt < KT) {
          bf16x8 bh, bl;
          const float* src = sx + (koff[kt] >= 0 ? koff[kt] : 0) + row * HALO_W + colB;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = src[(e < 4) ? e : e + 4];          // columns colB + e, then 8 + colB + (e - 4)
            if (koff[kt] < 0) v = 0.f;
            const __bf16 h = (__bf16)v;
            bh[e] = h;
            bl[e] = (__bf16)(v - (float)h);
          }
          acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, bh, acc[kt], 0, 0, 0);
          acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, bl, acc[kt], 0, 0, 0);
        }
      }
    }
  }
  // D row = co (4g + e) of the wave's 16, D column = k index n16 of the k tile
  const int KS = K + 1;
  float* s = slab + ((size_t)blockIdx.x * gridDim.y + ct) * 64 * KS;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    const int k = 16 * kt + n16;
    if (kt < KT && k < K) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[(16 * wave + 4 * g + e) * KS + k] = acc[kt][e];
    }
  }
  if (n16 == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) s[(16 * wave + 4 * g + e) * KS + K] = bacc[e];
  }
}

// Deterministic reduction of the per-workgroup slabs: 16 outputs x 16 slab parts per 256-thread block; part p adds slabs
// p, p+16, p+32, .. in order (independent loads, in flight together), the 16 parts are added in a fixed order through LDS.
__global__ void __launch_bounds__(256) head_wgrad_reduce_kernel(const float* __restrict__ slab, int nwg, int C, int cout, float scale,
                                                                float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ float part[16][17];
  const int K = 9 * C;
  const int total = cout * (K + 1);
  const int el = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s0 = 0.f, s1 = 0.f;
  if (e < total) {
    int g = p;
    for (; g + 16 < nwg; g += 32) {
      s0 += slab[(size_t)g * total + e];
      s1 += slab[(size_t)(g + 16) * total + e];
    }
    if (g < nwg) s0 += slab[(size_t)g * total + e];
  }
  part[p][el] = s0 + s1;
  __syncthreads();
  if (p == 0 && e < total) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += part[i][el];
    const int co = e / (K + 1), k = e - co * (K + 1);
    if (k < K) gw[(size_t)co * K + k] = s * scale;
    else gb[co] = s * scale;
  }
}

// three workgroups per CU (24 KB of LDS each): their load / barrier / MFMA phases overlap
static int head_wgrad_grid() { return 3 * rumpy_device_cus(); }

extern "C" int rumpy_head_wgrad_slabs(int32_t N, int32_t H, int32_t W) {
  const int ntiles = N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  return ntiles < head_wgrad_grid() ? ntiles : head_wgrad_grid();
}
extern "C" int64_t rumpy_head_wgrad_slab_floats(int32_t C, int32_t cout) {
  return (int64_t)head_wgrad_grid() * cout * (9 * C + 1);
}

extern "C" int rumpy_head_fwd(const rumpy_head_fwd_args* p, void* stream) {
  if (!p || !p->x || !p->w || !p->b || !p->out) { rumpy_set_error("rumpy_head_fwd: null pointer"); return RUMPY_E_ARG; }
  if (p->C < 1 || p->C > HEAD_MAXC || p->cout <= 0 || p->cout % 64 || p->N <= 0 || p->H <= 0 || p->W <= 0) {
    rumpy_set_error("rumpy_head_fwd: unsupported shape (C=%d cout=%d)", p->C, p->cout); return RUMPY_E_ARG; }
  const size_t total = (size_t)p->N * p->H * p->W;
  dim3 grid((unsigned)((total + HEAD_FWD_THREADS - 1) / HEAD_FWD_THREADS), p->cout / 64);
  hipStream_t s = (hipStream_t)stream;
  uint16_t* o = (uint16_t*)p->out;
  if (p->fmt != RUMPY_FMT_BF16 && p->fmt != RUMPY_FMT_F16 && p->fmt != RUMPY_FMT_F32) { rumpy_set_error("rumpy_head_fwd: bad fmt %d", p->fmt); return RUMPY_E_ARG; }
#define HEAD_LAUNCH(C_, F_) hipLaunchKernelGGL((head_fwd_kernel<C_, F_>), grid, dim3(HEAD_FWD_THREADS), 0, s, p->x, p->x_ind, p->w, p->b, o, p->N, p->H, p->W, p->cout, p->neg_slope_m1)
  if (p->fmt == RUMPY_FMT_F32) {
    switch (p->C) {
      case 1: HEAD_LAUNCH(1, RUMPY_FMT_F32); break;
      case 2: HEAD_LAUNCH(2, RUMPY_FMT_F32); break;
      case 3: HEAD_LAUNCH(3, RUMPY_FMT_F32); break;
      default: HEAD_LAUNCH(4, RUMPY_FMT_F32); break;
    }
  } else if (p->fmt == RUMPY_FMT_F16) {
    switch (p->C) {
      case 1: HEAD_LAUNCH(1, RUMPY_FMT_F16); break;
      case 2: HEAD_LAUNCH(2, RUMPY_FMT_F16); break;
      case 3: HEAD_LAUNCH(3, RUMPY_FMT_F16); break;
      default: HEAD_LAUNCH(4, RUMPY_FMT_F16); break;
    }
  } else {
    switch (p->C) {
      case 1: HEAD_LAUNCH(1, RUMPY_FMT_BF16); break;
      case 2: HEAD_LAUNCH(2, RUMPY_FMT_BF16); break;
      case 3: HEAD_LAUNCH(3, RUMPY_FMT_BF16); break;
      default: HEAD_LAUNCH(4, RUMPY_FMT_BF16); break;
    }
  }
#undef HEAD_LAUNCH
  return rumpy_check_launch("rumpy_head_fwd");
}

extern "C" int rumpy_head_wgrad(const rumpy_head_wgrad_args* p, void* stream) {
  if (!p || !p->x || !p->dy || !p->slab || (p->gw && !p->gb)) { rumpy_set_error("rumpy_head_wgrad: null pointer"); return RUMPY_E_ARG; }
  if (p->C < 1 || p->C > HEAD_MAXC || p->cout <= 0 || p->cout % 64 || p->N <= 0 || p->H <= 0 || p->W <= 0) {
    rumpy_set_error("rumpy_head_wgrad: unsupported shape"); return RUMPY_E_ARG; }
  const int ntiles = p->N * ((p->H + TH - 1) / TH) * ((p->W + TW - 1) / TW);
  const int nwg = ntiles < head_wgrad_grid() ? ntiles : head_wgrad_grid();
  hipStream_t s = (hipStream_t)stream;
  static const bool valu = getenv("RUMPY_HEAD_WGRAD_VALU") != nullptr;     // A/B switch: the fp32 VALU version
  if (valu) hipLaunchKernelGGL(head_wgrad_kernel, dim3(nwg, p->cout / 64), dim3(256), 0, s, p->x, p->x_ind, (const uint16_t*)p->dy, p->slab,
                               p->N, p->C, p->H, p->W, p->cout);
  else hipLaunchKernelGGL(head_wgrad_mfma_kernel, dim3(nwg, p->cout / 64), dim3(256), 0, s, p->x, p->x_ind, (const uint16_t*)p->dy, p->slab,
                          p->N, p->C, p->H, p->W, p->cout);
  const int total = p->cout * (9 * p->C + 1);
  if (p->gw)          // (gw == NULL: rumpy_finish_reduce adds the slabs up together with the other layers')
  hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3((total + 15) / 16), dim3(256), 0, s, p->slab, nwg, p->C, p->cout,
                     p->scale, p->gw, p->gb);
  return rumpy_check_launch("rumpy_head_wgrad");
}

// ------------------------------------------------------------------------------------------------------
// the pointer table behind x_ind / target_ind
// ------------------------------------------------------------------------------------------------------
__global__ void set_pointers_kernel(const void** table, const void* p0, const void* p1) { table[0] = p0; table[1] = p1; }

extern "C" int rumpy_set_pointers(void* table, const void* p0, const void* p1, void* stream) {
  if (!table) { rumpy_set_error("rumpy_set_pointers: null table"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(set_pointers_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const void**)table, p0, p1);
  return rumpy_check_launch("rumpy_set_pointers");
}

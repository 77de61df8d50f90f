// Optimizer, filter packing and eval post-processing kernels (all bandwidth-trivial next to the convolutions).
#include "common.hpp"

// torch.optim.Adam (no amsgrad, no weight decay) in torch's operation order:
//   m.lerp_(g, 1-b1); v = v*b2 + (1-b2) g*g; denom = sqrt(v)/sqrt(bias_c2) + eps; p -= (lr/bias_c1) * m/denom
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, const rumpy_adam_hyper* __restrict__ hp, rumpy_adam_hyper hv, const float* __restrict__ sumsq,
                            const uint32_t* __restrict__ skip_if) {
  if (skip_if && *skip_if) return;      // the step's watchdog word is set (a hand-off of its persistent launches timed out): its gradients never reach the weights
  const rumpy_adam_hyper h = hp ? *hp : hv;
  float gm = h.grad_mult;
  if (h.max_norm > 0.f && sumsq) {
    // nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    const float total = sqrtf(sumsq[0]) * fabsf(gm);
    float coef = h.max_norm / (total + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    gm *= coef;
  }
  const float step = h.lr / h.bias_c1;
  const float w1 = 1.f - h.beta1, w2 = 1.f - h.beta2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)      // no fused multiply-adds: every product is rounded where torch rounds it (finish.hip::adam_one is the same arithmetic)
    const float gi = g[i] * gm;
    float mi = m[i], vi = v[i];
    mi = mi + w1 * (gi - mi);
    vi = vi * h.beta2 + w2 * gi * gi;
    const float denom = sqrtf(vi) / h.sqrt_bias_c2 + h.eps;
    p[i] = p[i] - step * (mi / denom);
    m[i] = mi; v[i] = vi;
  }
}

__global__ void sumsq_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
  __shared__ float red[256];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s = fmaf(g[i], g[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void sum_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// ---- filter packing --------------------------------------------------------------------------------------
// kind 0 forward image : [ct][ch][wave][s = tap*2+half][lane][8]   A-fragment of conv3x3_kernel:
//   lane (r = lane&15, g = lane>>4), element e -> W[co(ct, c = 16*wave + r)][ci = 64*ch + 32*half + 8g + e][ky][kx]
// kind 0 dgrad image   : roles swapped and the filter flipped; "output" tile ct' = forward cin chunk, "input" chunk ch' =
//   forward cout tile:  W[co(ch', cc = 32*half + 8g + e)][ci = 64*ct' + 16*wave + r][2-ky][2-kx]
// co(t, c) = shuffle ? 4*c + t : 64*t + c
// kind 2 (tail, Cout <= 4, Cin = 64): forward image [s][lane][8]: row r < Cout real, zero otherwise;
//   dgrad image [wave][ks][lane][8]: row ci = 16*wave + r, k = 32*ks + 8g + e -> tap = k>>2, c = k&3: W[c][ci][2-ky][2-kx]
template <int FMT> __device__ __forceinline__ uint16_t pack1(float f) { return (uint16_t)(pack2<FMT>(f, 0.f) & 0xffffu); }
// PFMT = RUMPY_FMT_F16_RESIDUAL: what fp16 drops of the value, itself as fp16 (the second image of a filter whose rounding must not show:
// evaluation plans run the conv on the fp16 image and add the conv on this one)
__device__ __forceinline__ float f16_residual(float v) { return v - (float)(_Float16)v; }
template <int PFMT> __device__ __forceinline__ uint2 pk4(float a, float b, float c, float d) {
  if (PFMT == RUMPY_FMT_F16_RESIDUAL) return pack4<RUMPY_FMT_F16>(f16_residual(a), f16_residual(b), f16_residual(c), f16_residual(d));
  return pack4 < PFMT == RUMPY_FMT_F16_RESIDUAL ? RUMPY_FMT_F16 : PFMT > (a, b, c, d);
}
template <int FMT>
__device__ __forceinline__ void pack_item(const rumpy_pack_item& it) {
  uint16_t* wf = (uint16_t*)it.w_fwd;
  uint16_t* wd = (uint16_t*)it.w_dgrad;
  const int Co = it.cout, Ci = it.cin;
  if (it.kind == 0) {
    const int ctn = Co / 64, chn = Ci / 64;
    const size_t total8 = (size_t)Co * Ci * 9 / 8;  // 16-byte vectors (8 consecutive K elements of one lane) of each image
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total8; v += (size_t)gridDim.x * blockDim.x) {
      const int lane = (int)(v & 63);
      size_t r2 = v >> 6;
      const int s = (int)(r2 % 18); r2 /= 18;
      const int wave = (int)(r2 & 3); r2 >>= 2;
      const int r = lane & 15, g = lane >> 4;
      const int half = s & 1, tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky;
      if (wf) {  // forward: r2 = ct*chn + ch; the 8 elements are 8 consecutive input channels (stride 9 floats in the master copy)
        const int ch = (int)(r2 % chn), ct = (int)(r2 / chn);
        const int c = 16 * wave + r;
        const int co = it.shuffle ? 4 * c + ct : 64 * ct + c;
        const float* src = it.w + ((size_t)co * Ci + 64 * ch + 32 * half + 8 * g) * 9 + ky * 3 + kx;
        const uint2 lo = pk4<FMT>(src[0], src[9], src[18], src[27]), hi = pk4<FMT>(src[36], src[45], src[54], src[63]);
        reinterpret_cast<uint4*>(wf)[v] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
      if (wd) {  // dgrad: r2 = ct'*ctn + ch'; the 8 elements are 8 output channels (stride Ci*9, or 4*Ci*9 when shuffled)
        const int chp = (int)(r2 % ctn), ctp = (int)(r2 / ctn);
        const int cc0 = 32 * half + 8 * g;
        const int ci = 64 * ctp + 16 * wave + r;
        const size_t co0 = it.shuffle ? (size_t)4 * cc0 + chp : (size_t)64 * chp + cc0;
        const size_t cstep = (size_t)(it.shuffle ? 4 : 1) * Ci * 9;
        const float* src = it.w + (co0 * Ci + ci) * 9 + (2 - ky) * 3 + (2 - kx);
        const uint2 lo = pk4<FMT>(src[0], src[cstep], src[2 * cstep], src[3 * cstep]);
        const uint2 hi = pk4<FMT>(src[4 * cstep], src[5 * cstep], src[6 * cstep], src[7 * cstep]);
        reinterpret_cast<uint4*>(wd)[v] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    if (it.b_packed) {
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Co; i += gridDim.x * blockDim.x) {
        const int ct = i >> 6, c = i & 63;
        it.b_packed[i] = it.b[it.shuffle ? 4 * c + ct : i];
      }
    }
    (void)ctn;
  } else {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 18 * 64 * 8; i += gridDim.x * blockDim.x) {
      const int e = i & 7, lane = (i >> 3) & 63, s = i >> 9;
      const int r = lane & 15, g = lane >> 4;
      const int half = s & 1, tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky;
      const int ci = 32 * half + 8 * g + e;
      const float wv = (r < Co) ? it.w[((size_t)r * Ci + ci) * 9 + ky * 3 + kx] : 0.f;
      wf[i] = (r < Co) ? pack1<FMT>(wv) : (uint16_t)0;
      if (FMT == RUMPY_FMT_F16) {          // second image behind the first: the rounding residual w - fp16(w), itself in fp16 (tail_fwd_kernel)
        float back[4];
        unpack4<FMT>(make_uint2((uint32_t)wf[i], 0u), back);
        wf[18 * 64 * 8 + i] = (r < Co) ? pack1<FMT>(wv - back[0]) : (uint16_t)0;
      }
    }
    if (wd) {
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 4 * 2 * 64 * 8; i += gridDim.x * blockDim.x) {
        const int e = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) & 1, wave = i >> 10;
        const int r = lane & 15, g = lane >> 4;
        const int k = 32 * ks + 8 * g + e, tap = k >> 2, c = k & 3;
        uint16_t v = 0;
        if (tap < 9 && c < Co) {
          const int ky = tap / 3, kx = tap - 3 * ky;
          v = pack1<FMT>(it.w[((size_t)c * Ci + 16 * wave + r) * 9 + (2 - ky) * 3 + (2 - kx)]);
        }
        wd[i] = v;
      }
    }
  }
}
__global__ void pack_kernel(const rumpy_pack_item* __restrict__ items) {
  const rumpy_pack_item it = items[blockIdx.y];
  if (it.fmt == RUMPY_FMT_F16) pack_item<RUMPY_FMT_F16>(it);
  else if (it.fmt == RUMPY_FMT_F16_RESIDUAL) pack_item<RUMPY_FMT_F16_RESIDUAL>(it);
  else pack_item<RUMPY_FMT_BF16>(it);
}

// ---- eval post-processing: clip, RGB -> YCbCr ('jpg' matrix), squared Y error vs the clipped reference ----
__global__ void eval_post_kernel(rumpy_eval_post_args a) {
  __shared__ float red[256];
  const size_t hw = (size_t)a.H * a.W, total = (size_t)a.N * hw;
  const float bias_c = 128.f * (1.f / 255.f);
  float sse = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t n = i / hw, p = i - n * hw;
    const size_t o = n * 3 * hw + p;
    const float r = fminf(fmaxf(a.out[o], 0.f), 1.f), g = fminf(fmaxf(a.out[o + hw], 0.f), 1.f), b = fminf(fmaxf(a.out[o + 2 * hw], 0.f), 1.f);
    const float y = 0.299f * r + 0.587f * g + 0.114f * b;
    if (a.rgb) { a.rgb[o] = r; a.rgb[o + hw] = g; a.rgb[o + 2 * hw] = b; }
    if (a.ycbcr) {
      a.ycbcr[o] = y;
      a.ycbcr[o + hw] = bias_c + (-0.168736f * r - 0.331264f * g + 0.5f * b);
      a.ycbcr[o + 2 * hw] = bias_c + (0.5f * r - 0.418688f * g - 0.081312f * b);
    }
    if (a.ref) {
      const float rr = fminf(fmaxf(a.ref[o], 0.f), 1.f), rg = fminf(fmaxf(a.ref[o + hw], 0.f), 1.f), rb = fminf(fmaxf(a.ref[o + 2 * hw], 0.f), 1.f);
      const float d = y - (0.299f * rr + 0.587f * rg + 0.114f * rb);
      sse = fmaf(d, d, sse);
    }
  }
  if (a.sse_partial) {
    red[threadIdx.x] = sse;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
      if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) a.sse_partial[blockIdx.x] = red[0];
  }
}

// ---- image save: fp32 planes -> interleaved uint8 (rumpy/sr_tools/visualization.py:56: np.clip(im * 255 / max_val, 0, 255).astype(np.uint8),
// i.e. TRUNCATION towards zero, after the CHW -> HWC transpose of :53-54) ----
__global__ void to_uint8_hwc_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, int N, int C, int H, int W, float max_val) {
  const size_t hw = (size_t)H * W, total = (size_t)N * hw * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t c = i % C, p = (i / C) % hw, n = i / (C * hw);
    const float v = fminf(fmaxf(src[(n * C + c) * hw + p] * 255.f / max_val, 0.f), 255.f);      // numpy's order: (im * 255) / max_val, both in fp32
    dst[i] = (unsigned char)(int)v;
  }
}
extern "C" int rumpy_to_uint8_hwc(const float* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, float max_val, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || !(max_val > 0.f)) { rumpy_set_error("rumpy_to_uint8_hwc: bad argument"); return RUMPY_E_ARG; }
  const size_t total = (size_t)N * C * H * W;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(to_uint8_hwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (unsigned char*)dst, N, C, H, W, max_val);
  return rumpy_check_launch("rumpy_to_uint8_hwc");
}

extern "C" int rumpy_adam_step(const rumpy_adam_args* a, void* stream) {
  if (!a || !a->p || !a->g || !a->m || !a->v || a->n <= 0) { rumpy_set_error("rumpy_adam_step: bad argument"); return RUMPY_E_ARG; }
  if (!a->hyper && !(a->hyper_value.bias_c1 > 0.f && a->hyper_value.sqrt_bias_c2 > 0.f)) { rumpy_set_error("rumpy_adam_step: neither a hyper pointer nor by-value hyper-parameters"); return RUMPY_E_ARG; }
  size_t blocks = ((size_t)a->n + 255) / 256;
  const size_t cap = (size_t)rumpy_device_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a->p, a->g, a->m, a->v, (size_t)a->n, a->hyper, a->hyper_value, a->sumsq, a->skip_if);
  return rumpy_check_launch("rumpy_adam_step");
}
extern "C" int rumpy_sumsq(const rumpy_sumsq_args* a, void* stream) {
  if (!a || !a->g || !a->partial || !a->out || a->n <= 0) { rumpy_set_error("rumpy_sumsq: bad argument"); return RUMPY_E_ARG; }
  size_t blocks = ((size_t)a->n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a->g, (size_t)a->n, a->partial);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, s, a->partial, (int)blocks, a->out);
  return rumpy_check_launch("rumpy_sumsq");
}
extern "C" int rumpy_pack_weights(const rumpy_pack_item* items_device, int32_t nitems, void* stream) {
  if (!items_device || nitems <= 0) { rumpy_set_error("rumpy_pack_weights: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(pack_kernel, dim3(32, nitems), dim3(256), 0, (hipStream_t)stream, items_device);
  return rumpy_check_launch("rumpy_pack_weights");
}
extern "C" int rumpy_eval_post(const rumpy_eval_post_args* a, void* stream) {
  if (!a || !a->out || a->N <= 0 || a->H <= 0 || a->W <= 0) { rumpy_set_error("rumpy_eval_post: bad argument"); return RUMPY_E_ARG; }
  if (a->ref && (!a->sse_partial || !a->sse)) { rumpy_set_error("rumpy_eval_post: ref needs sse_partial and sse"); return RUMPY_E_ARG; }
  const size_t total = (size_t)a->N * a->H * a->W;
  size_t blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(eval_post_kernel, dim3((unsigned)blocks), dim3(256), 0, s, *a);
  if (a->sse_partial && a->sse) hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, s, a->sse_partial, (int)blocks, a->sse);
  return rumpy_check_launch("rumpy_eval_post");
}

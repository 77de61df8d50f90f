// End-of-step housekeeping in two launches instead of seven (wgrad_reduce 25 + pack 17 + tail_wgrad_reduce 10 + adam 9 +
// head_wgrad_reduce 7.5 us, each a dependent launch):
//   rumpy_finish_reduce : the slab reductions of the 64-channel convs, the tail conv and the head conv in ONE launch (block ranges = roles;
//                         the arithmetic of wgrad_reduce_kernel / tail_wgrad_reduce_kernel / head_wgrad_reduce_kernel, same summation order:
//                         bitwise the separate launches);
//   rumpy_adam_pack     : torch.optim.Adam (adam_kernel's arithmetic) AND the re-packing of the bf16 MFMA filter images in ONE launch.  A
//                         workgroup owns a closed set of weights - 16 output channels x 32 input channels x 9 taps of a conv: exactly 9
//                         fragments of the forward image and 9 x 4 half-fragments (whole 16-byte vectors) of the data-gradient image -
//                         updates them in the master copy's order (rows of 288 contiguous floats: coalesced), keeps the new values in LDS and
//                         emits both images' vectors from there (pack_kernel gathers 4-byte values at a 36-byte stride instead).  Biases (+ their
//                         packed copy), the tail conv (+ its two images) and every other parameter range are items of the same table.
// Data-parallel runs call them around the all-reduce: reduce -> RCCL -> adam_pack.
#include "common.hpp"

// ---------------------------------------------------------------------------------------------------------------- reductions
// Four output-channel rows (4 x 576 values: contiguous in every slab AND in the gradient tensor) per 256-thread block.  16-byte loads, the
// njobs loads of a thread independent of each other (one add chain in job order, as wgrad_reduce_kernel: same bits); the [tap][ci] order of a
// slab row is turned into the tensor's [ci][tap] order through LDS so that both sides are coalesced (the one-element-per-thread form wrote
// 4-byte values at a 36-byte stride and read 4 bytes per lane: 2.1 TB/s over 70 MB of slabs).
constexpr int FIN_ROWS = 4;
__device__ __forceinline__ void fin_reduce_rows(const rumpy_reduce_item& it, int r0, int nrows, float* __restrict__ T) {
  const int tid = threadIdx.x;
  const int rows = (it.co_count - r0 < nrows) ? it.co_count - r0 : nrows;
  if (rows <= 0) return;
  const size_t stride4 = (size_t)it.slab_stride >> 2;
  for (int v = tid; v < rows * 144; v += 256) {
    const float4* sp = reinterpret_cast<const float4*>(it.slab + (size_t)r0 * 576) + v;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 9 <= it.njobs; k += 9) {            // nine loads in flight (a layer at 48 x 48 has nine jobs), added in job order
      float4 t[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) t[u] = sp[(size_t)(k + u) * stride4];
#pragma unroll
      for (int u = 0; u < 9; ++u) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
    }
    for (; k < it.njobs; ++k) {
      const float4 t = sp[(size_t)k * stride4];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const int e = 4 * v, cl = e / 576, rem = e - cl * 576, tap = rem >> 6, ci = rem & 63;
    float* d = T + cl * 576 + ci * 9 + tap;
    d[0] = s.x * it.scale; d[9] = s.y * it.scale; d[18] = s.z * it.scale; d[27] = s.w * it.scale;
  }
  __syncthreads();
  for (int idx = tid; idx < rows * 576; idx += 256) {
    const int cl = idx / 576, j = idx - cl * 576, c = r0 + cl;
    const int co = it.co_mode ? 4 * c + it.co_off : it.co_off + c;
    it.gw[((size_t)co * it.ci_total + it.ci_off) * 9 + j] = T[idx];
  }
  if (it.write_bias && r0 == 0) {
    for (int c = tid; c < it.co_count; c += 256) {
      float s = 0.f;
      for (int k = 0; k < it.njobs; ++k) s += it.slab[(size_t)k * it.slab_stride + 16 * it.mt * 576 + c];
      const int co = it.co_mode ? 4 * c + it.co_off : it.co_off + c;
      it.gb[co] = s * it.scale;
    }
  }
}
// block = 16 elements x 16 slab groups; group p adds slabs p, p+16, .. (independent loads), the 16 partial sums are added in a fixed order
__device__ __forceinline__ float fin_slab_sum16(const float* __restrict__ slabs, int nslabs, size_t stride, int src, bool live, float (&part)[16][17]) {
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  float s0 = 0.f, s1 = 0.f;
  if (live) {
    int k = grp;
#pragma unroll 4
    for (; k + 16 < nslabs; k += 32) { s0 += slabs[(size_t)k * stride + src]; s1 += slabs[(size_t)(k + 16) * stride + src]; }
    if (k < nslabs) s0 += slabs[(size_t)k * stride + src];
  }
  part[grp][el] = s0 + s1;
  __syncthreads();
  float s = 0.f;
  if (grp == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s += part[i][el];
  }
  return s;
}
__global__ void __launch_bounds__(256) finish_reduce_kernel(rumpy_finish_reduce_args a, int nb_main, int nb_tail) {
  __shared__ float part[16][17];
  __shared__ float T[FIN_ROWS * 576];
  const int b = blockIdx.x;
  if (b < nb_main) {
    fin_reduce_rows(a.items[b >> 4], FIN_ROWS * (b & 15), FIN_ROWS, T);
  } else if (b < nb_main + nb_tail) {          // tail conv: slabs [nslabs][16*576 + 16], rows 0 .. C-1 used
    const int el = threadIdx.x & 15, nelem = a.tail_C * 576;
    const int e = (b - nb_main) * 16 + el;
    const bool live = e < nelem + a.tail_C;
    const float s = fin_slab_sum16(a.tail_slabs, a.tail_nslabs, 16 * 576 + 16, (e < nelem) ? e : 16 * 576 + (e - nelem), live, part) * a.tail_scale;
    if ((threadIdx.x >> 4) == 0 && live) {
      if (e < nelem) {
        const int co = e / 576, rem = e - co * 576, tap = rem >> 6, ci = rem & 63;
        a.tail_gw[((size_t)co * 64 + ci) * 9 + tap] = s;
      } else a.tail_gb[e - nelem] = s;
    }
  } else {                                     // head conv: slabs [nwg][cout * (9C + 1)]
    const int el = threadIdx.x & 15, K = 9 * a.head_C, total = a.head_cout * (K + 1);
    const int e = (b - nb_main - nb_tail) * 16 + el;
    const bool live = e < total;
    const float s = fin_slab_sum16(a.head_slabs, a.head_nslabs, (size_t)total, e, live, part);
    if ((threadIdx.x >> 4) == 0 && live) {
      const int co = e / (K + 1), k = e - co * (K + 1);
      if (k < K) a.head_gw[(size_t)co * K + k] = s * a.head_scale;
      else a.head_gb[co] = s * a.head_scale;
    }
  }
}

extern "C" int rumpy_finish_reduce(const rumpy_finish_reduce_args* p, void* stream) {
  if (!p || p->nitems < 0 || (p->nitems > 0 && !p->items)) { rumpy_set_error("rumpy_finish_reduce: bad argument"); return RUMPY_E_ARG; }
  if (p->tail_slabs && (!p->tail_gw || !p->tail_gb || p->tail_nslabs <= 0 || p->tail_C < 1 || p->tail_C > 4)) { rumpy_set_error("rumpy_finish_reduce: bad tail arguments"); return RUMPY_E_ARG; }
  if (p->head_slabs && (!p->head_gw || !p->head_gb || p->head_nslabs <= 0 || p->head_C < 1 || p->head_C > 4 || p->head_cout <= 0)) { rumpy_set_error("rumpy_finish_reduce: bad head arguments"); return RUMPY_E_ARG; }
  const int nb_main = 16 * p->nitems;           // 64 output channels per item at most, FIN_ROWS per block
  const int nb_tail = p->tail_slabs ? (p->tail_C * 576 + p->tail_C + 15) / 16 : 0;
  const int nb_head = p->head_slabs ? (p->head_cout * (9 * p->head_C + 1) + 15) / 16 : 0;
  if (nb_main + nb_tail + nb_head == 0) return RUMPY_OK;
  hipLaunchKernelGGL(finish_reduce_kernel, dim3(nb_main + nb_tail + nb_head), dim3(256), 0, (hipStream_t)stream, *p, nb_main, nb_tail);
  return rumpy_check_launch("rumpy_finish_reduce");
}

// ---------------------------------------------------------------------------------------------------------------- Adam + re-pack
struct AdamCoef { float gm, step, w1, w2, beta2, sb2, eps; };
__device__ __forceinline__ AdamCoef adam_coef(const rumpy_adam_hyper& h, const float* sumsq) {
  AdamCoef c;
  float gm = h.grad_mult;
  if (h.max_norm > 0.f && sumsq) {            // nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    const float total = sqrtf(sumsq[0]) * fabsf(gm);
    float coef = h.max_norm / (total + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    gm *= coef;
  }
  c.gm = gm; c.step = h.lr / h.bias_c1; c.w1 = 1.f - h.beta1; c.w2 = 1.f - h.beta2; c.beta2 = h.beta2; c.sb2 = h.sqrt_bias_c2; c.eps = h.eps;
  return c;
}
// adam_kernel's update of one element (optim.hip) on register values
__device__ __forceinline__ void adam_reg(const AdamCoef& c, float& p, float graw, float& m, float& v) {
#pragma clang fp contract(off)      // every product is rounded where torch rounds it (clip / mean factor first, then the moments): same bits as adam_kernel
  const float gi = graw * c.gm;
  m = m + c.w1 * (gi - m);
  v = v * c.beta2 + c.w2 * gi * gi;
  const float denom = sqrtf(v) / c.sb2 + c.eps;
  p = p - c.step * (m / denom);
}
// ... of one element in memory, returns the new parameter value
__device__ __forceinline__ float adam_one(const AdamCoef& c, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t i) {
  float pi = p[i], mi = m[i], vi = v[i];
  adam_reg(c, pi, g[i], mi, vi);
  p[i] = pi; m[i] = mi; v[i] = vi;
  return pi;
}

constexpr int AP_ROW = 288;                   // 32 input channels x 9 taps: contiguous in the OIHW master copy
__global__ void __launch_bounds__(256) adam_pack_kernel(rumpy_adam_pack_args a) {
  __shared__ __attribute__((aligned(16))) float P[16 * AP_ROW];          // 18 KB: the updated values of this workgroup's set
  if (a.skip_if && *a.skip_if) return;      // (uniform over the grid) the step's watchdog word is set: weights, moments and images stay as they are
  const rumpy_update_item it = a.items[blockIdx.x];
  const rumpy_adam_hyper h = a.hyper ? *a.hyper : a.hyper_value;
  const AdamCoef c = adam_coef(h, a.sumsq);
  const int tid = threadIdx.x;
  if (it.kind == 0) {
    // ---- 16 output channels (quarter q of cout tile ct) x 32 input channels (half hf of cin chunk ch) x 9 taps of a 64-multiple conv ----
    const int chn = it.cin / 64, ctn = it.cout / 64;
    if ((it.woff & 3) == 0) {
      // rows of 288 contiguous floats, 16 bytes per lane (the 4-byte form ran at 2 TB/s); element-wise the same arithmetic
      for (int v = tid; v < 16 * (AP_ROW / 4); v += 256) {
        const int cl = v / (AP_ROW / 4), j4 = v - cl * (AP_ROW / 4);
        const int cc = 16 * it.q + cl;
        const int co = it.shuffle ? 4 * cc + it.ct : 64 * it.ct + cc;
        const size_t i = (size_t)it.woff + ((size_t)co * it.cin + 64 * it.ch + 32 * it.hf) * 9 + 4 * j4;
        const float4 g4 = *reinterpret_cast<const float4*>(a.g + i);
        float4 p4 = *reinterpret_cast<const float4*>(a.p + i), m4 = *reinterpret_cast<const float4*>(a.m + i), v4 = *reinterpret_cast<const float4*>(a.v + i);
        adam_reg(c, p4.x, g4.x, m4.x, v4.x); adam_reg(c, p4.y, g4.y, m4.y, v4.y); adam_reg(c, p4.z, g4.z, m4.z, v4.z); adam_reg(c, p4.w, g4.w, m4.w, v4.w);
        *reinterpret_cast<float4*>(a.p + i) = p4; *reinterpret_cast<float4*>(a.m + i) = m4; *reinterpret_cast<float4*>(a.v + i) = v4;
        *reinterpret_cast<float4*>(P + cl * AP_ROW + 4 * j4) = p4;
      }
    } else {
      for (int idx = tid; idx < 16 * AP_ROW; idx += 256) {
        const int cl = idx / AP_ROW, j = idx - cl * AP_ROW;
        const int cc = 16 * it.q + cl;
        const int co = it.shuffle ? 4 * cc + it.ct : 64 * it.ct + cc;
        P[idx] = adam_one(c, a.p, a.g, a.m, a.v, (size_t)it.woff + ((size_t)co * it.cin + 64 * it.ch + 32 * it.hf) * 9 + j);
      }
    }
    __syncthreads();
    uint4* wf = reinterpret_cast<uint4*>(it.w_fwd);
    uint4* wd = reinterpret_cast<uint4*>(it.w_dgrad);
    for (int vi = tid; vi < 576; vi += 256) {
      {   // forward image: fragment s = tap*2 + hf of quarter q; lane (r, g4) holds input channels 8*g4 .. +7 (of this half) of output channel r
        const int tap = vi >> 6, lane = vi & 63, r = lane & 15, g4 = lane >> 4;
        const float* src = P + r * AP_ROW + (8 * g4) * 9 + tap;
        const uint2 lo = pack4_bf16(src[0], src[9], src[18], src[27]), hi = pack4_bf16(src[36], src[45], src[54], src[63]);
        wf[(((size_t)(it.ct * chn + it.ch) * 4 + it.q) * 18 + (tap * 2 + it.hf)) * 64 + lane] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
      if (wd) {   // data-gradient image (roles swapped, filter flipped): 8 consecutive OUTPUT channels per vector
        const int r = vi & 15, gg = (vi >> 4) & 1, wv = (vi >> 5) & 1, tapd = vi >> 6;
        const float* src = P + (8 * gg) * AP_ROW + (16 * wv + r) * 9 + (8 - tapd);
        const uint2 lo = pack4_bf16(src[0], src[AP_ROW], src[2 * AP_ROW], src[3 * AP_ROW]);
        const uint2 hi = pack4_bf16(src[4 * AP_ROW], src[5 * AP_ROW], src[6 * AP_ROW], src[7 * AP_ROW]);
        const int lane = r + 16 * (2 * (it.q & 1) + gg);
        wd[(((size_t)(it.ch * ctn + it.ct) * 4 + (2 * it.hf + wv)) * 18 + (tapd * 2 + (it.q >> 1))) * 64 + lane] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
  } else if (it.kind == 1) {
    // ---- any other parameter range (head conv, channel-attention / q-layer tensors, ...): plain Adam ----
    for (int i = tid; i < it.n; i += 256) adam_one(c, a.p, a.g, a.m, a.v, (size_t)it.woff + i);
  } else if (it.kind == 2) {
    // ---- bias of a 64-multiple conv (n = cout <= 4096): Adam + the copy in packed channel order ----
    for (int i = tid; i < it.n; i += 256) P[i] = adam_one(c, a.p, a.g, a.m, a.v, (size_t)it.woff + i);
    __syncthreads();
    for (int i = tid; i < it.n; i += 256) it.b_packed[i] = P[it.shuffle ? 4 * (i & 63) + (i >> 6) : i];
  } else {
    // ---- tail conv (kind 3): weights [C,64,3,3] (n = C*576) and its bias [C] at `boff` ----
    const int C = it.cout;
    for (int i = tid; i < it.n; i += 256) P[i] = adam_one(c, a.p, a.g, a.m, a.v, (size_t)it.woff + i);
    for (int i = tid; i < C; i += 256) adam_one(c, a.p, a.g, a.m, a.v, (size_t)it.boff + i);
    __syncthreads();
    uint16_t* wf = reinterpret_cast<uint16_t*>(it.w_fwd);
    uint16_t* wd = reinterpret_cast<uint16_t*>(it.w_dgrad);
    for (int i = tid; i < 18 * 64 * 8; i += 256) {          // pack_kernel, kind 2
      const int e = i & 7, lane = (i >> 3) & 63, s = i >> 9;
      const int r = lane & 15, g = lane >> 4;
      const int half = s & 1, tap = s >> 1;
      const int ci = 32 * half + 8 * g + e;
      wf[i] = (r < C) ? f32_to_bf16_bits(P[(r * 64 + ci) * 9 + tap]) : (uint16_t)0;
    }
    if (wd) {
      for (int i = tid; i < 4 * 2 * 64 * 8; i += 256) {
        const int e = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) & 1, wave = i >> 10;
        const int r = lane & 15, g = lane >> 4;
        const int k = 32 * ks + 8 * g + e, tap = k >> 2, cc = k & 3;
        wd[i] = (tap < 9 && cc < C) ? f32_to_bf16_bits(P[(cc * 64 + 16 * wave + r) * 9 + (8 - tap)]) : (uint16_t)0;
      }
    }
  }
}

extern "C" int rumpy_adam_pack(const rumpy_adam_pack_args* p, void* stream) {
  if (!p || !p->items || p->nitems <= 0 || !p->p || !p->g || !p->m || !p->v) { rumpy_set_error("rumpy_adam_pack: bad argument"); return RUMPY_E_ARG; }
  if (!p->hyper && !(p->hyper_value.bias_c1 > 0.f && p->hyper_value.sqrt_bias_c2 > 0.f)) { rumpy_set_error("rumpy_adam_pack: neither a hyper pointer nor by-value hyper-parameters"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(adam_pack_kernel, dim3(p->nitems), dim3(256), 0, (hipStream_t)stream, *p);
  return rumpy_check_launch("rumpy_adam_pack");
}

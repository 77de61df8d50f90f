// Channel attention (RCAN CALayer): squeeze-excite MLP on the pooled vector, gate * feature + skip, and backward.
// The global average pool itself is fused into the producing conv's epilogue (per-tile channel sums,
// wave-shuffle reduced over the 16 pixel lanes); everything here is tiny or purely bandwidth-bound.
#include "common.hpp"

constexpr int CA_MAXC = 256;
constexpr int CA_MAXR = 64;

// one workgroup per image; thread c < C
__global__ void ca_mlp_fwd_kernel(rumpy_ca_mlp_fwd_args a) {
  __shared__ float sp[CA_MAXC];
  __shared__ float sh[CA_MAXR];
  const int n = blockIdx.x, c = threadIdx.x;
  if (c < a.C) {
    float s = 0.f;
    for (int t = 0; t < a.ntiles; ++t) s += a.pool[((size_t)n * a.ntiles + t) * a.C + c];
    s *= a.inv_hw;
    sp[c] = s;
    a.mean[(size_t)n * a.C + c] = s;
  }
  __syncthreads();
  if (c < a.Cr) {
    float h = a.b1[c];
    for (int k = 0; k < a.C; ++k) h = fmaf(a.w1[(size_t)c * a.C + k], sp[k], h);
    h = fmaxf(h, 0.f);
    sh[c] = h;
    a.hidden[(size_t)n * a.Cr + c] = h;
  }
  __syncthreads();
  if (c < a.C) {
    float z = a.b2[c];
    for (int r = 0; r < a.Cr; ++r) z = fmaf(a.w2[(size_t)c * a.Cr + r], sh[r], z);
    a.gate[(size_t)n * a.C + c] = 1.f / (1.f + expf(-z));
  }
}

// out = res + t * gate ; 8 channels (16 B) per thread
__global__ void ca_scale_res_kernel(const uint4* __restrict__ t, const uint4* __restrict__ res, const float* __restrict__ gate,
                                    uint4* __restrict__ out, int HW, int C, size_t total_vec) {
  const int cv = C / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cv);
    const size_t n = i / ((size_t)cv * HW);
    const float* gp = gate + n * C + c8 * 8;
    const uint4 tv = t[i];
    float a[4], b[4], ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
    unpack4_bf16(make_uint2(tv.x, tv.y), a);
    unpack4_bf16(make_uint2(tv.z, tv.w), b);
    if (res) {
      const uint4 rv = res[i];
      unpack4_bf16(make_uint2(rv.x, rv.y), ra);
      unpack4_bf16(make_uint2(rv.z, rv.w), rb);
    }
    const uint2 lo = pack4_bf16(fmaf(a[0], gp[0], ra[0]), fmaf(a[1], gp[1], ra[1]), fmaf(a[2], gp[2], ra[2]), fmaf(a[3], gp[3], ra[3]));
    const uint2 hi = pack4_bf16(fmaf(b[0], gp[4], rb[0]), fmaf(b[1], gp[5], rb[1]), fmaf(b[2], gp[6], rb[2]), fmaf(b[3], gp[7], rb[3]));
    out[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
}

// partial[n][chunk][c] = sum over the chunk's <=128 pixels of dy*t.  256 threads: (pixel lane = tid / (C/8), 8 channels)
__global__ void ca_bwd_reduce_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ t, float* __restrict__ partial,
                                     int HW, int C, int nchunks) {
  __shared__ float red[256 * 8];
  const int cv = C / 8;               // vectors per pixel
  const int plane = 256 / cv;         // pixels handled concurrently
  const int n = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
  const int c8 = threadIdx.x % cv, pl = threadIdx.x / cv;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pl < plane) {
    const int p_end = min(HW, (chunk + 1) * 128);
    for (int p = chunk * 128 + pl; p < p_end; p += plane) {
      const size_t i = ((size_t)n * HW + p) * cv + c8;
      const uint4 dv = dy[i], tv = t[i];
      float d0[4], d1[4], t0[4], t1[4];
      unpack4_bf16(make_uint2(dv.x, dv.y), d0); unpack4_bf16(make_uint2(dv.z, dv.w), d1);
      unpack4_bf16(make_uint2(tv.x, tv.y), t0); unpack4_bf16(make_uint2(tv.z, tv.w), t1);
#pragma unroll
      for (int k = 0; k < 4; ++k) { acc[k] = fmaf(d0[k], t0[k], acc[k]); acc[4 + k] = fmaf(d1[k], t1[k], acc[4 + k]); }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = (pl < plane) ? acc[k] : 0.f;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x, v = c / 8, k = c % 8;
    float s = 0.f;
    for (int q = 0; q < plane; ++q) s += red[(q * cv + v) * 8 + k];
    partial[((size_t)n * nchunks + chunk) * C + c] = s;
  }
}

// Backward of the squeeze-excite MLP, two launches (a single workgroup walking over the images one by one - the first
// version - spent 200 us per channel-attention layer in dependent global-load latencies: 70 % of an RCAN training step):
//   image kernel, one workgroup per image n: ds = sum of the pooled-gradient partials, dz = ds * s * (1 - s) (sigmoid'),
//     dh = relu'(hidden) * W2^T dz, dpool = W1^T dh / HW.  dz replaces partial[n][0][:] for the second kernel.
//   parameter kernel, one workgroup: dh for all images into LDS (recomputed: 64-long dot products), then the sums over images
//     of dz x hidden (gW2), dh x mean (gW1), dz (gb2), dh (gb1): 4 groups of 64 threads take every 4th image, their partial sums
//     are added in a fixed order -> bitwise reproducible parameter gradients.
__global__ void __launch_bounds__(CA_MAXC) ca_mlp_bwd_image_kernel(rumpy_ca_mlp_bwd_args a) {
  __shared__ float sdz[CA_MAXC];
  __shared__ float sdh[CA_MAXR];
  const int c = threadIdx.x, n = blockIdx.x;
  float* part = const_cast<float*>(a.partial) + (size_t)n * a.nchunks * a.C;
  if (c < a.C) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= a.nchunks; k += 4) {
      s0 += part[(size_t)k * a.C + c]; s1 += part[(size_t)(k + 1) * a.C + c];
      s2 += part[(size_t)(k + 2) * a.C + c]; s3 += part[(size_t)(k + 3) * a.C + c];
    }
    for (; k < a.nchunks; ++k) s0 += part[(size_t)k * a.C + c];
    const float ds = (s0 + s1) + (s2 + s3);
    const float s = a.gate[(size_t)n * a.C + c];
    const float dz = ds * s * (1.f - s);
    sdz[c] = dz;
    part[c] = dz;                       // this thread was the only reader of column c
  }
  __syncthreads();
  if (c < a.Cr) {
    float dh = 0.f;
    for (int k = 0; k < a.C; ++k) dh = fmaf(a.w2[(size_t)k * a.Cr + c], sdz[k], dh);
    sdh[c] = (a.hidden[(size_t)n * a.Cr + c] > 0.f) ? dh : 0.f;
  }
  __syncthreads();
  if (c < a.C) {
    float dp = 0.f;
    for (int r = 0; r < a.Cr; ++r) dp = fmaf(a.w1[(size_t)r * a.C + c], sdh[r], dp);
    a.dpool[(size_t)n * a.C + c] = dp * a.inv_hw;
  }
}

constexpr int CA_PGROUPS = 4;
constexpr int CA_PR = 16;                               // rows of the squeeze layer kept in registers (Cr <= 16: ca_shape_ok)
// blockDim = CA_PGROUPS * Cp (Cp = C rounded up to 64); dynamic LDS: sdh[N][Cr], red[groups][Cp][2*Cr+1], redb1[groups][Cr]
__device__ __forceinline__ void ca_mlp_bwd_params_body(const rumpy_ca_mlp_bwd_args& a, int Cp, float* dyn) {
  float* sdh = dyn;
  float* red = sdh + a.N * a.Cr;
  const int RS = 2 * a.Cr + 1;
  float* redb1 = red + CA_PGROUPS * Cp * RS;
  const int tid = threadIdx.x, c = tid % Cp, grp = tid / Cp, nthreads = CA_PGROUPS * Cp;
  const float* dzp = a.partial;                        // dz[n][c] = partial[n][0][c]
  const size_t nstride = (size_t)a.nchunks * a.C;
  for (int i = tid; i < a.N * a.Cr; i += nthreads) {
    const int n = i / a.Cr, r = i - n * a.Cr;
    float dh = 0.f;
    for (int k = 0; k < a.C; ++k) dh = fmaf(a.w2[(size_t)k * a.Cr + r], dzp[n * nstride + k], dh);
    sdh[i] = (a.hidden[(size_t)n * a.Cr + r] > 0.f) ? dh : 0.f;
  }
  __syncthreads();
  float gw2[CA_PR], gw1c[CA_PR];
#pragma unroll
  for (int r = 0; r < CA_PR; ++r) { gw2[r] = 0.f; gw1c[r] = 0.f; }
  float gb2 = 0.f;
  if (c < a.C) {
    for (int n = grp; n < a.N; n += CA_PGROUPS) {
      const float dz = dzp[n * nstride + c];
      const float pm = a.mean[(size_t)n * a.C + c];
      gb2 += dz;
#pragma unroll
      for (int r = 0; r < CA_PR; ++r) {
        if (r < a.Cr) {
          gw2[r] = fmaf(dz, a.hidden[(size_t)n * a.Cr + r], gw2[r]);
          gw1c[r] = fmaf(sdh[n * a.Cr + r], pm, gw1c[r]);
        }
      }
    }
  }
  float* mine = red + ((size_t)grp * Cp + c) * RS;
#pragma unroll
  for (int r = 0; r < CA_PR; ++r) {
    if (r < a.Cr) { mine[r] = gw2[r]; mine[a.Cr + r] = gw1c[r]; }
  }
  mine[2 * a.Cr] = gb2;
  if (c < a.Cr) {
    float gb1 = 0.f;
    for (int n = grp; n < a.N; n += CA_PGROUPS) gb1 += sdh[n * a.Cr + c];
    redb1[grp * a.Cr + c] = gb1;
  }
  __syncthreads();
  if (grp == 0 && c < a.C) {
    auto sum4 = [&](int idx) { return (red[((size_t)0 * Cp + c) * RS + idx] + red[((size_t)1 * Cp + c) * RS + idx]) +
                                      (red[((size_t)2 * Cp + c) * RS + idx] + red[((size_t)3 * Cp + c) * RS + idx]); };
    for (int r = 0; r < a.Cr; ++r) {
      a.gw2[(size_t)c * a.Cr + r] = sum4(r) * a.scale;
      a.gw1[(size_t)r * a.C + c] = sum4(a.Cr + r) * a.scale;
    }
    a.gb2[c] = sum4(2 * a.Cr) * a.scale;
    if (c < a.Cr) a.gb1[c] = ((redb1[c] + redb1[a.Cr + c]) + (redb1[2 * a.Cr + c] + redb1[3 * a.Cr + c])) * a.scale;
  }
}

extern __shared__ float ca_dyn_lds[];
__global__ void ca_mlp_bwd_params_kernel(rumpy_ca_mlp_bwd_args a, int Cp) { ca_mlp_bwd_params_body(a, Cp, ca_dyn_lds); }
// all channel-attention layers of a network in ONE launch (one workgroup per layer, arguments from a device table): the
// parameter gradients are off the critical path of the backward pass, 200 launches of ~13 us each were 12 % of an RCAN step
__global__ void ca_mlp_bwd_params_batch_kernel(const rumpy_ca_mlp_bwd_args* __restrict__ items, int Cp) {
  const rumpy_ca_mlp_bwd_args a = items[blockIdx.x];
  ca_mlp_bwd_params_body(a, Cp, ca_dyn_lds);
}

// dt = dy * gate + dpool
__global__ void ca_bwd_apply_kernel(const uint4* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ dpool,
                                    uint4* __restrict__ dt, int HW, int C, size_t total_vec) {
  const int cv = C / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % cv);
    const size_t n = i / ((size_t)cv * HW);
    const float* gp = gate + n * C + c8 * 8;
    const float* dp = dpool + n * C + c8 * 8;
    const uint4 dv = dy[i];
    float a[4], b[4];
    unpack4_bf16(make_uint2(dv.x, dv.y), a);
    unpack4_bf16(make_uint2(dv.z, dv.w), b);
    const uint2 lo = pack4_bf16(fmaf(a[0], gp[0], dp[0]), fmaf(a[1], gp[1], dp[1]), fmaf(a[2], gp[2], dp[2]), fmaf(a[3], gp[3], dp[3]));
    const uint2 hi = pack4_bf16(fmaf(b[0], gp[4], dp[4]), fmaf(b[1], gp[5], dp[5]), fmaf(b[2], gp[6], dp[6]), fmaf(b[3], gp[7], dp[7]));
    dt[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
}

// ---- fused forms used by the engine: the squeeze-excite MLP is ~600 flops per image, so every workgroup of the streaming
// kernel recomputes it for its image (a few KB of L2 reads) instead of waiting for a separate launch; workgroup 0 of each
// image also stores what the backward pass needs.  Same arithmetic, same order as the separate kernels: identical results.
struct CaFwdFused {
  const float* pool; const float* w1; const float* b1; const float* w2; const float* b2;
  float* mean; float* hidden; float* gate;
  const uint4* t; const uint4* res; uint4* out;
  int N, HW, C, Cr, ntiles, per_image; float inv_hw; int image_stride;
  const float* qgate;      // [N,C] meta-attention gate (QRCAB q-layer) multiplied into the channel-attention gate, or NULL
};
constexpr int CA_PRE = 4;       // vectors per thread requested BEFORE the MLP is recomputed: their latency hides behind it
template <int FMT>
__global__ void __launch_bounds__(256) ca_fwd_fused_kernel(CaFwdFused a) {
  __shared__ float sp[CA_MAXC];
  __shared__ float sh[CA_MAXR];
  __shared__ float sg[CA_MAXC];
  const int n = blockIdx.x / a.per_image, j = blockIdx.x - n * a.per_image, c = threadIdx.x;
  const int cv = a.C / 8;
  const size_t img_vec = (size_t)a.HW * cv, base = (size_t)n * img_vec, stride = (size_t)a.per_image * 256;
  const size_t v0 = (size_t)j * 256 + threadIdx.x;
  uint4 T[CA_PRE], R[CA_PRE];
#pragma unroll
  for (int k = 0; k < CA_PRE; ++k) {
    const size_t v = v0 + k * stride;
    const size_t i = base + (v < img_vec ? v : 0);
    T[k] = a.t[i];
    R[k] = a.res ? a.res[i] : make_uint4(0, 0, 0, 0);
  }
  if (c < a.C) {
    float s = 0.f;
    for (int t = 0; t < a.ntiles; ++t) s += a.pool[(size_t)n * a.image_stride + (size_t)t * a.C + c];
    s *= a.inv_hw;
    sp[c] = s;
    if (j == 0) a.mean[(size_t)n * a.C + c] = s;
  }
  __syncthreads();
  if (c < a.Cr) {
    float h = a.b1[c];
    for (int k = 0; k < a.C; ++k) h = fmaf(a.w1[(size_t)c * a.C + k], sp[k], h);
    h = fmaxf(h, 0.f);
    sh[c] = h;
    if (j == 0) a.hidden[(size_t)n * a.Cr + c] = h;
  }
  __syncthreads();
  if (c < a.C) {
    float z = a.b2[c];
    for (int r = 0; r < a.Cr; ++r) z = fmaf(a.w2[(size_t)c * a.Cr + r], sh[r], z);
    const float gt = 1.f / (1.f + expf(-z));
    sg[c] = a.qgate ? gt * a.qgate[(size_t)n * a.C + c] : gt;      // x * gate_ca, then * gate_q (QRCAB order)
    if (j == 0) a.gate[(size_t)n * a.C + c] = gt;
  }
  __syncthreads();
  auto apply = [&](size_t v, uint4 tv, uint4 rv) {
    const float* gp = sg + (int)(v % cv) * 8;
    float x[4], y[4], ra[4], rb[4];
    unpack4<FMT>(make_uint2(tv.x, tv.y), x);
    unpack4<FMT>(make_uint2(tv.z, tv.w), y);
    unpack4<FMT>(make_uint2(rv.x, rv.y), ra);
    unpack4<FMT>(make_uint2(rv.z, rv.w), rb);
    const uint2 lo = pack4<FMT>(fmaf(x[0], gp[0], ra[0]), fmaf(x[1], gp[1], ra[1]), fmaf(x[2], gp[2], ra[2]), fmaf(x[3], gp[3], ra[3]));
    const uint2 hi = pack4<FMT>(fmaf(y[0], gp[4], rb[0]), fmaf(y[1], gp[5], rb[1]), fmaf(y[2], gp[6], rb[2]), fmaf(y[3], gp[7], rb[3]));
    a.out[base + v] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
#pragma unroll
  for (int k = 0; k < CA_PRE; ++k) {
    const size_t v = v0 + k * stride;
    if (v < img_vec) apply(v, T[k], R[k]);
  }
  for (size_t v = v0 + CA_PRE * stride; v < img_vec; v += stride)
    apply(v, a.t[base + v], a.res ? a.res[base + v] : make_uint4(0, 0, 0, 0));
}

// Large images leave thousands of pool partial rows per image (two per 6 x 48 strip); every workgroup of the fused kernel walking
// over all of them made a 2K-image RCAN inference spend 300 us per channel-attention layer.  They are first folded IN PLACE to
// CA_FOLD rows: block (n, j) adds rows j, j + CA_FOLD, .. (four groups of threads, fixed order) and overwrites row j, the only
// row of its residue class that anyone reads afterwards.
constexpr int CA_FOLD = 16;
__global__ void __launch_bounds__(256) ca_pool_fold_kernel(float* __restrict__ pool, int ntiles, int C) {
  __shared__ float part[4][64];
  const int n = blockIdx.y, j = blockIdx.x, grp = threadIdx.x >> 6, cl = threadIdx.x & 63;
  float* img = pool + (size_t)n * ntiles * C;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
      int t = j + grp * CA_FOLD;
      for (; t + 4 * CA_FOLD < ntiles; t += 8 * CA_FOLD) { s0 += img[(size_t)t * C + c]; s1 += img[(size_t)(t + 4 * CA_FOLD) * C + c]; }
      if (t < ntiles) s0 += img[(size_t)t * C + c];
    }
    part[grp][cl] = s0 + s1;
    __syncthreads();
    if (grp == 0 && c < C) img[(size_t)j * C + c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
    __syncthreads();
  }
}

struct CaBwdFused {
  const uint4* dy; const float* partial; const float* hidden; const float* gate; const float* w1; const float* w2;
  float* dz; uint4* dt;
  int N, HW, C, Cr, nchunks, per_image; float inv_hw;
  const float* qgate; float* dzq;      // meta-attention gate [N,C] and the gradient before ITS sigmoid (out), or NULL
};
__global__ void __launch_bounds__(256) ca_bwd_fused_kernel(CaBwdFused a) {
  __shared__ float sdz[CA_MAXC];
  __shared__ float sdh[CA_MAXR];
  __shared__ float sg[CA_MAXC];
  __shared__ float sdp[CA_MAXC];
  const int n = blockIdx.x / a.per_image, j = blockIdx.x - n * a.per_image, c = threadIdx.x;
  const int cv = a.C / 8;
  const size_t img_vec = (size_t)a.HW * cv, base = (size_t)n * img_vec, stride = (size_t)a.per_image * 256;
  const size_t v0 = (size_t)j * 256 + threadIdx.x;
  uint4 D[CA_PRE];
#pragma unroll
  for (int k = 0; k < CA_PRE; ++k) {
    const size_t v = v0 + k * stride;
    D[k] = a.dy[base + (v < img_vec ? v : 0)];
  }
  const float* part = a.partial + (size_t)n * a.nchunks * a.C;
  if (c < a.C) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= a.nchunks; k += 4) {
      s0 += part[(size_t)k * a.C + c]; s1 += part[(size_t)(k + 1) * a.C + c];
      s2 += part[(size_t)(k + 2) * a.C + c]; s3 += part[(size_t)(k + 3) * a.C + c];
    }
    for (; k < a.nchunks; ++k) s0 += part[(size_t)k * a.C + c];
    const float ds = (s0 + s1) + (s2 + s3);                // d(loss)/d(total gate): the block output is t * (gate_ca * gate_q)
    const float s = a.gate[(size_t)n * a.C + c];
    const float gq = a.qgate ? a.qgate[(size_t)n * a.C + c] : 1.f;
    const float dz = (ds * gq) * s * (1.f - s);
    sdz[c] = dz;
    sg[c] = s * gq;
    if (j == 0) {
      a.dz[(size_t)n * a.C + c] = dz;                      // for the parameter-gradient launch (rumpy_ca_mlp_bwd_params, nchunks = 1)
      if (a.dzq) a.dzq[(size_t)n * a.C + c] = (ds * s) * gq * (1.f - gq);
    }
  }
  __syncthreads();
  if (c < a.Cr) {
    float dh = 0.f;
    for (int k = 0; k < a.C; ++k) dh = fmaf(a.w2[(size_t)k * a.Cr + c], sdz[k], dh);
    sdh[c] = (a.hidden[(size_t)n * a.Cr + c] > 0.f) ? dh : 0.f;
  }
  __syncthreads();
  if (c < a.C) {
    float dp = 0.f;
    for (int r = 0; r < a.Cr; ++r) dp = fmaf(a.w1[(size_t)r * a.C + c], sdh[r], dp);
    sdp[c] = dp * a.inv_hw;
  }
  __syncthreads();
  auto apply = [&](size_t v, uint4 dv) {
    const int c8 = (int)(v % cv) * 8;
    const float* gp = sg + c8;
    const float* dp = sdp + c8;
    float x[4], y[4];
    unpack4_bf16(make_uint2(dv.x, dv.y), x);
    unpack4_bf16(make_uint2(dv.z, dv.w), y);
    const uint2 lo = pack4_bf16(fmaf(x[0], gp[0], dp[0]), fmaf(x[1], gp[1], dp[1]), fmaf(x[2], gp[2], dp[2]), fmaf(x[3], gp[3], dp[3]));
    const uint2 hi = pack4_bf16(fmaf(y[0], gp[4], dp[4]), fmaf(y[1], gp[5], dp[5]), fmaf(y[2], gp[6], dp[6]), fmaf(y[3], gp[7], dp[7]));
    a.dt[base + v] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
#pragma unroll
  for (int k = 0; k < CA_PRE; ++k) {
    const size_t v = v0 + k * stride;
    if (v < img_vec) apply(v, D[k]);
  }
  for (size_t v = v0 + CA_PRE * stride; v < img_vec; v += stride) apply(v, a.dy[base + v]);
}

static int stream_blocks(size_t total_vec) {
  size_t b = (total_vec + 255) / 256;
  const size_t cap = (size_t)rumpy_device_cus() * 8;
  return (int)(b > cap ? cap : (b ? b : 1));
}
// (round 6: C / 8 need not divide the 256 threads any more - 192 features: ca_bwd_reduce_kernel leaves the surplus threads idle, every other kernel indexes by vector)
static bool ca_shape_ok(int C, int Cr) { return C > 0 && C <= CA_MAXC && C % 8 == 0 && Cr > 0 && Cr <= 16; }

extern "C" int rumpy_ca_mlp_fwd(const rumpy_ca_mlp_fwd_args* p, void* stream) {
  if (!p || !p->pool || !p->w1 || !p->b1 || !p->w2 || !p->b2 || !p->mean || !p->hidden || !p->gate) { rumpy_set_error("rumpy_ca_mlp_fwd: null pointer"); return RUMPY_E_ARG; }
  if (!ca_shape_ok(p->C, p->Cr) || p->N <= 0 || p->ntiles <= 0) { rumpy_set_error("rumpy_ca_mlp_fwd: unsupported shape C=%d Cr=%d", p->C, p->Cr); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(ca_mlp_fwd_kernel, dim3(p->N), dim3(CA_MAXC), 0, (hipStream_t)stream, *p);
  return rumpy_check_launch("rumpy_ca_mlp_fwd");
}
extern "C" int rumpy_ca_scale_res_fwd(const rumpy_ca_scale_args* p, void* stream) {
  if (!p || !p->t || !p->gate || !p->out || p->N <= 0 || p->HW <= 0 || p->C <= 0 || p->C % 8) { rumpy_set_error("rumpy_ca_scale_res_fwd: bad argument"); return RUMPY_E_ARG; }
  const size_t tv = (size_t)p->N * p->HW * (p->C / 8);
  hipLaunchKernelGGL(ca_scale_res_kernel, dim3(stream_blocks(tv)), dim3(256), 0, (hipStream_t)stream, (const uint4*)p->t,
                     (const uint4*)p->res, p->gate, (uint4*)p->out, p->HW, p->C, tv);
  return rumpy_check_launch("rumpy_ca_scale_res_fwd");
}
extern "C" int rumpy_ca_bwd_reduce(const rumpy_ca_bwd_reduce_args* p, void* stream) {
  if (!p || !p->dy || !p->t || !p->partial || p->N <= 0 || p->HW <= 0 || !ca_shape_ok(p->C, 1)) { rumpy_set_error("rumpy_ca_bwd_reduce: bad argument"); return RUMPY_E_ARG; }
  const int nchunks = (p->HW + 127) / 128;
  hipLaunchKernelGGL(ca_bwd_reduce_kernel, dim3(p->N * nchunks), dim3(256), 0, (hipStream_t)stream, (const uint4*)p->dy,
                     (const uint4*)p->t, p->partial, p->HW, p->C, nchunks);
  return rumpy_check_launch("rumpy_ca_bwd_reduce");
}
constexpr size_t CA_PARAMS_LDS_MAX = 150 * 1024;      // 256 channels x 16 hidden units (RCAN at 256 features, round 5) need 135 KB: opted in below
static size_t ca_params_lds(int N, int C, int Cr, int* Cp) {
  *Cp = (C + 63) / 64 * 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)ca_mlp_bwd_params_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CA_PARAMS_LDS_MAX);
    (void)hipFuncSetAttribute((const void*)ca_mlp_bwd_params_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CA_PARAMS_LDS_MAX);
    attr_set = true;
  }
  return ((size_t)N * Cr + (size_t)CA_PGROUPS * *Cp * (2 * Cr + 1) + (size_t)CA_PGROUPS * Cr) * sizeof(float);
}
extern "C" int rumpy_ca_mlp_bwd(const rumpy_ca_mlp_bwd_args* p, void* stream) {
  if (!p || !p->partial || !p->mean || !p->hidden || !p->gate || !p->w1 || !p->w2 || !p->dpool) {
    rumpy_set_error("rumpy_ca_mlp_bwd: null pointer"); return RUMPY_E_ARG; }
  const bool with_params = p->gw1 || p->gb1 || p->gw2 || p->gb2;
  if (with_params && (!p->gw1 || !p->gb1 || !p->gw2 || !p->gb2)) { rumpy_set_error("rumpy_ca_mlp_bwd: give all four gradient pointers or none"); return RUMPY_E_ARG; }
  if (!ca_shape_ok(p->C, p->Cr) || p->N <= 0 || p->nchunks <= 0) { rumpy_set_error("rumpy_ca_mlp_bwd: unsupported shape"); return RUMPY_E_ARG; }
  int Cp;
  const size_t dyn = ca_params_lds(p->N, p->C, p->Cr, &Cp);
  if (p->Cr > CA_PR || dyn > CA_PARAMS_LDS_MAX) { rumpy_set_error("rumpy_ca_mlp_bwd: shape too large for the parameter kernel (N=%d C=%d Cr=%d)", p->N, p->C, p->Cr); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(ca_mlp_bwd_image_kernel, dim3(p->N), dim3(CA_MAXC), 0, (hipStream_t)stream, *p);
  if (with_params) hipLaunchKernelGGL(ca_mlp_bwd_params_kernel, dim3(1), dim3(CA_PGROUPS * Cp), dyn, (hipStream_t)stream, *p, Cp);
  return rumpy_check_launch("rumpy_ca_mlp_bwd");
}
extern "C" int rumpy_ca_mlp_bwd_params(const rumpy_ca_mlp_bwd_args* items_device, int32_t nitems, int32_t N, int32_t C, int32_t Cr, void* stream) {
  if (!items_device || nitems <= 0 || N <= 0 || !ca_shape_ok(C, Cr)) { rumpy_set_error("rumpy_ca_mlp_bwd_params: bad argument"); return RUMPY_E_ARG; }
  int Cp;
  const size_t dyn = ca_params_lds(N, C, Cr, &Cp);
  if (Cr > CA_PR || dyn > CA_PARAMS_LDS_MAX) { rumpy_set_error("rumpy_ca_mlp_bwd_params: shape too large (N=%d C=%d Cr=%d)", N, C, Cr); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(ca_mlp_bwd_params_batch_kernel, dim3(nitems), dim3(CA_PGROUPS * Cp), dyn, (hipStream_t)stream, items_device, Cp);
  return rumpy_check_launch("rumpy_ca_mlp_bwd_params");
}
extern "C" int rumpy_ca_bwd_apply(const rumpy_ca_bwd_apply_args* p, void* stream) {
  if (!p || !p->dy || !p->gate || !p->dpool || !p->dt || p->N <= 0 || p->HW <= 0 || p->C <= 0 || p->C % 8) { rumpy_set_error("rumpy_ca_bwd_apply: bad argument"); return RUMPY_E_ARG; }
  const size_t tv = (size_t)p->N * p->HW * (p->C / 8);
  hipLaunchKernelGGL(ca_bwd_apply_kernel, dim3(stream_blocks(tv)), dim3(256), 0, (hipStream_t)stream, (const uint4*)p->dy,
                     p->gate, p->dpool, (uint4*)p->dt, p->HW, p->C, tv);
  return rumpy_check_launch("rumpy_ca_bwd_apply");
}

static int ca_per_image(int N, int HW, int C) {        // workgroups per image: ~4 vectors per thread, at most 8 workgroups per CU overall
  const size_t img_vec = (size_t)HW * (C / 8);
  int per = (int)((img_vec + 1023) / 1024);
  const int cap = (rumpy_device_cus() * 8 + N - 1) / N;
  if (per > cap) per = cap;
  return per < 1 ? 1 : per;
}
extern "C" int rumpy_ca_fwd_fused(const rumpy_ca_fwd_fused_args* p, void* stream) {
  if (!p || !p->pool || !p->w1 || !p->b1 || !p->w2 || !p->b2 || !p->mean || !p->hidden || !p->gate || !p->t || !p->out) {
    rumpy_set_error("rumpy_ca_fwd_fused: null pointer"); return RUMPY_E_ARG; }
  if (!ca_shape_ok(p->C, p->Cr) || p->N <= 0 || p->HW <= 0 || p->ntiles <= 0) { rumpy_set_error("rumpy_ca_fwd_fused: unsupported shape"); return RUMPY_E_ARG; }
  CaFwdFused d;
  d.pool = p->pool; d.w1 = p->w1; d.b1 = p->b1; d.w2 = p->w2; d.b2 = p->b2; d.mean = p->mean; d.hidden = p->hidden; d.gate = p->gate;
  d.t = (const uint4*)p->t; d.res = (const uint4*)p->res; d.out = (uint4*)p->out;
  d.N = p->N; d.HW = p->HW; d.C = p->C; d.Cr = p->Cr; d.ntiles = p->ntiles; d.inv_hw = p->inv_hw; d.per_image = ca_per_image(p->N, p->HW, p->C);
  d.image_stride = p->ntiles * p->C;
  d.qgate = p->qgate;
  if (p->ntiles > 2 * CA_FOLD) {      // fold the partial rows in place first (pool is scratch of the producing conv)
    hipLaunchKernelGGL(ca_pool_fold_kernel, dim3(CA_FOLD, p->N), dim3(256), 0, (hipStream_t)stream, const_cast<float*>(p->pool), p->ntiles, p->C);
    d.ntiles = CA_FOLD;
  }
  if (p->fmt == RUMPY_FMT_F16) hipLaunchKernelGGL(ca_fwd_fused_kernel<RUMPY_FMT_F16>, dim3(p->N * d.per_image), dim3(256), 0, (hipStream_t)stream, d);
  else if (p->fmt == RUMPY_FMT_BF16) hipLaunchKernelGGL(ca_fwd_fused_kernel<RUMPY_FMT_BF16>, dim3(p->N * d.per_image), dim3(256), 0, (hipStream_t)stream, d);
  else { rumpy_set_error("rumpy_ca_fwd_fused: bad fmt %d", p->fmt); return RUMPY_E_ARG; }
  return rumpy_check_launch("rumpy_ca_fwd_fused");
}
extern "C" int rumpy_ca_bwd_fused(const rumpy_ca_bwd_fused_args* p, void* stream) {
  if (!p || !p->dy || !p->partial || !p->hidden || !p->gate || !p->w1 || !p->w2 || !p->dz || !p->dt) {
    rumpy_set_error("rumpy_ca_bwd_fused: null pointer"); return RUMPY_E_ARG; }
  if (!ca_shape_ok(p->C, p->Cr) || p->N <= 0 || p->HW <= 0 || p->nchunks <= 0) { rumpy_set_error("rumpy_ca_bwd_fused: unsupported shape"); return RUMPY_E_ARG; }
  CaBwdFused d;
  d.dy = (const uint4*)p->dy; d.partial = p->partial; d.hidden = p->hidden; d.gate = p->gate; d.w1 = p->w1; d.w2 = p->w2;
  d.dz = p->dz; d.dt = (uint4*)p->dt; d.qgate = p->qgate; d.dzq = p->dzq;
  if (p->dzq && !p->qgate) { rumpy_set_error("rumpy_ca_bwd_fused: dzq without qgate"); return RUMPY_E_ARG; }   // a constant gate (style 'modulate') has no dzq
  d.N = p->N; d.HW = p->HW; d.C = p->C; d.Cr = p->Cr; d.nchunks = p->nchunks; d.inv_hw = p->inv_hw; d.per_image = ca_per_image(p->N, p->HW, p->C);
  hipLaunchKernelGGL(ca_bwd_fused_kernel, dim3(p->N * d.per_image), dim3(256), 0, (hipStream_t)stream, d);
  return rumpy_check_launch("rumpy_ca_bwd_fused");
}

// ---- meta-attention (q-layer, rumpy/SISR/models/attention_manipulators/q_layer.py:5-45): gate_q = sigmoid(W2 relu(W1 m + b1) + b2) from
// the per-image metadata vector m [M]; it depends on metadata and weights only, so all layers of a network are evaluated by ONE
// launch before the forward pass and their parameter gradients by ONE launch after the backward pass.
constexpr int Q_MAXN = 64, Q_MAXH = 160, Q_MAXM = 256, Q_MAXC = 64;     // 256 -> 160 -> 64 is the contrastive-embedding q-layer
constexpr int Q_NB = 8;                                                  // images per forward workgroup
// grid (layers, ceil(N / Q_NB)), 256 threads.  Thread h owns hidden unit h for Q_NB images: its W1 row streams through L1 once,
// the metadata rows are broadcast reads from LDS.  Then thread c owns output channel c the same way.
__global__ void __launch_bounds__(256) q_mlp_fwd_kernel(const rumpy_q_mlp_item* __restrict__ items, const float* __restrict__ meta,
                                                        int N, int M, int Hq, int C) {
  __shared__ float sm[Q_NB * Q_MAXM];
  __shared__ float sh[Q_NB * Q_MAXH];
  const rumpy_q_mlp_item it = items[blockIdx.x];
  const int n0 = blockIdx.y * Q_NB, t = threadIdx.x;
  const int nb = min(Q_NB, N - n0);
  for (int i = t; i < Q_NB * M; i += 256) sm[i] = (i < nb * M) ? meta[(size_t)n0 * M + i] : 0.f;
  __syncthreads();
  if (t < Hq) {
    float acc[Q_NB];
    const float b = it.b1[t];
#pragma unroll
    for (int k = 0; k < Q_NB; ++k) acc[k] = b;
    const float* wr = it.w1 + (size_t)t * M;
    for (int m = 0; m < M; ++m) {
      const float w = wr[m];
#pragma unroll
      for (int k = 0; k < Q_NB; ++k) acc[k] = fmaf(w, sm[k * M + m], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < Q_NB; ++k) {
      const float h = fmaxf(acc[k], 0.f);
      sh[k * Hq + t] = h;
      if (k < nb) it.hidden[(size_t)(n0 + k) * Hq + t] = h;
    }
  }
  __syncthreads();
  if (t < C) {
    float acc[Q_NB];
    const float b = it.b2[t];
#pragma unroll
    for (int k = 0; k < Q_NB; ++k) acc[k] = b;
    const float* wr = it.w2 + (size_t)t * Hq;
    for (int h = 0; h < Hq; ++h) {
      const float w = wr[h];
#pragma unroll
      for (int k = 0; k < Q_NB; ++k) acc[k] = fmaf(w, sh[k * Hq + h], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < Q_NB; ++k)
      if (k < nb) it.gate[(size_t)(n0 + k) * C + t] = 1.f / (1.f + expf(-acc[k]));
  }
}

// grid (layers, Q_PARTS), 256 threads.  Every workgroup rebuilds dh = relu'(hidden) * (W2^T dz) for its layer in LDS (cheap), then
// writes its 1/Q_PARTS slice of the gradient entries; sums over images run in a fixed order (deterministic).
constexpr int Q_PARTS = 8;
__global__ void __launch_bounds__(256) q_mlp_bwd_params_kernel(const rumpy_q_mlp_item* __restrict__ items, const float* __restrict__ meta,
                                                               int N, int M, int Hq, int C) {
  __shared__ float sdz[Q_MAXN * Q_MAXC];
  __shared__ float sdh[Q_MAXN * Q_MAXH];
  const rumpy_q_mlp_item it = items[blockIdx.x];
  const int tid = threadIdx.x, part = blockIdx.y;
  for (int i = tid; i < N * C; i += 256) sdz[i] = it.dzq[i];
  __syncthreads();
  for (int i = tid; i < N * Hq; i += 256) {           // dh[n][h] = relu'(hidden) * sum_c W2[c][h] dz[n][c]
    const int n = i / Hq, h = i - n * Hq;
    float d = 0.f;
    for (int c = 0; c < C; ++c) d = fmaf(it.w2[(size_t)c * Hq + h], sdz[n * C + c], d);
    sdh[i] = (it.hidden[i] > 0.f) ? d : 0.f;
  }
  __syncthreads();
  const int gt = part * 256 + tid, gstride = Q_PARTS * 256;
  for (int i = gt; i < C * Hq; i += gstride) {         // gW2[c][h] = sum_n dz[n][c] hidden[n][h]
    const int c = i / Hq, h = i - c * Hq;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(sdz[n * C + c], it.hidden[(size_t)n * Hq + h], s);
    it.gw2[i] = s * it.scale;
  }
  for (int i = gt; i < Hq * M; i += gstride) {         // gW1[h][m] = sum_n dh[n][h] meta[n][m]
    const int h = i / M, m = i - h * M;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(sdh[n * Hq + h], meta[(size_t)n * M + m], s);
    it.gw1[i] = s * it.scale;
  }
  for (int c = gt; c < C; c += gstride) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += sdz[n * C + c];
    it.gb2[c] = s * it.scale;
  }
  for (int h = gt; h < Hq; h += gstride) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += sdh[n * Hq + h];
    it.gb1[h] = s * it.scale;
  }
}

// d loss / d metadata of all q-layers of a network: dmeta[n][m] = scale * sum_layers sum_h dh_l[n][h] * W1_l[h][m], dh as above.  grid (N),
// 256 threads; the layers are summed in table order (deterministic).  The metadata of the blind pipeline is the degradation encoder's
// embedding: this is the gradient that reaches the encoder from the SR loss when its trunk is trained jointly.
__global__ void __launch_bounds__(256) q_mlp_bwd_meta_kernel(const rumpy_q_mlp_item* __restrict__ items, int nitems, int N, int M, int Hq, int C,
                                                             float* __restrict__ dmeta) {
  __shared__ float sdz[Q_MAXC];
  __shared__ float sdh[Q_MAXH];
  const int n = blockIdx.x, tid = threadIdx.x;
  float acc = 0.f;                                     // M <= 256: one output per thread
  for (int l = 0; l < nitems; ++l) {
    const rumpy_q_mlp_item it = items[l];
    for (int c = tid; c < C; c += 256) sdz[c] = it.dzq[(size_t)n * C + c];
    __syncthreads();
    for (int h = tid; h < Hq; h += 256) {
      float d = 0.f;
      for (int c = 0; c < C; ++c) d = fmaf(it.w2[(size_t)c * Hq + h], sdz[c], d);
      sdh[h] = (it.hidden[(size_t)n * Hq + h] > 0.f) ? d : 0.f;
    }
    __syncthreads();
    if (tid < M) {
      float s = 0.f;
      for (int h = 0; h < Hq; ++h) s = fmaf(sdh[h], it.w1[(size_t)h * M + tid], s);
      acc += s * it.scale;
    }
    __syncthreads();
  }
  if (tid < M) dmeta[(size_t)n * M + tid] = acc;
}

// ---- the same q-layer with any number of FC layers (ParaCALayer's num_layers, q_layer.py:22-41: 1 .. RUMPY_QN_MAX_LAYERS; the two-layer kernels above
// are what the reference's default builds and stay the hot path).  Layer l: [n[l+1], n[l]] weights, ReLU behind every layer but the last, sigmoid
// behind the last.  Same launch structure and the same accumulation order per output (bias first, inputs ascending): with two layers the gates are
// bitwise those of q_mlp_fwd_kernel. ----
constexpr int QN_MAXW = 256;                     // widest layer (input included)
constexpr int QN_MAXSUM = 448;                   // sum of the layer outputs n[1 .. L]: the backward kernel keeps every layer's delta for 64 images in LDS
__global__ void __launch_bounds__(256) q_mlpn_fwd_kernel(const rumpy_q_mlpn_item* __restrict__ items, const float* __restrict__ meta, int N) {
  __shared__ float buf[2][Q_NB * QN_MAXW];
  const rumpy_q_mlpn_item it = items[blockIdx.x];
  const int n0 = blockIdx.y * Q_NB, t = threadIdx.x, M = it.n[0];
  const int nb = min(Q_NB, N - n0);
  for (int i = t; i < Q_NB * M; i += 256) buf[0][i] = (i < nb * M) ? meta[(size_t)n0 * M + i] : 0.f;
  __syncthreads();
  int hoff = 0, hsum = 0;
  for (int l = 0; l + 1 < it.nlayers; ++l) hsum += it.n[l + 1];
  for (int l = 0; l < it.nlayers; ++l) {
    const int ni = it.n[l], no = it.n[l + 1];
    const float* in = buf[l & 1];
    float* out = buf[(l + 1) & 1];
    const bool last = l + 1 == it.nlayers;
    if (t < no) {
      float acc[Q_NB];
      const float b = it.b[l][t];
#pragma unroll
      for (int k = 0; k < Q_NB; ++k) acc[k] = b;
      const float* wr = it.w[l] + (size_t)t * ni;
      for (int m = 0; m < ni; ++m) {
        const float w = wr[m];
#pragma unroll
        for (int k = 0; k < Q_NB; ++k) acc[k] = fmaf(w, in[k * ni + m], acc[k]);
      }
#pragma unroll
      for (int k = 0; k < Q_NB; ++k) {
        if (last) {
          if (k < nb) it.gate[(size_t)(n0 + k) * no + t] = 1.f / (1.f + expf(-acc[k]));
        } else {
          const float h = fmaxf(acc[k], 0.f);
          out[k * no + t] = h;
          if (k < nb) it.acts[(size_t)(n0 + k) * hsum + hoff + t] = h;
        }
      }
    }
    hoff += no;
    __syncthreads();
  }
}

// grid (layers of the network, Q_PARTS): every workgroup rebuilds the deltas of its q-layer for all images in LDS, then writes its share of the
// gradient entries; sums over images in image order
__global__ void __launch_bounds__(256) q_mlpn_bwd_params_kernel(const rumpy_q_mlpn_item* __restrict__ items, const float* __restrict__ meta, int N) {
  extern __shared__ float sdl[];                 // [N][dsum]: image n's deltas of layer l at n * dsum + doff[l]
  const rumpy_q_mlpn_item it = items[blockIdx.x];
  const int tid = threadIdx.x, part = blockIdx.y, L = it.nlayers, C = it.n[L];
  int doff[RUMPY_QN_MAX_LAYERS + 1], dsum = 0, hsum = 0;
  for (int l = 0; l < L; ++l) { doff[l] = dsum; dsum += it.n[l + 1]; }
  for (int l = 0; l + 1 < L; ++l) hsum += it.n[l + 1];
  for (int i = tid; i < N * C; i += 256) { const int n = i / C, c = i - n * C; sdl[n * dsum + doff[L - 1] + c] = it.dzq[i]; }
  __syncthreads();
  for (int l = L - 2; l >= 0; --l) {             // delta of layer l's output = relu'(act) * W_{l+1}^T delta_{l+1}
    const int nh = it.n[l + 1], no = it.n[l + 2];
    for (int i = tid; i < N * nh; i += 256) {
      const int n = i / nh, h = i - n * nh;
      float d = 0.f;
      for (int o = 0; o < no; ++o) d = fmaf(it.w[l + 1][(size_t)o * nh + h], sdl[n * dsum + doff[l + 1] + o], d);
      sdl[n * dsum + doff[l] + h] = (it.acts[(size_t)n * hsum + doff[l] + h] > 0.f) ? d : 0.f;
    }
    __syncthreads();
  }
  const int gt = part * 256 + tid, gstride = Q_PARTS * 256;
  for (int l = L - 1; l >= 0; --l) {
    const int ni = it.n[l], no = it.n[l + 1];
    for (int i = gt; i < no * ni; i += gstride) {  // gW_l[o][m] = sum_n delta_l[n][o] in_l[n][m]
      const int o = i / ni, m = i - o * ni;
      float s = 0.f;
      for (int n = 0; n < N; ++n) {
        const float x = (l == 0) ? meta[(size_t)n * ni + m] : it.acts[(size_t)n * hsum + doff[l - 1] + m];
        s = fmaf(sdl[n * dsum + doff[l] + o], x, s);
      }
      it.gw[l][i] = s * it.scale;
    }
    for (int o = gt; o < no; o += gstride) {
      float s = 0.f;
      for (int n = 0; n < N; ++n) s += sdl[n * dsum + doff[l] + o];
      it.gb[l][o] = s * it.scale;
    }
  }
}

// d loss / d metadata: grid (N), 256 threads, the q-layers of the network in table order
__global__ void __launch_bounds__(256) q_mlpn_bwd_meta_kernel(const rumpy_q_mlpn_item* __restrict__ items, int nitems, int N, float* __restrict__ dmeta) {
  __shared__ float sd[2][QN_MAXW];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int M = items[0].n[0];
  float acc = 0.f;                                 // M <= 256: one output per thread
  for (int q = 0; q < nitems; ++q) {
    const rumpy_q_mlpn_item it = items[q];
    const int L = it.nlayers, C = it.n[L];
    int hoff[RUMPY_QN_MAX_LAYERS + 1], hsum = 0;
    for (int l = 0; l + 1 < L; ++l) { hoff[l] = hsum; hsum += it.n[l + 1]; }
    for (int c = tid; c < C; c += 256) sd[(L - 1) & 1][c] = it.dzq[(size_t)n * C + c];
    __syncthreads();
    for (int l = L - 2; l >= 0; --l) {
      const int nh = it.n[l + 1], no = it.n[l + 2];
      for (int h = tid; h < nh; h += 256) {
        float d = 0.f;
        for (int o = 0; o < no; ++o) d = fmaf(it.w[l + 1][(size_t)o * nh + h], sd[(l + 1) & 1][o], d);
        sd[l & 1][h] = (it.acts[(size_t)n * hsum + hoff[l] + h] > 0.f) ? d : 0.f;
      }
      __syncthreads();
    }
    if (tid < M) {
      float s = 0.f;
      const int n1 = it.n[1];
      for (int h = 0; h < n1; ++h) s = fmaf(sd[0][h], it.w[0][(size_t)h * M + tid], s);
      acc += s * it.scale;
    }
    __syncthreads();
  }
  if (tid < M) dmeta[(size_t)n * M + tid] = acc;
}

// host-side shape check of the q-layers of a network (the items live in device memory: the caller states the shape they share)
static bool qn_shape_ok(int N, const int32_t* n, int L) {
  if (N <= 0 || N > Q_MAXN || !n || L < 1 || L > RUMPY_QN_MAX_LAYERS) return false;
  int sum = 0;
  for (int l = 0; l <= L; ++l) { if (n[l] <= 0 || n[l] > QN_MAXW) return false; if (l) sum += n[l]; }
  return sum <= QN_MAXSUM;
}
extern "C" int rumpy_q_mlpn_fwd(const rumpy_q_mlpn_item* items_device, int32_t nitems, const float* meta, int32_t N, const int32_t* n, int32_t nlayers, void* stream) {
  if (!items_device || !meta || nitems <= 0 || !qn_shape_ok(N, n, nlayers)) { rumpy_set_error("rumpy_q_mlpn_fwd: bad argument (N=%d layers=%d)", N, nlayers); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(q_mlpn_fwd_kernel, dim3(nitems, (N + Q_NB - 1) / Q_NB), dim3(256), 0, (hipStream_t)stream, items_device, meta, N);
  return rumpy_check_launch("rumpy_q_mlpn_fwd");
}
extern "C" int rumpy_q_mlpn_bwd_params(const rumpy_q_mlpn_item* items_device, int32_t nitems, const float* meta, int32_t N, const int32_t* n, int32_t nlayers, void* stream) {
  if (!items_device || !meta || nitems <= 0 || !qn_shape_ok(N, n, nlayers)) { rumpy_set_error("rumpy_q_mlpn_bwd_params: bad argument (N=%d layers=%d)", N, nlayers); return RUMPY_E_ARG; }
  int dsum = 0;
  for (int l = 1; l <= nlayers; ++l) dsum += n[l];
  const size_t lds = (size_t)N * dsum * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)q_mlpn_bwd_params_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Q_MAXN * QN_MAXSUM * (int)sizeof(float)); attr_set = true; }
  hipLaunchKernelGGL(q_mlpn_bwd_params_kernel, dim3(nitems, Q_PARTS), dim3(256), lds, (hipStream_t)stream, items_device, meta, N);
  return rumpy_check_launch("rumpy_q_mlpn_bwd_params");
}
extern "C" int rumpy_q_mlpn_bwd_meta(const rumpy_q_mlpn_item* items_device, int32_t nitems, int32_t N, const int32_t* n, int32_t nlayers, float* dmeta, void* stream) {
  if (!items_device || !dmeta || nitems <= 0 || !qn_shape_ok(N, n, nlayers)) { rumpy_set_error("rumpy_q_mlpn_bwd_meta: bad argument (N=%d layers=%d)", N, nlayers); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(q_mlpn_bwd_meta_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, items_device, nitems, N, dmeta);
  return rumpy_check_launch("rumpy_q_mlpn_bwd_meta");
}

static bool q_shape_ok(int N, int M, int Hq, int C) { return N > 0 && N <= Q_MAXN && M > 0 && M <= Q_MAXM && Hq > 0 && Hq <= Q_MAXH && C > 0 && C <= Q_MAXC; }
extern "C" int rumpy_q_mlp_fwd(const rumpy_q_mlp_item* items_device, int32_t nitems, const float* meta, int32_t N, int32_t M, int32_t Hq, int32_t C, void* stream) {
  if (!items_device || !meta || nitems <= 0 || !q_shape_ok(N, M, Hq, C)) { rumpy_set_error("rumpy_q_mlp_fwd: bad argument (N=%d M=%d Hq=%d C=%d)", N, M, Hq, C); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(q_mlp_fwd_kernel, dim3(nitems, (N + Q_NB - 1) / Q_NB), dim3(256), 0, (hipStream_t)stream, items_device, meta, N, M, Hq, C);
  return rumpy_check_launch("rumpy_q_mlp_fwd");
}
extern "C" int rumpy_q_mlp_bwd_meta(const rumpy_q_mlp_item* items_device, int32_t nitems, int32_t N, int32_t M, int32_t Hq, int32_t C, float* dmeta, void* stream) {
  if (!items_device || !dmeta || nitems <= 0 || !q_shape_ok(N, M, Hq, C)) { rumpy_set_error("rumpy_q_mlp_bwd_meta: bad argument (N=%d M=%d Hq=%d C=%d)", N, M, Hq, C); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(q_mlp_bwd_meta_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, items_device, nitems, N, M, Hq, C, dmeta);
  return rumpy_check_launch("rumpy_q_mlp_bwd_meta");
}
extern "C" int rumpy_q_mlp_bwd_params(const rumpy_q_mlp_item* items_device, int32_t nitems, const float* meta, int32_t N, int32_t M, int32_t Hq, int32_t C, void* stream) {
  if (!items_device || !meta || nitems <= 0 || !q_shape_ok(N, M, Hq, C)) { rumpy_set_error("rumpy_q_mlp_bwd_params: bad argument (N=%d M=%d Hq=%d C=%d)", N, M, Hq, C); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(q_mlp_bwd_params_kernel, dim3(nitems, Q_PARTS), dim3(256), 0, (hipStream_t)stream, items_device, meta, N, M, Hq, C);
  return rumpy_check_launch("rumpy_q_mlp_bwd_params");
}

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

// single workgroup: fixed-order sums over images -> bitwise reproducible parameter gradients
__global__ void ca_mlp_bwd_kernel(rumpy_ca_mlp_bwd_args a) {
  __shared__ float sdz[CA_MAXC];
  __shared__ float sdh[CA_MAXR];
  const int c = threadIdx.x;
  float gw2[CA_MAXR > 16 ? 16 : CA_MAXR];  // Cr <= 16 rows kept in registers per thread c
  float gb2 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) gw2[r] = 0.f;
  float gw1c[16];  // thread c accumulates gW1[r][c] for r < Cr
#pragma unroll
  for (int r = 0; r < 16; ++r) gw1c[r] = 0.f;
  float gb1 = 0.f;  // thread r < Cr
  for (int n = 0; n < a.N; ++n) {
    float dz = 0.f;
    if (c < a.C) {
      float ds = 0.f;
      for (int k = 0; k < a.nchunks; ++k) ds += a.partial[((size_t)n * a.nchunks + k) * a.C + c];
      const float s = a.gate[(size_t)n * a.C + c];
      dz = ds * s * (1.f - s);
      sdz[c] = dz;
      gb2 += dz;
    }
    __syncthreads();
    if (c < a.Cr) {
      float dh = 0.f;
      for (int k = 0; k < a.C; ++k) dh = fmaf(a.w2[(size_t)k * a.Cr + c], sdz[k], dh);
      dh = (a.hidden[(size_t)n * a.Cr + c] > 0.f) ? dh : 0.f;
      sdh[c] = dh;
      gb1 += dh;
    }
    __syncthreads();
    if (c < a.C) {
      float dp = 0.f;
      const float pm = a.mean[(size_t)n * a.C + c];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < a.Cr) {
          dp = fmaf(a.w1[(size_t)r * a.C + c], sdh[r], dp);
          gw2[r] = fmaf(dz, a.hidden[(size_t)n * a.Cr + r], gw2[r]);
          gw1c[r] = fmaf(sdh[r], pm, gw1c[r]);
        }
      }
      a.dpool[(size_t)n * a.C + c] = dp * a.inv_hw;
    }
    __syncthreads();
  }
  if (c < a.C) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < a.Cr) {
        a.gw2[(size_t)c * a.Cr + r] = gw2[r] * a.scale;
        a.gw1[(size_t)r * a.C + c] = gw1c[r] * a.scale;
      }
    }
    a.gb2[c] = gb2 * a.scale;
  }
  if (c < a.Cr) a.gb1[c] = gb1 * a.scale;
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

static int stream_blocks(size_t total_vec) {
  size_t b = (total_vec + 255) / 256;
  const size_t cap = (size_t)rumpy_device_cus() * 8;
  return (int)(b > cap ? cap : (b ? b : 1));
}
static bool ca_shape_ok(int C, int Cr) { return C > 0 && C <= CA_MAXC && C % 8 == 0 && 256 % (C / 8) == 0 && Cr > 0 && Cr <= 16; }

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
extern "C" int rumpy_ca_mlp_bwd(const rumpy_ca_mlp_bwd_args* p, void* stream) {
  if (!p || !p->partial || !p->mean || !p->hidden || !p->gate || !p->w1 || !p->w2 || !p->dpool || !p->gw1 || !p->gb1 || !p->gw2 || !p->gb2) {
    rumpy_set_error("rumpy_ca_mlp_bwd: null pointer"); return RUMPY_E_ARG; }
  if (!ca_shape_ok(p->C, p->Cr) || p->N <= 0 || p->nchunks <= 0) { rumpy_set_error("rumpy_ca_mlp_bwd: unsupported shape"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(ca_mlp_bwd_kernel, dim3(1), dim3(CA_MAXC), 0, (hipStream_t)stream, *p);
  return rumpy_check_launch("rumpy_ca_mlp_bwd");
}
extern "C" int rumpy_ca_bwd_apply(const rumpy_ca_bwd_apply_args* p, void* stream) {
  if (!p || !p->dy || !p->gate || !p->dpool || !p->dt || p->N <= 0 || p->HW <= 0 || p->C <= 0 || p->C % 8) { rumpy_set_error("rumpy_ca_bwd_apply: bad argument"); return RUMPY_E_ARG; }
  const size_t tv = (size_t)p->N * p->HW * (p->C / 8);
  hipLaunchKernelGGL(ca_bwd_apply_kernel, dim3(stream_blocks(tv)), dim3(256), 0, (hipStream_t)stream, (const uint4*)p->dy,
                     p->gate, p->dpool, (uint4*)p->dt, p->HW, p->C, tv);
  return rumpy_check_launch("rumpy_ca_bwd_apply");
}

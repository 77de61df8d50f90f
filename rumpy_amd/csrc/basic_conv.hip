// Direct fp32 convolution for the reference's "basic" models (SRCNN 9-5-5 / VDSR 20 x 3x3 on single-channel Y images,
// rumpy/SISR/models/basic/architectures.py:6-77): forward, data gradient (same kernel, filter read transposed + flipped),
// weight / bias gradient, MSE loss.  fp32 NCHW, the reference's OIHW filters in place.  These layers are K = 81 -> 64 channels,
// 64 -> 32, 32 -> 1: not MFMA-shaped; the kernels are LDS-tiled VALU code (16x16 output pixels per workgroup, the input halo
// tile and the filter slice of one input channel staged per step, filter taps fetched as wave-uniform 16-byte LDS broadcasts).
#include "common.hpp"

namespace {

constexpr int DT = 16;                 // output tile edge
constexpr int DK_MAX = 11;             // largest kernel size
constexpr int DHALO_MAX = DT + DK_MAX - 1;

struct DconvDev {
  const float* x; const float* w; const float* bias; const float* mask; const float* res; float* y;
  int N, Cin, Cout, H, W, k, relu, transposed, tiles_x, tiles_y;
};

// OB output channels per workgroup (16 / 4 / 1)
template <int OB>
__global__ void __launch_bounds__(256) dconv_kernel(DconvDev a) {
  __shared__ float sx[DHALO_MAX * DHALO_MAX];
  __shared__ __attribute__((aligned(16))) float sw[DK_MAX * DK_MAX * OB];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const TileCoord t = decode_tile(blockIdx.x, a.tiles_x, a.tiles_y);
  const int o0 = blockIdx.y * OB, k = a.k, p = k >> 1, hw = DT + k - 1, kk = k * k;
  const int y0 = t.ty * DT - p, x0 = t.tx * DT - p;
  float acc[OB];
#pragma unroll
  for (int o = 0; o < OB; ++o) acc[o] = 0.f;
  for (int i = 0; i < a.Cin; ++i) {
    const float* xi = a.x + ((size_t)t.n * a.Cin + i) * a.H * a.W;
    for (int e = tid; e < hw * hw; e += 256) {
      const int r = e / hw, c = e - r * hw, yy = y0 + r, xx = x0 + c;
      sx[e] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? xi[(size_t)yy * a.W + xx] : 0.f;
    }
    for (int e = tid; e < kk * OB; e += 256) {
      const int tap = e / OB, o = e - tap * OB, oc = o0 + o;
      float v = 0.f;
      if (oc < a.Cout) {
        if (!a.transposed) v = a.w[((size_t)oc * a.Cin + i) * kk + tap];
        else v = a.w[((size_t)i * a.Cout + oc) * kk + (kk - 1 - tap)];      // w_ref[i][oc][k-1-ky][k-1-kx]
      }
      sw[e] = v;
    }
    __syncthreads();
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        const float v = sx[(ty + ky) * hw + tx + kx];
        const float* wt = sw + (ky * k + kx) * OB;
        if (OB >= 4) {
#pragma unroll
          for (int o = 0; o < OB; o += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wt + o);
            acc[o] = fmaf(w4.x, v, acc[o]); acc[o + 1] = fmaf(w4.y, v, acc[o + 1]);
            acc[o + 2] = fmaf(w4.z, v, acc[o + 2]); acc[o + 3] = fmaf(w4.w, v, acc[o + 3]);
          }
        } else {
          acc[0] = fmaf(wt[0], v, acc[0]);
        }
      }
    __syncthreads();
  }
  const int yy = t.ty * DT + ty, xx = t.tx * DT + tx;
  if (yy < a.H && xx < a.W) {
#pragma unroll
    for (int o = 0; o < OB; ++o) {
      const int oc = o0 + o;
      if (oc < a.Cout) {
        const size_t idx = (((size_t)t.n * a.Cout + oc) * a.H + yy) * a.W + xx;
        float v = acc[o] + (a.bias ? a.bias[oc] : 0.f);
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.res) v += a.res[idx];
        if (a.mask && !(a.mask[idx] > 0.f)) v = 0.f;
        a.y[idx] = v;
      }
    }
  }
}

struct DwgradDev {
  const float* x; const float* dy; float* partial;
  int N, Cin, Cout, H, W, k, S, OG, tiles_x, tiles_y, row;      // row = floats per slab
};

// grid (Cin * ceil(Cout / OG), S): thread = (output channel of the group, filter tap); pixels of a tile are the reduction loop
__global__ void __launch_bounds__(256) dconv_wgrad_kernel(DwgradDev a) {
  __shared__ float sx[DHALO_MAX * DHALO_MAX];
  __shared__ float sdy[16 * DT * DT];
  const int tid = threadIdx.x, k = a.k, p = k >> 1, hw = DT + k - 1, kk = k * k;
  const int groups = (a.Cout + a.OG - 1) / a.OG;
  const int i = blockIdx.x / groups, o0 = (blockIdx.x - i * groups) * a.OG;
  const int ol = tid / kk, tap = tid - ol * kk, ky = tap / k, kx = tap - ky * k;
  const bool active = ol < a.OG && (o0 + ol) < a.Cout;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  float acc = 0.f;
  for (int tile = blockIdx.y; tile < ntiles; tile += a.S) {
    const TileCoord t = decode_tile(tile, a.tiles_x, a.tiles_y);
    const int y0 = t.ty * DT, x0 = t.tx * DT;
    const float* xi = a.x + ((size_t)t.n * a.Cin + i) * a.H * a.W;
    for (int e = tid; e < hw * hw; e += 256) {
      const int r = e / hw, c = e - r * hw, yy = y0 - p + r, xx = x0 - p + c;
      sx[e] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? xi[(size_t)yy * a.W + xx] : 0.f;
    }
    for (int o = 0; o < a.OG; ++o) {
      const int oc = o0 + o, yy = y0 + (tid >> 4), xx = x0 + (tid & 15);
      sdy[o * 256 + tid] = (oc < a.Cout && yy < a.H && xx < a.W) ? a.dy[(((size_t)t.n * a.Cout + oc) * a.H + yy) * a.W + xx] : 0.f;
    }
    __syncthreads();
    if (active) {
      const float* d = sdy + ol * 256;
      const float* s = sx + ky * hw + kx;
      for (int r = 0; r < DT; ++r)
#pragma unroll
        for (int c = 0; c < DT; ++c) acc = fmaf(d[r * DT + c], s[r * hw + c], acc);
    }
    __syncthreads();
  }
  if (active) a.partial[(size_t)blockIdx.y * a.row + ((size_t)(o0 + ol) * a.Cin + i) * kk + tap] = acc;
}

__global__ void dconv_wgrad_reduce_kernel(const float* partial, int S, int row, int n, float scale, float* gw) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float s = 0.f;
  for (int q = 0; q < S; ++q) s += partial[(size_t)q * row + j];
  gw[j] = s * scale;
}

// one workgroup per output channel: gb[o] = scale * sum over images and pixels, fixed order
__global__ void __launch_bounds__(256) dconv_bgrad_kernel(const float* dy, int N, int Cout, int HW, float scale, float* gb) {
  __shared__ float red[256];
  const int o = blockIdx.x;
  float s = 0.f;
  for (int n = 0; n < N; ++n) {
    const float* d = dy + ((size_t)n * Cout + o) * HW;
    for (int e = threadIdx.x; e < HW; e += 256) s += d[e];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) gb[o] = red[0] * scale;
}

__global__ void __launch_bounds__(256) mse_partial_kernel(const float* out, const float* target, float* grad, int64_t n, float gscale, float* partial) {
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float d = out[e] - target[e];
    s = fmaf(d, d, s);
    if (grad) grad[e] = d * gscale;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void mse_finalize_kernel(const float* partial, int n, float inv_numel, float* loss) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] * inv_numel;
}

inline int cdiv_i(int a, int b) { return (a + b - 1) / b; }

inline bool dconv_shape_ok(int N, int Cin, int Cout, int H, int W, int k) {
  return N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && k >= 1 && k <= DK_MAX && (k & 1);
}

// slabs and channel grouping of the weight gradient (shared by the launcher and the scratch-size query)
inline void dwgrad_split(int N, int Cin, int Cout, int H, int W, int k, int* S, int* OG) {
  int og = 256 / (k * k);
  if (og > 16) og = 16;
  if (og > Cout) og = Cout;
  const int ntiles = N * cdiv_i(H, DT) * cdiv_i(W, DT);
  const int blocks = Cin * cdiv_i(Cout, og);
  int s = cdiv_i(2048, blocks);            // enough workgroups for 256 CUs
  if (s > ntiles) s = ntiles;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  *S = s; *OG = og;
}

}  // namespace

extern "C" int rumpy_dconv(const rumpy_dconv_args* p, void* stream) {
  if (!p || !p->x || !p->w || !p->y || !dconv_shape_ok(p->N, p->Cin, p->Cout, p->H, p->W, p->k)) {
    rumpy_set_error("rumpy_dconv: bad argument (odd kernel sizes 1..%d only)", DK_MAX);
    return RUMPY_E_ARG;
  }
  DconvDev d;
  d.x = p->x; d.w = p->w; d.bias = p->bias; d.mask = p->mask; d.res = p->res; d.y = p->y;
  d.N = p->N; d.Cin = p->Cin; d.Cout = p->Cout; d.H = p->H; d.W = p->W; d.k = p->k; d.relu = p->relu; d.transposed = p->transposed;
  d.tiles_x = cdiv_i(p->W, DT); d.tiles_y = cdiv_i(p->H, DT);
  const int tiles = d.N * d.tiles_x * d.tiles_y;
  hipStream_t s = (hipStream_t)stream;
  if (p->Cout >= 16) hipLaunchKernelGGL(dconv_kernel<16>, dim3(tiles, cdiv_i(p->Cout, 16)), dim3(256), 0, s, d);
  else if (p->Cout >= 4) hipLaunchKernelGGL(dconv_kernel<4>, dim3(tiles, cdiv_i(p->Cout, 4)), dim3(256), 0, s, d);
  else hipLaunchKernelGGL(dconv_kernel<1>, dim3(tiles, p->Cout), dim3(256), 0, s, d);
  return rumpy_check_launch("rumpy_dconv");
}

extern "C" int64_t rumpy_dconv_wgrad_partial_floats(int32_t N, int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t k) {
  if (!dconv_shape_ok(N, Cin, Cout, H, W, k)) return 0;
  int S, OG;
  dwgrad_split(N, Cin, Cout, H, W, k, &S, &OG);
  return (int64_t)S * Cout * Cin * k * k;
}

extern "C" int rumpy_dconv_wgrad(const rumpy_dconv_wgrad_args* p, void* stream) {
  if (!p || !p->x || !p->dy || !p->partial || !p->gw || !dconv_shape_ok(p->N, p->Cin, p->Cout, p->H, p->W, p->k)) {
    rumpy_set_error("rumpy_dconv_wgrad: bad argument");
    return RUMPY_E_ARG;
  }
  DwgradDev d;
  d.x = p->x; d.dy = p->dy; d.partial = p->partial;
  d.N = p->N; d.Cin = p->Cin; d.Cout = p->Cout; d.H = p->H; d.W = p->W; d.k = p->k;
  dwgrad_split(p->N, p->Cin, p->Cout, p->H, p->W, p->k, &d.S, &d.OG);
  d.tiles_x = cdiv_i(p->W, DT); d.tiles_y = cdiv_i(p->H, DT);
  d.row = p->Cout * p->Cin * p->k * p->k;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dconv_wgrad_kernel, dim3(p->Cin * cdiv_i(p->Cout, d.OG), d.S), dim3(256), 0, s, d);
  hipLaunchKernelGGL(dconv_wgrad_reduce_kernel, dim3(cdiv_i(d.row, 256)), dim3(256), 0, s, (const float*)p->partial, d.S, d.row, d.row, p->scale, p->gw);
  if (p->gb) hipLaunchKernelGGL(dconv_bgrad_kernel, dim3(p->Cout), dim3(256), 0, s, p->dy, p->N, p->Cout, p->H * p->W, p->scale, p->gb);
  return rumpy_check_launch("rumpy_dconv_wgrad");
}

extern "C" int rumpy_mse_loss(const rumpy_mse_args* p, void* stream) {
  if (!p || !p->out || !p->target || !p->partial || !p->loss || p->n <= 0) { rumpy_set_error("rumpy_mse_loss: bad argument"); return RUMPY_E_ARG; }
  int blocks = (int)((p->n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, s, p->out, p->target, p->grad, p->n, 2.0f / (float)p->n, p->partial);
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)p->partial, blocks, 1.0f / (float)p->n, p->loss);
  return rumpy_check_launch("rumpy_mse_loss");
}

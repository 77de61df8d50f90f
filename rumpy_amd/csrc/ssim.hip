// SSIM with Gaussian weights as the reference's eval metrics compute it: rumpy/sr_tools/metrics.py:123-149 calls
// skimage.metrics.structural_similarity(a, b, data_range=1, gaussian_weights=True, use_sample_covariance=False, sigma=1.5)
// (scikit-image >= 0.16.2, requirements.txt:13 - third party, absent from this image; published algorithm: 11-tap Gaussian
// window = scipy.ndimage.gaussian_filter with truncate 3.5, local means / variances / covariance, the SSIM map
//   S = (2 ux uy + C1)(2 vxy + C2) / ((ux^2 + uy^2 + C1)(vx + vy + C2)),  C1 = (0.01 R)^2, C2 = (0.03 R)^2,
// cropped by (window-1)/2 = 5 pixels on every side before the mean).  Because the crop equals the window radius, no output that
// is averaged ever sees the filter's boundary handling: the kernel only evaluates interior pixels from in-image neighbours.
// One workgroup = a 32 x 32 block of interior pixels of one plane: 42 x 42 inputs of both planes in LDS, horizontal pass of the five
// moments (x, y, xx, yy, xy) into LDS, vertical pass + SSIM map + block sum; a second launch adds the block sums per plane in
// double precision.
#include "common.hpp"

constexpr int SS_T = 32, SS_R = 5, SS_IN = SS_T + 2 * SS_R;      // 42

struct SsimDev { const float* a; const float* b; float* partial; int P, H, W, tiles_x, tiles_y; float C1, C2; float g[2 * SS_R + 1]; };

__global__ void __launch_bounds__(256) ssim_kernel(SsimDev d) {
  __shared__ float sa[SS_IN][SS_IN + 1], sb[SS_IN][SS_IN + 1];
  __shared__ float hm[5][SS_IN][SS_T + 1];
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int tpp = d.tiles_x * d.tiles_y;
  const int p = blockIdx.x / tpp, t = blockIdx.x - p * tpp, ty = t / d.tiles_x, tx = t - ty * d.tiles_x;
  const int oy = ty * SS_T, ox = tx * SS_T;                 // top-left input pixel of the block (= interior pixel (oy, ox) + radius)
  const float* pa = d.a + (size_t)p * d.H * d.W;
  const float* pb = d.b + (size_t)p * d.H * d.W;
  for (int i = tid; i < SS_IN * SS_IN; i += 256) {
    const int r = i / SS_IN, c = i - r * SS_IN;
    const int y = min(oy + r, d.H - 1), x = min(ox + c, d.W - 1);      // clamped reads feed only outputs that are masked out below
    sa[r][c] = pa[(size_t)y * d.W + x];
    sb[r][c] = pb[(size_t)y * d.W + x];
  }
  __syncthreads();
  for (int i = tid; i < SS_IN * SS_T; i += 256) {           // horizontal pass: 42 rows x 32 columns x 5 moments
    const int r = i / SS_T, c = i - r * SS_T;
    float mx = 0.f, my = 0.f, mxx = 0.f, myy = 0.f, mxy = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * SS_R + 1; ++k) {
      const float x = sa[r][c + k], y = sb[r][c + k], w = d.g[k];
      mx = fmaf(w, x, mx); my = fmaf(w, y, my); mxx = fmaf(w, x * x, mxx); myy = fmaf(w, y * y, myy); mxy = fmaf(w, x * y, mxy);
    }
    hm[0][r][c] = mx; hm[1][r][c] = my; hm[2][r][c] = mxx; hm[3][r][c] = myy; hm[4][r][c] = mxy;
  }
  __syncthreads();
  float sum = 0.f;
  for (int i = tid; i < SS_T * SS_T; i += 256) {            // vertical pass + SSIM map
    const int r = i / SS_T, c = i - r * SS_T;
    float ux = 0.f, uy = 0.f, uxx = 0.f, uyy = 0.f, uxy = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * SS_R + 1; ++k) {
      const float w = d.g[k];
      ux = fmaf(w, hm[0][r + k][c], ux); uy = fmaf(w, hm[1][r + k][c], uy); uxx = fmaf(w, hm[2][r + k][c], uxx);
      uyy = fmaf(w, hm[3][r + k][c], uyy); uxy = fmaf(w, hm[4][r + k][c], uxy);
    }
    const float vx = uxx - ux * ux, vy = uyy - uy * uy, vxy = uxy - ux * uy;
    const float s = ((2.f * ux * uy + d.C1) * (2.f * vxy + d.C2)) / ((ux * ux + uy * uy + d.C1) * (vx + vy + d.C2));
    // interior pixel (oy + r, ox + c) of the cropped map exists if its centre oy + r + 5 < H - 5
    if (oy + r + 2 * SS_R < d.H && ox + c + 2 * SS_R < d.W) sum += s;
  }
  red[tid] = sum;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  if (tid == 0) d.partial[blockIdx.x] = red[0];
}

__global__ void ssim_finalize_kernel(const float* partial, int tiles, double inv_count, float* out) {
  __shared__ double red[256];
  const float* p = partial + (size_t)blockIdx.x * tiles;
  double s = 0.0;
  for (int i = threadIdx.x; i < tiles; i += 256) s += (double)p[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(red[0] * inv_count);
}

extern "C" int64_t rumpy_ssim_partial_floats(int32_t P, int32_t H, int32_t W) {
  if (P <= 0 || H <= 2 * SS_R || W <= 2 * SS_R) return 0;
  return (int64_t)P * ((H - 2 * SS_R + SS_T - 1) / SS_T) * ((W - 2 * SS_R + SS_T - 1) / SS_T);
}

extern "C" int rumpy_ssim(const rumpy_ssim_args* p, void* stream) {
  if (!p || !p->a || !p->b || !p->partial || !p->out || p->P <= 0) { rumpy_set_error("rumpy_ssim: bad argument"); return RUMPY_E_ARG; }
  if (p->H <= 2 * SS_R || p->W <= 2 * SS_R) { rumpy_set_error("rumpy_ssim: image smaller than the 11x11 window (%dx%d)", p->H, p->W); return RUMPY_E_ARG; }
  SsimDev d;
  d.a = p->a; d.b = p->b; d.partial = p->partial; d.P = p->P; d.H = p->H; d.W = p->W;
  d.tiles_x = (p->W - 2 * SS_R + SS_T - 1) / SS_T; d.tiles_y = (p->H - 2 * SS_R + SS_T - 1) / SS_T;
  d.C1 = (0.01f * p->data_range) * (0.01f * p->data_range); d.C2 = (0.03f * p->data_range) * (0.03f * p->data_range);
  double g[2 * SS_R + 1], gs = 0.0;                        // scipy.ndimage._gaussian_kernel1d(sigma = 1.5, radius = 5)
  for (int k = 0; k < 2 * SS_R + 1; ++k) { const double x = k - SS_R; g[k] = exp(-0.5 / (1.5 * 1.5) * x * x); gs += g[k]; }
  for (int k = 0; k < 2 * SS_R + 1; ++k) d.g[k] = (float)(g[k] / gs);
  hipStream_t s = (hipStream_t)stream;
  const int tiles = d.tiles_x * d.tiles_y;
  hipLaunchKernelGGL(ssim_kernel, dim3(p->P * tiles), dim3(256), 0, s, d);
  const double inv = 1.0 / ((double)(p->H - 2 * SS_R) * (double)(p->W - 2 * SS_R));
  hipLaunchKernelGGL(ssim_finalize_kernel, dim3(p->P), dim3(256), 0, s, p->partial, tiles, inv, p->out);
  return rumpy_check_launch("rumpy_ssim");
}

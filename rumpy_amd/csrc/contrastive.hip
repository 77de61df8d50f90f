// The contrastive head of the degradation-encoder training step in HIP (round 3; SURVEY.md 8f.4 names the MoCo / SupMoCo logits):
//   mlp head            rumpy/regression/models/contrastive_learning/encoding_models.py:43-55 (Linear 256-256, LeakyReLU 0.1, Linear 256-256)
//   L2 normalisation    moco.py:147,162 / supmoco.py:75,85 (nn.functional.normalize(., dim=1))
//   logits              moco.py:150-177 (q . k / T | q @ queue / T), supmoco.py:88-119 (positives from the queue by class label)
//   cross-entropy       handlers.py:57 (nn.CrossEntropyLoss on [N, 1 + K] logits)
// forward and backward.  Round 2 ran them as torch ops: rocBLAS GEMMs + ATen element-wise / softmax / reduce kernels with torch autograd, a
// quarter of the 1.1 ms MoCo step.  All matrices are small ([N <= 256, 256], [N, 8193], [256, 8192]) and the reference computes them in fp32,
// which a cosine / 0.07 needs: so this is exact-fp32 VALU code, not MFMA - one tiled GEMM kernel with the epilogues the head needs (bias,
// LeakyReLU, scale) and a split-K form for the two products that contract over the 8192 queue entries with only N x 256 outputs, plus
// row kernels for the normalisation and the softmax cross-entropy.  Every reduction runs in a fixed order: results are bitwise reproducible.
#include "common.hpp"

// ---------------------------------------------------------------------------------------------------------------- fp32 GEMM
// C[m, n] = act(alpha * sum_k A(m, k) * B(k, n) + bias[n]),  A(m, k) = A[m * sam + k * sak],  B(k, n) = B[k * sbk + n * sbn],  C row stride ldc.
// 64 x 64 tile per 256-thread workgroup, 16-deep steps through LDS, 4 x 4 outputs per thread.  grid (n tiles, m tiles, K splits); with more
// than one split every workgroup writes its partial tile to `partial` [split][M][N] and sgemm_reduce_kernel finishes (fixed split order).
struct Sgemm {
  const float* A; const float* B; float* C; const float* bias; float* partial;
  int M, N, K, ksplit, ldc; long long sam, sak, sbk, sbn; float alpha, slope; int accumulate;
};
constexpr int GM = 64, GN = 64, GK = 64;

__device__ __forceinline__ float head_act(float v, float slope) { return v > 0.f ? v : v * slope; }

// These products are small and latency-bound, not flop-bound ([32, 256] x [256, 256] is four workgroups): a step's 2 x 16 element loads per
// thread are all issued before its one barrier pair, so that a workgroup pays one memory latency per 64 k's (a 16-deep step took 2 us of
// dependent latency each: 33 us for the mlp head's 4 MFLOP).
__global__ void __launch_bounds__(256) sgemm_kernel(Sgemm a) {
  __shared__ __attribute__((aligned(16))) float As[GK][GM + 4];
  __shared__ __attribute__((aligned(16))) float Bs[GK][GN + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
  const int k0 = blockIdx.z * a.ksplit, k1 = min(a.K, k0 + a.ksplit);
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  // tile loads: consecutive threads walk the unit-stride direction of each operand (wave-uniform choice)
  const bool a_k_fast = a.sak == 1, b_n_fast = a.sbn == 1;
  for (int kk = k0; kk < k1; kk += GK) {
    float ra[16], rb[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = tid + 256 * e;                        // 4096 elements of each tile
      const int am = a_k_fast ? idx >> 6 : idx & 63, ak = a_k_fast ? idx & 63 : idx >> 6;
      const int m = m0 + am, k = kk + ak;
      // unconditional loads from clamped addresses, zeroed afterwards: a load under a condition makes the compiler branch around it and wait
      // for each one (32 dependent round trips per step: 25 us for a 4 MFLOP product)
      const bool oka = (m < a.M) & (k < k1);
      float va = a.A[oka ? m * a.sam + k * a.sak : 0];
      ra[e] = oka ? va : 0.f;
      const int bn = b_n_fast ? idx & 63 : idx >> 6, bk = b_n_fast ? idx >> 6 : idx & 63;
      const int n = n0 + bn, k2 = kk + bk;
      const bool okb = (n < a.N) & (k2 < k1);
      float vb = a.B[okb ? k2 * a.sbk + n * a.sbn : 0];
      rb[e] = okb ? vb : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = tid + 256 * e;
      const int am = a_k_fast ? idx >> 6 : idx & 63, ak = a_k_fast ? idx & 63 : idx >> 6;
      As[ak][am] = ra[e];
      const int bn = b_n_fast ? idx & 63 : idx >> 6, bk = b_n_fast ? idx >> 6 : idx & 63;
      Bs[bk][bn] = rb[e];
    }
    __syncthreads();
#pragma unroll 16
    for (int k = 0; k < GK; ++k) {
      const float4 av = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
      const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
    }
    __syncthreads();
  }
  // epilogue: bias and (accumulate) the old C values are fetched as two batches of unconditional loads, then everything is stored
  float bs[4] = {0.f, 0.f, 0.f, 0.f}, old[4][4];
  if (a.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] = a.bias[min(n0 + tx * 4 + j, a.N - 1)];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) old[i][j] = 0.f;
  if (a.accumulate && gridDim.z == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) old[i][j] = a.C[(size_t)min(m0 + ty * 4 + i, a.M - 1) * a.ldc + min(n0 + tx * 4 + j, a.N - 1)];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (m >= a.M || n >= a.N) continue;
      if (gridDim.z > 1) { a.partial[((size_t)blockIdx.z * a.M + m) * a.N + n] = acc[i][j]; continue; }
      a.C[(size_t)m * a.ldc + n] = old[i][j] + head_act(a.alpha * acc[i][j] + bs[j], a.slope);
    }
  }
}
__global__ void __launch_bounds__(64) sgemm_reduce_kernel(Sgemm a, int splits) {
  const size_t total = (size_t)a.M * a.N;
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < total; i += (size_t)gridDim.x * 64) {
    const int m = (int)(i / a.N), n = (int)(i - (size_t)m * a.N);
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= splits; z += 8) {                      // eight loads in flight; summed in split order all the same
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = a.partial[(size_t)(z + u) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < splits; ++z) s += a.partial[(size_t)z * total + i];
    const float bv = a.bias ? a.bias[n] : 0.f;                 // (wave-uniform conditions)
    float* c = a.C + (size_t)m * a.ldc + n;
    const float ov = a.accumulate ? *c : 0.f;
    *c = ov + head_act(a.alpha * s + bv, a.slope);
  }
}

extern "C" int64_t rumpy_sgemm_partial_floats(int32_t M, int32_t N, int32_t K) {
  const int tiles = ((M + GM - 1) / GM) * ((N + GN - 1) / GN);
  int splits = 1;
  // few output tiles: one wave per SIMD issues an FMA every 4 cycles, so a 64 x 64 x 256 tile alone is ~8 us of VALU issue on its CU - the K axis is
  // cut until about a hundred workgroups share the work (each keeps at least one 64-deep step)
  if (tiles < 64 && K >= 128) { splits = min(32, min(128 / max(1, tiles), K / 64)); if (splits < 1) splits = 1; }
  return splits > 1 ? (int64_t)splits * M * N : 0;
}
extern "C" int rumpy_sgemm(const rumpy_sgemm_args* p, void* stream) {
  if (!p || !p->A || !p->B || !p->C || p->M <= 0 || p->N <= 0 || p->K <= 0 || p->ldc < p->N) { rumpy_set_error("rumpy_sgemm: bad argument"); return RUMPY_E_ARG; }
  Sgemm a;
  a.A = p->A; a.B = p->B; a.C = p->C; a.bias = p->bias; a.partial = p->partial;
  a.M = p->M; a.N = p->N; a.K = p->K; a.ldc = p->ldc; a.sam = p->sam; a.sak = p->sak; a.sbk = p->sbk; a.sbn = p->sbn;
  a.alpha = p->alpha; a.slope = p->leaky_slope; a.accumulate = p->accumulate;
  const int tm = (p->M + GM - 1) / GM, tn = (p->N + GN - 1) / GN;
  const int64_t pf = rumpy_sgemm_partial_floats(p->M, p->N, p->K);
  int splits = pf ? (int)(pf / ((int64_t)p->M * p->N)) : 1;
  if (splits > 1 && !p->partial) { rumpy_set_error("rumpy_sgemm: this shape runs split over K and needs `partial` (rumpy_sgemm_partial_floats)"); return RUMPY_E_ARG; }
  a.ksplit = splits > 1 ? ((p->K + splits - 1) / splits + GK - 1) / GK * GK : p->K;
  if (splits > 1) splits = (p->K + a.ksplit - 1) / a.ksplit;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sgemm_kernel, dim3(tn, tm, splits), dim3(256), 0, s, a);
  if (splits > 1) {
    const size_t total = (size_t)p->M * p->N;
    hipLaunchKernelGGL(sgemm_reduce_kernel, dim3((unsigned)min((size_t)1024, (total + 63) / 64)), dim3(64), 0, s, a, splits);
  }
  return rumpy_check_launch("rumpy_sgemm");
}

// ---------------------------------------------------------------------------------------------------------------- row kernels
__device__ __forceinline__ float block_sum_256(float v, float* red) {      // fixed order: wave shuffles, then the 4 wave sums in order
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// y = x / max(||x||_2, eps) per row (nn.functional.normalize, eps 1e-12); inv[n] = 1 / max(||x||, eps).  One workgroup per row.
__global__ void __launch_bounds__(256) l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv, int C) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) { const float v = x[(size_t)n * C + c]; s = fmaf(v, v, s); }
  const float tot = block_sum_256(s, red);
  const float r = 1.f / fmaxf(sqrtf(tot), 1e-12f);
  for (int c = threadIdx.x; c < C; c += 256) y[(size_t)n * C + c] = x[(size_t)n * C + c] * r;
  if (threadIdx.x == 0 && inv) inv[n] = r;
}
// dx = (dy - y * (y . dy)) * inv
__global__ void __launch_bounds__(256) l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ inv,
                                                         float* __restrict__ dx, int C) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s = fmaf(y[(size_t)n * C + c], dy[(size_t)n * C + c], s);
  const float dot = block_sum_256(s, red);
  const float r = inv[n];
  for (int c = threadIdx.x; c < C; c += 256) dx[(size_t)n * C + c] = (dy[(size_t)n * C + c] - y[(size_t)n * C + c] * dot) * r;
}
extern "C" int rumpy_l2norm_rows(const float* x, float* y, float* inv, int32_t N, int32_t C, void* stream) {
  if (!x || !y || N <= 0 || C <= 0) { rumpy_set_error("rumpy_l2norm_rows: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, y, inv, C);
  return rumpy_check_launch("rumpy_l2norm_rows");
}
extern "C" int rumpy_l2norm_rows_bwd(const float* dy, const float* y, const float* inv, float* dx, int32_t N, int32_t C, void* stream) {
  if (!dy || !y || !inv || !dx || N <= 0 || C <= 0) { rumpy_set_error("rumpy_l2norm_rows_bwd: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, dy, y, inv, dx, C);
  return rumpy_check_launch("rumpy_l2norm_rows_bwd");
}

// out[n, col] = scale * sum_c a[n, c] * b[n, c]  (the positive logit q . v) ; one workgroup per row
__global__ void __launch_bounds__(256) rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int C, int ldo, float scale) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s = fmaf(a[(size_t)n * C + c], b[(size_t)n * C + c], s);
  const float tot = block_sum_256(s, red);
  if (threadIdx.x == 0) out[(size_t)n * ldo] = tot * scale;
}
extern "C" int rumpy_rowdot(const float* a, const float* b, float* out, int32_t N, int32_t C, int32_t ldo, float scale, void* stream) {
  if (!a || !b || !out || N <= 0 || C <= 0 || ldo <= 0) { rumpy_set_error("rumpy_rowdot: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(rowdot_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, a, b, out, C, ldo, scale);
  return rumpy_check_launch("rumpy_rowdot");
}

// SupMoCo's positives from the queue (supmoco.py:93-112): same[n, j] = (labels[n] == queue_labels[j]) as 0 / 1 floats, cnt[n] = its row sum
__global__ void __launch_bounds__(256) label_match_kernel(const long long* __restrict__ labels, const long long* __restrict__ qlabels, float* __restrict__ same,
                                                          float* __restrict__ cnt, int K) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  const long long l = labels[n];
  float s = 0.f;
  for (int j = threadIdx.x; j < K; j += 256) { const float v = qlabels[j] == l ? 1.f : 0.f; same[(size_t)n * K + j] = v; s += v; }
  const float tot = block_sum_256(s, red);
  if (threadIdx.x == 0) cnt[n] = tot;
}
extern "C" int rumpy_label_match(const void* labels, const void* queue_labels, float* same, float* cnt, int32_t N, int32_t K, void* stream) {
  if (!labels || !queue_labels || !same || !cnt || N <= 0 || K <= 0) { rumpy_set_error("rumpy_label_match: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(label_match_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const long long*)labels, (const long long*)queue_labels, same, cnt, K);
  return rumpy_check_launch("rumpy_label_match");
}

// v[n, c] = (ksum[n, c] + s[n, c]) * rscale / (P + cnt[n])   (s, cnt may be NULL: MoCo's v = kmean / T with ksum = sum over the P key crops)
__global__ void __launch_bounds__(256) pos_vector_kernel(const float* __restrict__ k, const float* __restrict__ s, const float* __restrict__ cnt, float* __restrict__ v,
                                                         int N, int P, int C, float rscale) {
  const int n = blockIdx.x;
  const float d = rscale / ((float)P + (cnt ? cnt[n] : 0.f));
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = s ? s[(size_t)n * C + c] : 0.f;
    for (int p = 0; p < P; ++p) t += k[((size_t)n * P + p) * C + c];          // the query's own key crops, in crop order
    v[(size_t)n * C + c] = t * d;
  }
}
extern "C" int rumpy_pos_vector(const float* k, const float* s, const float* cnt, float* v, int32_t N, int32_t P, int32_t C, float rscale, void* stream) {
  if (!k || !v || N <= 0 || P <= 0 || C <= 0 || ((s == nullptr) != (cnt == nullptr))) { rumpy_set_error("rumpy_pos_vector: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(pos_vector_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, k, s, cnt, v, N, P, C, rscale);
  return rumpy_check_launch("rumpy_pos_vector");
}

// ---- softmax cross-entropy over the rows of logits [N, M] (row stride M), integer targets: nn.CrossEntropyLoss(reduction='mean') ----
// forward: lse[n] = log sum_j exp(l[n, j]) ; rowloss[n] = lse[n] - l[n, t_n] ; loss = mean_n rowloss (one workgroup, fixed order)
__global__ void __launch_bounds__(256) ce_rows_kernel(const float* __restrict__ l, const long long* __restrict__ target, float* __restrict__ lse,
                                                      float* __restrict__ rowloss, int M) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  const float* row = l + (size_t)n * M;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < M; j += 256) mx = fmaxf(mx, row[j]);
  mx = block_max_256(mx, red);
  float s = 0.f;
  for (int j = threadIdx.x; j < M; j += 256) s += expf(row[j] - mx);
  const float tot = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    const float ls = mx + logf(tot);
    lse[n] = ls;
    rowloss[n] = ls - row[target[n]];
  }
}
__global__ void __launch_bounds__(256) ce_mean_kernel(const float* __restrict__ rowloss, float* __restrict__ loss, int N) {
  __shared__ float red[4];
  float s = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) s += rowloss[n];
  const float tot = block_sum_256(s, red);
  if (threadIdx.x == 0) *loss = tot / (float)N;
}
// backward: dl[n, j] = (exp(l[n, j] - lse[n]) - [j == t_n]) * (*gout) / N
__global__ void __launch_bounds__(256) ce_bwd_kernel(const float* __restrict__ l, const long long* __restrict__ target, const float* __restrict__ lse,
                                                     const float* __restrict__ gout, float* __restrict__ dl, int N, int M) {
  const int n = blockIdx.y;
  const float g = *gout / (float)N, ls = lse[n];
  const long long t = target[n];
  for (int j = blockIdx.x * 256 + threadIdx.x; j < M; j += gridDim.x * 256)
    dl[(size_t)n * M + j] = (expf(l[(size_t)n * M + j] - ls) - (j == t ? 1.f : 0.f)) * g;
}
extern "C" int rumpy_ce_rows(const float* logits, const void* target, float* lse, float* rowloss, float* loss, int32_t N, int32_t M, void* stream) {
  if (!logits || !target || !lse || !rowloss || !loss || N <= 0 || M <= 0) { rumpy_set_error("rumpy_ce_rows: bad argument"); return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(N), dim3(256), 0, s, logits, (const long long*)target, lse, rowloss, M);
  hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, s, rowloss, loss, N);
  return rumpy_check_launch("rumpy_ce_rows");
}
extern "C" int rumpy_ce_rows_bwd(const float* logits, const void* target, const float* lse, const float* gout, float* dlogits, int32_t N, int32_t M, void* stream) {
  if (!logits || !target || !lse || !gout || !dlogits || N <= 0 || M <= 0) { rumpy_set_error("rumpy_ce_rows_bwd: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)min(16, (M + 255) / 256), N), dim3(256), 0, (hipStream_t)stream, logits, (const long long*)target, lse, gout, dlogits, N, M);
  return rumpy_check_launch("rumpy_ce_rows_bwd");
}

// column sums of x [N, C] -> out [C] (bias gradients of the mlp head), fixed order
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += x[(size_t)n * C + c];
  out[c] = s;
}
// dh *= (h > 0 ? 1 : slope) where h is the LeakyReLU OUTPUT (slope > 0 keeps the sign)
__global__ void __launch_bounds__(256) lrelu_bwd_kernel(float* __restrict__ dh, const float* __restrict__ h, size_t n, float slope) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dh[i] = h[i] > 0.f ? dh[i] : dh[i] * slope;
}
extern "C" int rumpy_colsum(const float* x, float* out, int32_t N, int32_t C, void* stream) {
  if (!x || !out || N <= 0 || C <= 0) { rumpy_set_error("rumpy_colsum: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, N, C);
  return rumpy_check_launch("rumpy_colsum");
}
extern "C" int rumpy_lrelu_bwd(float* dh, const float* h, int64_t n, float slope, void* stream) {
  if (!dh || !h || n <= 0) { rumpy_set_error("rumpy_lrelu_bwd: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3((unsigned)min((int64_t)1024, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh, h, (size_t)n, slope);
  return rumpy_check_launch("rumpy_lrelu_bwd");
}

// dst[n, c] += coef[n * ldcoef] * src[n, c]   (the positive logit's share of d loss / d q_normalised: dl[n, 0] * v[n, :])
__global__ void __launch_bounds__(256) row_axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, const float* __restrict__ coef, int C, int ldcoef) {
  const int n = blockIdx.x;
  const float a = coef[(size_t)n * ldcoef];
  for (int c = threadIdx.x; c < C; c += 256) dst[(size_t)n * C + c] = fmaf(a, src[(size_t)n * C + c], dst[(size_t)n * C + c]);
}
extern "C" int rumpy_row_axpy(float* dst, const float* src, const float* coef, int32_t N, int32_t C, int32_t ldcoef, void* stream) {
  if (!dst || !src || !coef || N <= 0 || C <= 0 || ldcoef <= 0) { rumpy_set_error("rumpy_row_axpy: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(row_axpy_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, dst, src, coef, C, ldcoef);
  return rumpy_check_launch("rumpy_row_axpy");
}

// MoCo's _dequeue_and_enqueue (moco.py:74-89; supmoco.py:34-50 with the label track) in one launch: queue[:, slot[i]] = keys[i * kstride, :]
// (one key per query), queue_labels[slot[i]] = labels[i], then the device-side slot vector and the queue pointer advance by n (mod K).
__global__ void __launch_bounds__(256) moco_enqueue_kernel(float* __restrict__ queue, const float* __restrict__ keys, long long* __restrict__ slots,
                                                           long long* __restrict__ queue_ptr, long long* __restrict__ qlabels,
                                                           const long long* __restrict__ labels, int n, int kstride, int C, int K) {
  const int i = blockIdx.x;                        // one workgroup per enqueued key: it alone reads and advances slots[i]
  const long long sl = slots[i];
  for (int c = threadIdx.x; c < C; c += 256) queue[(size_t)c * K + sl] = keys[(size_t)i * kstride * C + c];
  __syncthreads();
  if (threadIdx.x == 0) {
    if (qlabels) qlabels[sl] = labels[i];
    slots[i] = (sl + n) % K;
    if (i == 0 && queue_ptr) *queue_ptr = (*queue_ptr + n) % K;
  }
}
extern "C" int rumpy_moco_enqueue(float* queue, const float* keys, void* slots, void* queue_ptr, void* queue_labels, const void* labels,
                                  int32_t n, int32_t key_stride, int32_t C, int32_t K, void* stream) {
  if (!queue || !keys || !slots || n <= 0 || key_stride <= 0 || C <= 0 || K <= 0 || ((queue_labels == nullptr) != (labels == nullptr))) {
    rumpy_set_error("rumpy_moco_enqueue: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(moco_enqueue_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, queue, keys, (long long*)slots, (long long*)queue_ptr,
                     (long long*)queue_labels, (const long long*)labels, n, key_stride, C, K);
  return rumpy_check_launch("rumpy_moco_enqueue");
}

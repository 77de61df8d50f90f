// Weight gradient of the 3x3 convolutions on bf16 MFMA, output-stationary, grouped over a job table.
//
//   dW[co][ky][kx][ci] = sum over pixels p of dy[p][co] * x[p + (ky-1, kx-1)][ci]
// is a GEMM whose K dimension is the PIXEL index, the slow axis of both NHWC operands, so both MFMA operands
// are read from LDS with the gfx950 transposing load ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane
// group, delivered channel-major).  One job = (layer, cin chunk, cout tile, image range): the workgroup
// streams 8x16-pixel tiles (x halo tile + dy tile, double buffered in LDS), wave w accumulates
// D[co 16*ct..][ci 16*w..] for every co tile ct and all 9 taps in registers (36 x 16x16 f32 tiles), and
// finally leaves its partial sums in the job's fp32 slab [co][tap][ci] (+ bias sums).  A second kernel adds
// the slabs of a layer in a fixed order (bitwise reproducible), scales, and scatters to OIHW.
// K order inside a 32-pixel k-step (rows 2t, 2t+1 of the tile): lane group g, element e ->
//   row 2t + g/2, column 4*(g%2) + e (e < 4) or 8 + 4*(g%2) + e - 4 (e >= 4): each 32-lane half touches 8
//   consecutive pixels per read, conflict-free at the 160-B pixel stride.
#include "common.hpp"
#include <cstdlib>

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

__device__ __forceinline__ short4v tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p));
}
__device__ __forceinline__ bf16x8 join8(short4v a, short4v b) {
  union { short8v s; bf16x8 h; } c;
  c.s = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return c.h;
}

constexpr int WG_STAGE_BYTES = X_STAGE_BYTES + DY_STAGE_BYTES;  // 49280

struct DyView { const uint16_t* dy; int H, W, dy_mode, dy_cstride, dy_coff; };

template <bool DY4>
__device__ __forceinline__ void dy_issue(uint4 (&R)[4], const DyView j, int n, int ty, int tx, int tid) {
  const uint16_t* dy = j.dy;
  if (DY4) {
    // [N,H,W,4]: one 8-byte piece per pixel, handled by the first 128 threads
    uint4 v = make_uint4(0, 0, 0, 0);
    if (tid < TH * TW) {
      const int r = tid >> 4, c = tid & 15;
      const int y = ty * TH + r, x = tx * TW + c;
      if (y < j.H && x < j.W) {
        const uint2 u = *reinterpret_cast<const uint2*>(dy + ((size_t)(n * j.H + y) * j.W + x) * 4);
        v.x = u.x; v.y = u.y;
      }
    }
    R[0] = v; R[1] = v; R[2] = v; R[3] = v;
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + 256 * i;
    const int pix = p >> 3, part = p & 7;
    const int r = pix >> 4, c = pix & 15;
    const int y = ty * TH + r, x = tx * TW + c;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (y < j.H && x < j.W) {
      size_t e;
      if (j.dy_mode == 0) e = ((size_t)(n * j.H + y) * j.W + x) * j.dy_cstride + j.dy_coff + part * 8;
      else e = ((size_t)(n * 2 * j.H + 2 * y + (j.dy_coff >> 1)) * (2 * j.W) + 2 * x + (j.dy_coff & 1)) * 64 + part * 8;
      v = *reinterpret_cast<const uint4*>(dy + e);
    }
    R[i] = v;
  }
}
template <bool DY4>
__device__ __forceinline__ void dy_write(const uint4 (&R)[4], unsigned char* lds, int tid) {
  if (DY4) {
    if (tid < TH * TW) *reinterpret_cast<uint2*>(lds + tid * PIX_STRIDE) = make_uint2(R[0].x, R[0].y);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + 256 * i;
    *reinterpret_cast<uint4*>(lds + (p >> 3) * PIX_STRIDE + (p & 7) * 16) = R[i];
  }
}

template <int MT>
__global__ void __launch_bounds__(256, 1) wgrad_kernel(const rumpy_wgrad_job* __restrict__ jobs) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * WG_STAGE_BYTES];
  const rumpy_wgrad_job* jp = jobs + blockIdx.x;
  struct { const uint16_t* x; float* slab; int n0, n1, t0, t1, H, W, x_cstride, x_coff, dy_mode; } j;
  j.x = (const uint16_t*)jp->x; j.slab = jp->slab; j.n0 = jp->n0; j.n1 = jp->n1; j.t0 = jp->t0; j.t1 = jp->t1; j.H = jp->H; j.W = jp->W;
  j.x_cstride = jp->x_cstride; j.x_coff = jp->x_coff; j.dy_mode = jp->dy_mode;
  DyView dv;
  dv.dy = (const uint16_t*)jp->dy; dv.H = j.H; dv.W = j.W; dv.dy_mode = j.dy_mode; dv.dy_cstride = jp->dy_cstride; dv.dy_coff = jp->dy_coff;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  const int tiles_x = (j.W + TW - 1) / TW, tiles_y = (j.H + TH - 1) / TH;
  const int ntiles = j.t1 - j.t0;   // tiles [t0, t1) of the image range

  if (MT == 1) {  // channels 4..15 of the dy image are never written: clear both dy buffers once
    for (int i = tid; i < 2 * (DY_STAGE_BYTES / 16); i += 256) {
      const int b = i / (DY_STAGE_BYTES / 16), o = i - b * (DY_STAGE_BYTES / 16);
      *reinterpret_cast<uint4*>(lds + b * WG_STAGE_BYTES + X_STAGE_BYTES + o * 16) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
  }

  f32x4 acc[MT][9];
#pragma unroll
  for (int ct = 0; ct < MT; ++ct)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[ct][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  uint4 RX[6], RD[4];
  if (ntiles > 0) {
    const TileCoord t = decode_tile(j.t0, tiles_x, tiles_y);
    halo_issue(RX, j.x, 0, j.x_cstride, j.x_coff, j.n0 + t.n, t.ty, t.tx, j.H, j.W, tid);
    dy_issue<MT == 1>(RD, dv, j.n0 + t.n, t.ty, t.tx, tid);
    halo_write(RX, lds, tid);
    dy_write<MT == 1>(RD, lds + X_STAGE_BYTES, tid);
  }
  __syncthreads();
  int buf = 0;
  // per-lane constants of the transposed reads
  const int rsel = g >> 1;               // row within the 2-row k-step
  const int col0 = 4 * (g & 1) + q;      // first read: pixel column ; second read: col0 + 8
  const int a_off = p4 * 8;              // 4 channels = 8 bytes within the 16-channel block
  for (int tile = 0; tile < ntiles; ++tile) {
    const bool has_next = tile + 1 < ntiles;
    if (has_next) {
      const TileCoord tn = decode_tile(j.t0 + tile + 1, tiles_x, tiles_y);
      halo_issue(RX, j.x, 0, j.x_cstride, j.x_coff, j.n0 + tn.n, tn.ty, tn.tx, j.H, j.W, tid);
      dy_issue<MT == 1>(RD, dv, j.n0 + tn.n, tn.ty, tn.tx, tid);
    }
    const unsigned char* xs = lds + buf * WG_STAGE_BYTES;
    const unsigned char* ds = xs + X_STAGE_BYTES;
#pragma unroll
    for (int t = 0; t < TH / 2; ++t) {
      const int row = 2 * t + rsel;
      bf16x8 A[MT];
#pragma unroll
      for (int ct = 0; ct < MT; ++ct) {
        const unsigned char* pa = ds + (row * TW + col0) * PIX_STRIDE + ct * 32 + a_off;
        A[ct] = join8(tr_read(pa), tr_read(pa + 8 * PIX_STRIDE));
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const unsigned char* pb = xs + ((row + ky) * HALO_W + col0 + kx) * PIX_STRIDE + wave * 32 + a_off;
        const bf16x8 B = join8(tr_read(pb), tr_read(pb + 8 * PIX_STRIDE));
#pragma unroll
        for (int ct = 0; ct < MT; ++ct)
          acc[ct][tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ct], B, acc[ct][tap], 0, 0, 0);
      }
    }
    // bias gradient: thread (channel co = tid & 63, quarter = tid >> 6) adds 32 pixels of the dy tile
    {
      const int co = tid & 63, qt = tid >> 6;
      if (co < 16 * MT) {
        float s = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
          const uint16_t hv = *reinterpret_cast<const uint16_t*>(ds + (qt * 32 + i) * PIX_STRIDE + co * 2);
          s += bf16_bits_to_f32(hv);
        }
        bsum += s;
      }
    }
    if (has_next) {
      halo_write(RX, lds + (buf ^ 1) * WG_STAGE_BYTES, tid);
      dy_write<MT == 1>(RD, lds + (buf ^ 1) * WG_STAGE_BYTES + X_STAGE_BYTES, tid);
    }
    __syncthreads();
    buf ^= 1;
  }
  // ---- slab: [co 16*MT][tap 9][ci 64] then [16*MT] bias sums.  D row = co (4g+e), D col = ci (lane & 15) ----
  float* slab = j.slab;
  const int ci = 16 * wave + (lane & 15);
#pragma unroll
  for (int ct = 0; ct < MT; ++ct)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int e = 0; e < 4; ++e) slab[((16 * ct + 4 * g + e) * 9 + tap) * 64 + ci] = acc[ct][tap][e];
  float* red = reinterpret_cast<float*>(lds);
  red[tid] = bsum;
  __syncthreads();
  if (tid < 16 * MT) slab[16 * MT * 576 + tid] = (red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192]);
}

// one thread per (c, tap, ci) element of an item; fixed summation order over the item's jobs
__global__ void wgrad_reduce_kernel(const rumpy_reduce_item* __restrict__ items) {
  const rumpy_reduce_item it = items[blockIdx.y];
  const int nelem = it.co_count * 576;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nelem + it.co_count; e += gridDim.x * blockDim.x) {
    if (e < nelem) {
      const int c = e / 576, rem = e - c * 576, tap = rem >> 6, ci = rem & 63;
      float s = 0.f;
      for (int k = 0; k < it.njobs; ++k) s += it.slab[(size_t)k * it.slab_stride + e];
      const int co = it.co_mode ? 4 * c + it.co_off : it.co_off + c;
      it.gw[((size_t)co * it.ci_total + it.ci_off + ci) * 9 + tap] = s * it.scale;
    } else if (it.write_bias) {
      const int c = e - nelem;
      float s = 0.f;
      for (int k = 0; k < it.njobs; ++k) s += it.slab[(size_t)k * it.slab_stride + 16 * it.mt * 576 + c];
      const int co = it.co_mode ? 4 * c + it.co_off : it.co_off + c;
      it.gb[co] = s * it.scale;
    }
  }
}

extern "C" int64_t rumpy_wgrad_slab_floats(int32_t mt) { return (int64_t)16 * mt * 576 + 16 * mt; }

int rumpy_wgrad_dma_launch(const rumpy_wgrad_job* jobs_device, int njobs, int mt, hipStream_t s);   // wgrad_dma.hip
int rumpy_wgrad_dma_launch_shares(const rumpy_wgrad_job* jobs_device, const int* first_device, int nshares, hipStream_t s);

extern "C" int rumpy_wgrad_shares(const rumpy_wgrad_job* jobs_device, const int32_t* first_device, int32_t nshares, void* stream) {
  if (!jobs_device || !first_device || nshares <= 0) { rumpy_set_error("rumpy_wgrad_shares: bad argument"); return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  rumpy_probe_pre(2, s);
  rumpy_wgrad_dma_launch_shares(jobs_device, first_device, nshares, s);
  rumpy_probe_post(2, s);
  return rumpy_check_launch("rumpy_wgrad_shares");
}

extern "C" int rumpy_wgrad_grouped(const rumpy_wgrad_job* jobs_device, int32_t njobs, int32_t mt, int32_t variant, void* stream) {
  if (!jobs_device || njobs <= 0 || (mt != 1 && mt != 4)) { rumpy_set_error("rumpy_wgrad_grouped: bad argument"); return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  rumpy_probe_pre(2, s);
  static const bool use_old = getenv("RUMPY_WGRAD_OLD") != nullptr;   // A/B switch for the register-staged kernel
  if (variant == 0 && !use_old) rumpy_wgrad_dma_launch(jobs_device, njobs, mt, s);
  else if (mt == 4) hipLaunchKernelGGL(wgrad_kernel<4>, dim3(njobs), dim3(256), 0, s, jobs_device);
  else hipLaunchKernelGGL(wgrad_kernel<1>, dim3(njobs), dim3(256), 0, s, jobs_device);
  rumpy_probe_post(2, s);
  return rumpy_check_launch("rumpy_wgrad_grouped");
}

extern "C" int rumpy_wgrad_reduce(const rumpy_reduce_item* items_device, int32_t nitems, void* stream) {
  if (!items_device || nitems <= 0) { rumpy_set_error("rumpy_wgrad_reduce: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(145, nitems), dim3(256), 0, (hipStream_t)stream, items_device);
  return rumpy_check_launch("rumpy_wgrad_reduce");
}

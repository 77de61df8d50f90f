// 3x3 same-padding convolution as an implicit GEMM on bf16 MFMA (gfx950), filter-stationary.
//
// One workgroup = 4 waves = one 64-output-channel tile; wave w owns output channels 16w..16w+15 and keeps
// its slice of the filter (16 co x 576 k = 18 MFMA A-fragments = 72 VGPRs per 64-channel input chunk) in
// registers for the whole kernel.  The workgroup is persistent: it walks 8x16-pixel tiles; each stage's
// 10x18-pixel x 64-channel halo tile is staged HBM -> registers -> LDS one stage ahead (two LDS buffers, one
// barrier per stage), and every wave reads it as MFMA B-fragments (ds_read_b128, conflict-free at the
// 160-B pixel stride).  One B-fragment (input row r, tap column kx, channel half) feeds the three output
// rows r-2..r (ky = 2,1,0), so LDS traffic is 10 reads per 24 MFMAs.
//   D[co][px] (16x16 f32) += A[co][k = 32 input channels of one tap] * B[k][px]
// The epilogue works on 4 consecutive channels per lane (8-byte accesses): bias, ReLU, scale, ReLU-mask
// (backward), per-tile channel sums (channel attention), two residual adds, PixelShuffle scatter.
#include "common.hpp"
#include <cstdlib>


template <int CHUNKS, int FMT = RUMPY_FMT_BF16>      // FMT: element format of activations and filters (RUMPY_FMT_F16: evaluation plans)
__global__ void __launch_bounds__(256, (CHUNKS == 1) ? 2 : 1) conv3x3_kernel(ConvDev a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * X_STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int ct = blockIdx.y;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const int cstride = 64 * CHUNKS;

  // stationary filter fragments: packed [ct][chunk][wave][s = tap*2 + half][lane] x 16 B
  bf16x8 F[CHUNKS][18];
  {
    const uint4* wp = a.w + (size_t)ct * CHUNKS * (4 * 18 * 64);
#pragma unroll
    for (int ch = 0; ch < CHUNKS; ++ch)
#pragma unroll
      for (int s = 0; s < 18; ++s) F[ch][s] = as_bf16x8(wp[((ch * 4 + wave) * 18 + s) * 64 + lane]);
  }
  float bj[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
    const float4 b4 = *reinterpret_cast<const float4*>(a.bias + ct * 64 + 16 * wave + 4 * g);
    bj[0] = b4.x; bj[1] = b4.y; bj[2] = b4.z; bj[3] = b4.w;
  }

  // Register-staged input pipeline (a stage = one 64-channel input chunk of one tile): while stage s is multiplied out of
  // LDS, stage s+1 is in flight into registers.  The request is issued on every path (past the end: the current tile
  // again, unused) so that the waits stay counted.  (Two stages deep measured no faster here and spills: with the whole
  // 256-channel filter in registers - 288 of them - this kernel runs one wave per SIMD.)
  int tile = blockIdx.x;                                // (the XCD-contiguous order of common.hpp measured no gain here)
  if (tile >= ntiles) return;
  const int gstride = (int)gridDim.x;
  uint4 R[6];
  {
    const TileCoord t = decode_tile(tile, a.tiles_x, a.tiles_y);
    halo_issue(R, a.x, a.in_mode, cstride, 0, t.n, t.ty, t.tx, a.H, a.W, tid);
    halo_write(R, lds, tid);
  }
  __syncthreads();
  int buf = 0;

  for (; tile < ntiles; tile += gstride) {
    const TileCoord tc = decode_tile(tile, a.tiles_x, a.tiles_y);
    f32x4 acc[TH];
#pragma unroll
    for (int r = 0; r < TH; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int ch = 0; ch < CHUNKS; ++ch) {
      {
        const int ntile0 = (ch + 1 < CHUNKS) ? tile : tile + gstride;
        const int nch = (ch + 1 < CHUNKS) ? ch + 1 : 0;
        const int ntile = (ntile0 < ntiles) ? ntile0 : tile;
        const TileCoord tn = decode_tile(ntile, a.tiles_x, a.tiles_y);
        halo_issue(R, a.x, a.in_mode, cstride, a.in_mode == 0 ? nch * 64 : nch, tn.n, tn.ty, tn.tx, a.H, a.W, tid);
      }
      const bool has_next = (ch + 1 < CHUNKS) || (tile + gstride < ntiles);
      const unsigned char* cur = lds + buf * X_STAGE_BYTES;
      {
        // 6 groups (tap column, channel half) of 10 B-fragment reads + 24 MFMAs, software pipelined as in conv_strip.hip:
        // the reads of group i+1 are issued before the MFMAs of group i (one wave per SIMD here: nothing else hides them)
        bf16x8 I[2][HALO_H];
        auto load_group = [&](int grp, bf16x8 (&dst)[HALO_H]) {
          const int kx = grp >> 1, half = grp & 1;
#pragma unroll
          for (int r = 0; r < HALO_H; ++r)
            dst[r] = *reinterpret_cast<const bf16x8*>(cur + (r * HALO_W + px + kx) * PIX_STRIDE + half * 64 + g * 16);
        };
        load_group(0, I[0]);
#pragma unroll
        for (int grp = 0; grp < 6; ++grp) {
          if (grp + 1 < 6) load_group(grp + 1, I[(grp + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          const int kx = grp >> 1, half = grp & 1;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int r = 0; r < TH; ++r)
              acc[r] = mfma16<FMT>(F[ch][(ky * 3 + kx) * 2 + half], I[grp & 1][r + ky], acc[r]);
        }
      }
      // stage s+1 goes to the other LDS buffer BEFORE the epilogue's stores (a wait issued behind them would drain them)
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) halo_write(R, lds + (buf ^ 1) * X_STAGE_BYTES, tid);
      if (ch == CHUNKS - 1) {
        // ---- epilogue: lane holds channels c0..c0+3 of pixel (row r, column px) ----
        const int c0 = 16 * wave + 4 * g;
        const int xx = tc.tx * TW + px;
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < TH; ++r) {
          const int y = tc.ty * TH + r;
          if (y < a.H && xx < a.W) {
            size_t o;
            if (a.out_mode == 0) o = ((size_t)(tc.n * a.H + y) * a.W + xx) * (64 * a.cout_tiles) + ct * 64 + c0;
            else o = ((size_t)(tc.n * 2 * a.H + 2 * y + (ct >> 1)) * (2 * a.W) + 2 * xx + (ct & 1)) * 64 + c0;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = acc[r][j] + bj[j];
              if (a.relu) v[j] = relu_f32(v[j]);
              v[j] *= a.scale;
            }
            if (a.mask) {
              float m[4];
              unpack4<FMT>(*reinterpret_cast<const uint2*>(a.mask + o), m);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = (m[j] > 0.f) ? v[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) ps[j] += v[j];
            if (a.res1) {
              float m[4];
              unpack4<FMT>(*reinterpret_cast<const uint2*>(a.res1 + o), m);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] += m[j];
            }
            if (a.res2) {
              float m[4];
              unpack4<FMT>(*reinterpret_cast<const uint2*>(a.res2 + o), m);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] += m[j];
            }
            *reinterpret_cast<uint2*>(a.out + o) = pack4<FMT>(v[0], v[1], v[2], v[3]);
          }
        }
        if (a.pool) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float s = ps[j];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
            ps[j] = s;
          }
          if (px == 0) {
            float* pp = a.pool + ((size_t)(tc.n * a.tiles_y + tc.ty) * a.tiles_x + tc.tx) * (64 * a.cout_tiles) + ct * 64 + c0;
            *reinterpret_cast<float4*>(pp) = make_float4(ps[0], ps[1], ps[2], ps[3]);
          }
        }
      }
      __syncthreads();
      buf ^= 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// tail conv 64 -> C (<=4): every wave holds the same 16-row filter block (rows >= C are zero) and takes two
// of the tile's eight rows; output fp32 NCHW, optional fused L1 loss + sign gradient.  HBM-bound (reads the
// largest activation), so MFMA row waste is irrelevant.
// ------------------------------------------------------------------------------------------------------
struct TailDev {
  const uint16_t* x; const uint4* w; const float* bias; float* out; const float* target; uint16_t* dy4;
  float* loss_partial; float* wslab; int N, C, H, W, tiles_x, tiles_y; unsigned* nonfinite; const float* const* target_ind;
};

typedef __attribute__((address_space(3))) short4v* tail_lds_s4;
typedef __attribute__((address_space(3))) unsigned char* tail_lds_u8;
__device__ __forceinline__ bf16x8 tail_tr8(unsigned addr, unsigned second) {      // two transposed 4x16 reads -> 8 K values
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tail_lds_s4)(size_t)addr);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tail_lds_s4)(size_t)(addr + second));
  union { short8v s; bf16x8 h; } c;
  c.s = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return c.h;
}
// WGRAD = true (training with the fused L1 loss): the weight gradient of this conv, dW[co][tap][ci] = sum_px dy[px][co] * x[px+tap][ci],
// is accumulated here as well - the input tile is in LDS and the sign gradient has just been computed, so the separate
// weight-gradient pass (which re-read the 151 MB activation: 71 us) disappears.  Pixels are the MFMA K axis: both operands
// come out of LDS through transposed reads (ds_read_b64_tr_b16, as in wgrad_dma.hip); wave w owns input channels 16w..16w+15,
// 9 accumulator tiles stay in registers for the whole kernel, each workgroup leaves one slab (format of rumpy_wgrad_slab_floats(1)).
// FMT: element format of x and w (RUMPY_FMT_F16: evaluation plans, WGRAD = false only)
template <bool WGRAD, int FMT = RUMPY_FMT_BF16>
__global__ void __launch_bounds__(256, 2) tail_fwd_kernel(TailDev a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * X_STAGE_BYTES];
  // WGRAD: the tile's sign gradient as THREE shifted copies [kx][halo index h][4 ch] bf16, entry h = r * 18 + c + kx = dy(r, c) (zero elsewhere;
  // 160 entries = 5 MFMA K steps) + 16 zero bytes: against input rows ky .. ky + 7 of the halo tile (flattened index ky * 18 + h) the MFMA rows
  // (kx, channel) give all three taps of a filter row at once - 15 MFMAs and 35 LDS reads per tile where one tap per MFMA took 36 and 80
  constexpr int DYC = 160 * 8;
  __shared__ __attribute__((aligned(16))) unsigned char ldy[WGRAD ? 3 * DYC + 16 : 16];
  __shared__ float red[4];
  __shared__ float redb[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const float* const target = a.target_ind ? load_global_ptr(a.target_ind) : a.target;      // the batch of this replay (rumpy_set_pointers)
  // forward filter: stationary in 72 VGPRs (round 2: in the WGRAD form too - its weight-gradient accumulators shrank from 36 to 12 registers;
  // the filter used to sit in 18 KB of LDS there, one more fragment read per MFMA)
  bf16x8 F[18];
#pragma unroll
  for (int s = 0; s < 18; ++s) F[s] = as_bf16x8(a.w[s * 64 + lane]);
  // fp16 evaluation plans: the filter as TWO fp16 images, the rounded filter and its rounding residual (rumpy_pack_weights, kind 2, fmt F16),
  // multiplied one after the other into the same accumulators.  This layer's weights shape the image itself: their fp16 rounding is a fixed
  // perturbation of the output, CORRELATED with the model's own error - it moved the Y-PSNR of 32 dB EDSRs by up to -0.034 dB (the other
  // layers' weight rounding: +0.006; all activation rounding together: 0.002; tests/tools/psnr_seeds.py).  HBM-bound kernel: free.
  // Round 4: the residual image lives in LDS (18 KB, lane-linear 16-byte fragments: conflict-free), not in 72 more registers - with both
  // images in registers this build spilled 22 VGPRs to scratch (92 bytes per lane), and every reload sat in the vector-memory queue in front
  // of the counted waits of the input pipeline: 239 us for the 354 MB of a DIV2K-size evaluation image (1.5 TB/s) where the bf16 build
  // of the same kernel moves its bytes at 3.4 TB/s.
  __shared__ __attribute__((aligned(16))) uint4 flo[FMT == RUMPY_FMT_F16 ? 18 * 64 : 1];
  if (FMT == RUMPY_FMT_F16) {
    for (int i = tid; i < 18 * 64; i += 256) flo[i % (FMT == RUMPY_FMT_F16 ? 18 * 64 : 1)] = a.w[18 * 64 + i];
  }
  float bj[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < a.C; ++j) bj[j] = a.bias[j];
  float lsum = 0.f;
  unsigned bad = 0u;
  f32x4 wacc[WGRAD ? 3 : 1];            // per filter row ky: D rows (kx, channel) x 16 input channels of this wave
#pragma unroll
  for (int t = 0; t < (WGRAD ? 3 : 1); ++t) wacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  if (WGRAD) {                          // entries no tile pixel maps to stay zero for the whole kernel; every other one is rewritten per tile
    for (int i = tid; i < (3 * DYC + 16) / 4; i += 256) *reinterpret_cast<unsigned*>(ldy + i * 4) = 0u;
  }

  // Register-staged input pipeline, TWO tiles deep: while tile t is multiplied out of LDS, tile t+1 sits in registers (its
  // loads were issued one iteration ago) and the loads of tile t+2 are issued - every load has two iterations to land, so
  // the kernel is no longer one HBM latency per tile (it reads 151 MB on the headline shape: a bandwidth kernel).
  const int stride = (int)gridDim.x;
  int tile = xcd_strip(blockIdx.x, gridDim.x);         // contiguous runs of tiles per XCD (halo overlap of neighbours in L2)
  uint4 R[2][6];
  if (tile < ntiles) {
    const TileCoord t = decode_tile(tile, a.tiles_x, a.tiles_y);
    halo_issue(R[0], a.x, 0, 64, 0, t.n, t.ty, t.tx, a.H, a.W, tid);
    halo_write(R[0], lds, tid);
  }
  if (tile < ntiles) {
    const TileCoord t = decode_tile((tile + stride < ntiles) ? tile + stride : tile, a.tiles_x, a.tiles_y);
    halo_issue(R[1], a.x, 0, 64, 0, t.n, t.ty, t.tx, a.H, a.W, tid);
  }
  __syncthreads();
  auto step = [&](int cur_tile, const unsigned char* cur, uint4 (&Rnext)[6], uint4 (&Rfar)[6], unsigned char* other) {
    const TileCoord tc = decode_tile(cur_tile, a.tiles_x, a.tiles_y);
    // Target values of this tile first (lanes 0..15 own the pixels; 4 bytes per channel plane), then the far tile's halo.
    // Both groups are issued on EVERY path with clamped addresses (no target / outside the image / past the last tile:
    // some valid address, value unused): a fixed number of loads per iteration lets the compiler wait with counted
    // vmcnt values - the epilogue below used to drain the whole queue once per channel.
    const int xx = tc.tx * TW + px;
    const float* tsrc = target ? target : a.out;
    float tg[2][4];
    size_t obase[2];
    bool inimg[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int y = tc.ty * TH + 2 * wave + r;
      inimg[r] = (g == 0) && y < a.H && xx < a.W;
      obase[r] = inimg[r] ? ((size_t)tc.n * a.C * a.H + y) * a.W + xx : 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) tg[r][j] = tsrc[(j < a.C) ? obase[r] + (size_t)j * a.H * a.W : 0];
    }
    __builtin_amdgcn_sched_barrier(0);      // targets first in the queue: they are waited for with vmcnt(6), under the far loads
    const int far = (cur_tile + 2 * stride < ntiles) ? cur_tile + 2 * stride : cur_tile;
    {
      const TileCoord tn = decode_tile(far, a.tiles_x, a.tiles_y);
      halo_issue(Rfar, a.x, 0, 64, 0, tn.n, tn.ty, tn.tx, a.H, a.W, tid);
    }
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8 I[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          I[r] = *reinterpret_cast<const bf16x8*>(cur + ((2 * wave + r) * HALO_W + px + kx) * PIX_STRIDE + half * 64 + g * 16);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            acc[r] = mfma16<FMT>(F[(ky * 3 + kx) * 2 + half], I[r + ky], acc[r]);
            if (FMT == RUMPY_FMT_F16) acc[r] = mfma16<FMT>(as_bf16x8(flo[(((ky * 3 + kx) * 2 + half) * 64 + lane) % (FMT == RUMPY_FMT_F16 ? 18 * 64 : 1)]), I[r + ky], acc[r]);
          }
      }
    // all target values are consumed BEFORE the first store: with loads and stores pending together the compiler's
    // s_waitcnt bookkeeping (gfx9: one vmcnt, loads and stores may retire out of order) falls back to draining everything
    float vout[2][4], dif[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vout[r][j] = acc[r][j] + bj[j];
        dif[r][j] = vout[r][j] - tg[r][j];
        asm volatile("" : "+v"(dif[r][j]));     // materialise here (the optimiser would sink it below the stores)
      }
    if (a.nonfinite) {      // exponent all ones = inf / NaN in an output value of this lane (the flag is raised at the end of the kernel)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) bad |= (inimg[r] && j < a.C && (__float_as_uint(vout[r][j]) & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next tile's registers go to the other LDS buffer here - after the MFMAs (its loads had the whole step to land)
    // and before the stores (a wait issued behind them would have to drain them)
    if (cur_tile + stride < ntiles) halo_write(Rnext, other, tid);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 2; ++r) {   // D rows 0..3 (= output channels) live in lanes 0..15
      float sg[4] = {0.f, 0.f, 0.f, 0.f};
      if (inimg[r]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < a.C) {
            a.out[obase[r] + (size_t)j * a.H * a.W] = vout[r][j];
            if (a.target) {
              const float d = dif[r][j];
              lsum += fabsf(d);
              sg[j] = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
            }
          }
        }
        if (a.dy4) {
          const int y = tc.ty * TH + 2 * wave + r;
          *reinterpret_cast<uint2*>(a.dy4 + ((size_t)(tc.n * a.H + y) * a.W + xx) * 4) = pack4_bf16(sg[0], sg[1], sg[2], sg[3]);
        }
      }
      if (WGRAD && g == 0) {          // the tile's sign gradient into its three shifted copies (zero outside the image), and its per-channel sums (bias gradient)
        const uint2 d4 = pack4_bf16(sg[0], sg[1], sg[2], sg[3]);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) *reinterpret_cast<uint2*>(ldy + kx * DYC + ((2 * wave + r) * HALO_W + px + kx) * 8) = d4;
#pragma unroll
        for (int j = 0; j < 4; ++j) bsum[j] += sg[j];
      }
    }
    if (WGRAD) {
      __syncthreads();                // the whole tile's sign gradient is in LDS
      const unsigned xb = (unsigned)(size_t)(tail_lds_u8)cur, yb = (unsigned)(size_t)(tail_lds_u8)ldy;
      const int qq = (lane >> 2) & 3, p4 = lane & 3;
      const unsigned chan = (unsigned)((2 * wave + (p4 >> 1)) * 16 + (p4 & 1) * 8);
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {          // 32 halo-index entries per MFMA K step
        const int h = 32 * ks + 16 * (g >> 1) + 4 * (g & 1) + qq;
        // A rows = (kx = p4, channel): lanes p4 < 3 read their copy, p4 = 3 the zero bytes
        const bf16x8 A = tail_tr8(yb + ((p4 < 3) ? (unsigned)(p4 * DYC + h * 8) : (unsigned)(3 * DYC)), (p4 < 3) ? 64u : 0u);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          // B = input pixels ky * 18 + h (and + 8); past the halo tile (last K step only, where A is zero) any finite pixel will do
          int i1 = ky * HALO_W + h, i2 = i1 + 8;
          if (ks == 4) { i1 = i1 < HALO_PIX ? i1 : HALO_PIX - 1; i2 = i2 < HALO_PIX ? i2 : HALO_PIX - 1; }
          const bf16x8 B = tail_tr8(xb + (unsigned)i1 * PIX_STRIDE + chan, (unsigned)(i2 - i1) * PIX_STRIDE);
          wacc[WGRAD ? ky : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, wacc[WGRAD ? ky : 0], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  };
  for (; tile < ntiles; tile += 2 * stride) {
    step(tile, lds, R[1], R[0], lds + X_STAGE_BYTES);
    if (tile + stride < ntiles) step(tile + stride, lds + X_STAGE_BYTES, R[0], R[1], lds);
  }
  if (WGRAD) {
    // slab of this workgroup: [co 16][tap 9][ci 64] + [16] bias sums
    float* slab = a.wslab + (size_t)blockIdx.x * (16 * 576 + 16);
    if (g < 3) {                       // D rows 4 g + e = (kx = g, channel e)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int e = 0; e < 4; ++e) slab[(e * 9 + ky * 3 + g) * 64 + 16 * wave + px] = wacc[WGRAD ? ky : 0][e];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = bsum[j];
      t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4); t += __shfl_xor(t, 8);
      if (lane == 0) redb[wave][j] = t;
    }
    __syncthreads();
    if (tid < 4) slab[16 * 576 + tid] = (redb[0][tid] + redb[1][tid]) + (redb[2][tid] + redb[3][tid]);
  }
  if (a.loss_partial) {
    float s = lsum;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) a.loss_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
  if (a.nonfinite && __any(bad != 0u) && lane == 0) atomicOr(a.nonfinite, 1u);
}

__global__ void loss_finalize_kernel(const float* partial, int n, float inv_numel, float* loss) {
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

// ------------------------------------------------------------------------------------------------------
// tail dgrad: dy4 [N,H,W,4] bf16 -> dx [N,H,W,64] bf16.  K = 9 taps x 4 channels = 36, padded to 64 (two MFMA
// k-steps); B-fragment element e of lane group g is (tap = 8*ks + 2g + e/4, channel e%4), i.e. two 8-byte LDS
// reads per k-step.  Write-bound (the output is the largest gradient tensor).
// ------------------------------------------------------------------------------------------------------
typedef unsigned int tail_u32x4 __attribute__((ext_vector_type(4)));
struct TailDgradDev { const uint16_t* dy4; const uint4* w; uint16_t* dx; int N, H, W, tiles_x, tiles_y; };

__global__ void __launch_bounds__(256, 2) tail_dgrad_kernel(TailDgradDev a) {
  __shared__ __attribute__((aligned(16))) uint2 sdy[HALO_PIX + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const TileCoord tc = decode_tile(blockIdx.x, a.tiles_x, a.tiles_y);    // (XCD-contiguous order: 5 us slower for this write-only kernel)
  if (tid < HALO_PIX) {
    const int r = tid / HALO_W, c = tid - r * HALO_W;
    const int y = tc.ty * TH + r - 1, x = tc.tx * TW + c - 1;
    uint2 v = make_uint2(0, 0);
    if (y >= 0 && y < a.H && x >= 0 && x < a.W)
      v = *reinterpret_cast<const uint2*>(a.dy4 + ((size_t)(tc.n * a.H + y) * a.W + x) * 4);
    sdy[tid] = v;
  }
  const bf16x8 F0 = as_bf16x8(a.w[(wave * 2 + 0) * 64 + lane]);
  const bf16x8 F1 = as_bf16x8(a.w[(wave * 2 + 1) * 64 + lane]);
  __syncthreads();
  const int c0 = 16 * wave + 4 * g;
  // taps of this lane group: ks=0 -> 2g, 2g+1 ; ks=1 -> 8+2g (only g==0 real), 9+2g (never)
  const int ta = 2 * g, tb = 2 * g + 1;
  const int kya = ta / 3, kxa = ta - 3 * kya, kyb = tb / 3, kxb = tb - 3 * kyb;
  // The MFMA leaves a lane 4 channels of a pixel: stored directly, the 151 MB output would leave as 8-byte pieces of lines that four waves
  // assemble in L2.  The tile goes through LDS instead (16 KB, 16-byte chunk index XOR pixel & 7) and leaves as whole 128-byte lines,
  // 2 KB contiguous per tile row, non-temporal (conv_block.hip: -2 us per launch there; here 42 -> see DESIGN.md 4).
  __shared__ __attribute__((aligned(16))) unsigned char sout[TH * TW * 128];
#pragma unroll
  for (int r = 0; r < TH; ++r) {
    union { uint2 u[2]; bf16x8 b; } B0, B1;
    B0.u[0] = sdy[(r + kya) * HALO_W + px + kxa];
    B0.u[1] = sdy[(r + kyb) * HALO_W + px + kxb];
    B1.u[0] = (g == 0) ? sdy[(r + 2) * HALO_W + px + 2] : make_uint2(0, 0);
    B1.u[1] = make_uint2(0, 0);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F0, B0.b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F1, B1.b, acc, 0, 0, 0);
    const int pix = r * TW + px;
    *reinterpret_cast<uint2*>(sout + pix * 128 + (((c0 >> 3) ^ (pix & 7)) << 4) + (g & 1) * 8) = pack4_bf16(acc[0], acc[1], acc[2], acc[3]);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TH * TW * 8 / 256; ++i) {
    const int p = tid + 256 * i, pix = p >> 3, c = p & 7;
    const int y = tc.ty * TH + (pix >> 4), x = tc.tx * TW + (pix & 15);
    if (y < a.H && x < a.W) {
      const uint4 v = *reinterpret_cast<const uint4*>(sout + pix * 128 + ((c ^ (pix & 7)) << 4));
      __builtin_nontemporal_store((tail_u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<tail_u32x4*>(a.dx + ((size_t)(tc.n * a.H + y) * a.W + x) * 64 + c * 8));
    }
  }
}

__global__ void nchw_to_nhwc4_kernel(const float* src, uint16_t* dst, int N, int C, int H, int W) {
  const size_t total = (size_t)N * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = i % ((size_t)H * W);
    const size_t n = i / ((size_t)H * W);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) v[c] = src[(n * C + c) * (size_t)H * W + hw];
    *reinterpret_cast<uint2*>(dst + i * 4) = pack4_bf16(v[0], v[1], v[2], v[3]);
  }
}

// ------------------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------------------
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
int rumpy_conv3x3_strip_launch(const rumpy_conv_args* p, hipStream_t s);   // conv_strip.hip
int rumpy_conv_up_launch(const rumpy_conv_args* p, hipStream_t s);          // conv_up.hip

int rumpy_conv4d_launch(ConvDev d, int grid_x, int cap, hipStream_t s, int fmt);      // conv_dgrad4.hip
extern "C" int rumpy_conv3x3(const rumpy_conv_args* p, void* stream) {
  if (!p || !p->x || !p->w || !p->out) { rumpy_set_error("rumpy_conv3x3: null pointer"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->cout_tiles <= 0) { rumpy_set_error("rumpy_conv3x3: bad shape"); return RUMPY_E_ARG; }
  if (p->cin_chunks < 1 || p->cin_chunks > 4) { rumpy_set_error("rumpy_conv3x3: cin_chunks must be 1 .. 4 (got %d)", p->cin_chunks); return RUMPY_E_ARG; }
  if (p->in_mode == 1 && p->cin_chunks != 4) { rumpy_set_error("rumpy_conv3x3: in_mode 1 needs cin_chunks 4"); return RUMPY_E_ARG; }
  // out_mode 1 (PixelShuffle fused into the store): no mask; a residual operand only in the Cin = 64 form (conv_up.hip reads it in
  // conv-output order [N, H, W, 256] - the rounding-residual launch of an fp16 evaluation plan's upsampler stage)
  const bool shuffled_res = p->out_mode == 1 && p->res1 && !p->res2 && p->cin_chunks == 1 && p->in_mode == 0 && !p->pool;
  if (p->out_mode == 1 && (p->cout_tiles != 4 || p->mask || ((p->res1 || p->res2) && !shuffled_res))) {
    rumpy_set_error("rumpy_conv3x3: out_mode 1 needs cout_tiles 4, no mask and no residual (one residual operand with cin_chunks 1)"); return RUMPY_E_ARG; }
  if (p->fmt != RUMPY_FMT_BF16 && !(p->fmt == RUMPY_FMT_F16 && !p->mask && p->in_mode == 0)) {
    rumpy_set_error("rumpy_conv3x3: fmt %d needs a plain input and no mask (forward launches of an evaluation plan)", p->fmt); return RUMPY_E_ARG; }
  // Cin = 256 (upsampler data gradients): the 8x16-tile kernel below measures faster than the 4-chunk strip build
  // (which spills at 256 VGPRs); RUMPY_CONV4_STRIP=1 selects the strip build for A/B runs.
  static const bool strip4 = getenv("RUMPY_CONV4_STRIP") != nullptr;
  if (p->cin_chunks == 1 || (strip4 && p->cin_chunks == 4)) {   // strip kernel (conv_strip.hip)
    hipStream_t s1 = (hipStream_t)stream;
    const int kid1 = (p->cout_tiles == 1 && p->cin_chunks == 1) ? 1 : 3;
    // (every argument check sits above the probe's start event: a refused call leaves no unbalanced record)
    const bool up_ok = p->cin_chunks == 1 && p->in_mode == 0 && !p->mask && !p->pool && (p->out_mode == 0 || (!p->res1 && !p->res2));
    if (p->w_lo && !(p->fmt == RUMPY_FMT_F16 && (shuffled_res || up_ok))) {
      rumpy_set_error("rumpy_conv3x3: w_lo goes with fmt F16 and a Cin = 64 forward launch without mask / pool (conv_up.hip)"); return RUMPY_E_ARG; }
    rumpy_probe_pre(kid1, s1);
    // plain forward convs with several output tiles (the upsampler convs): conv_up.hip (output through LDS as whole non-temporal lines)
    const bool up_old = getenv("RUMPY_UP_OLD") != nullptr;             // A/B switch (read per call: the tests toggle it)
    // conv_up.hip (output through LDS as whole non-temporal lines) takes every Cin = 64 launch without ReLU mask / pool sums that runs more
    // than one strip per workgroup or several output tiles: the upsampler convs, and all plain layers of a whole-image evaluation
    const int up_strips = p->N * cdiv(p->H, 6) * cdiv(p->W, 48);
    const char* up_force = getenv("RUMPY_UP_FORCE");                    // diagnostic (kbench.py up1): "1" = every eligible launch, "0" = multi-tile ones only
    const bool up_many = up_force ? up_force[0] == '1' : up_strips > rumpy_device_cus();
    if (shuffled_res || p->w_lo || (!up_old && (p->cout_tiles > 1 || up_many) && up_ok))
      rumpy_conv_up_launch(p, s1);
    else
    rumpy_conv3x3_strip_launch(p, s1);
    rumpy_probe_post(kid1, s1);
    return rumpy_check_launch("rumpy_conv3x3");
  }
  if (p->w_lo) { rumpy_set_error("rumpy_conv3x3: w_lo goes with cin_chunks 1"); return RUMPY_E_ARG; }
  ConvDev d;
  d.x = (const uint16_t*)p->x; d.w = (const uint4*)p->w; d.bias = p->bias; d.out = (uint16_t*)p->out;
  d.mask = (const uint16_t*)p->mask; d.res1 = (const uint16_t*)p->res1; d.res2 = (const uint16_t*)p->res2; d.pool = p->pool;
  d.N = p->N; d.H = p->H; d.W = p->W; d.cout_tiles = p->cout_tiles; d.in_mode = p->in_mode; d.out_mode = p->out_mode;
  d.relu = p->relu; d.scale = p->scale; d.tiles_x = cdiv(p->W, TW); d.tiles_y = cdiv(p->H, TH);
  const int ntiles = d.N * d.tiles_x * d.tiles_y;
  int gx = p->grid_x;
  if (gx <= 0) {
    const int slots = (p->cin_chunks == 1 ? 2 : 1) * rumpy_device_cus();
    const int per_ct = slots / p->cout_tiles > 0 ? slots / p->cout_tiles : 1;
    const int rounds = cdiv(ntiles, per_ct);       // balance: every workgroup gets `rounds` or rounds-1 tiles
    gx = cdiv(ntiles, rounds);
  }
  if (gx > ntiles) gx = ntiles;
  hipStream_t s = (hipStream_t)stream;
  const int kid = 3;
  rumpy_probe_pre(kid, s);
  dim3 grid(gx, p->cout_tiles);
  // the data gradients of the upsampler convs (what the engine launches with Cin = 256)
  const bool old4 = getenv("RUMPY_CONV4_OLD") != nullptr;             // A/B switch (read per call: tests toggle it)
  // streaming form (conv_dgrad4.hip): every Cin = 256 launch that writes the plain layout without pool sums - the upsampler's data gradients
  // and the body convs of the wide EDSR; RUMPY_CONV4_OLD=1 keeps the register-staged kernel below (A/B, tests)
  const bool plain = p->out_mode == 0 && !p->pool;
  if (plain && !old4 && p->cin_chunks == 4) {
    rumpy_conv4d_launch(d, p->grid_x, rumpy_device_cus() / p->cout_tiles > 0 ? rumpy_device_cus() / p->cout_tiles : 1, s, p->fmt);
  } else if (p->fmt == RUMPY_FMT_F16) {      // evaluation plans of the wide nets
    if (p->cin_chunks == 4) hipLaunchKernelGGL((conv3x3_kernel<4, RUMPY_FMT_F16>), grid, dim3(256), 0, s, d);
    else if (p->cin_chunks == 3) hipLaunchKernelGGL((conv3x3_kernel<3, RUMPY_FMT_F16>), grid, dim3(256), 0, s, d);
    else hipLaunchKernelGGL((conv3x3_kernel<2, RUMPY_FMT_F16>), grid, dim3(256), 0, s, d);
  } else if (p->cin_chunks == 4)
  hipLaunchKernelGGL(conv3x3_kernel<4>, grid, dim3(256), 0, s, d);
  else if (p->cin_chunks == 3) hipLaunchKernelGGL(conv3x3_kernel<3>, grid, dim3(256), 0, s, d);      // 192 / 128 features (EDSR widths between the
  else hipLaunchKernelGGL(conv3x3_kernel<2>, grid, dim3(256), 0, s, d);                               // baseline's 64 and the shipped 256)
  rumpy_probe_post(kid, s);
  return rumpy_check_launch("rumpy_conv3x3");
}

// sum of the per-workgroup slabs of the fused tail weight gradient: block = 16 elements x 16 slab groups; group p adds slabs
// p, p+16, .. (independent loads), the 16 partial sums are added in a fixed order
__global__ void __launch_bounds__(256) tail_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslabs, int C, float scale,
                                                                float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ float part[16][17];
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int nelem = C * 576;
  const int e = blockIdx.x * 16 + el;                    // e < nelem: weight (co, tap, ci); nelem <= e < nelem + C: bias co
  const int stride = 16 * 576 + 16;
  float s0 = 0.f, s1 = 0.f;
  if (e < nelem + C) {
    const int src = (e < nelem) ? e : 16 * 576 + (e - nelem);
    int k = grp;
    for (; k + 16 < nslabs; k += 32) { s0 += slabs[(size_t)k * stride + src]; s1 += slabs[(size_t)(k + 16) * stride + src]; }
    if (k < nslabs) s0 += slabs[(size_t)k * stride + src];
  }
  part[grp][el] = s0 + s1;
  __syncthreads();
  if (grp == 0 && e < nelem + C) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += part[i][el];
    s *= scale;
    if (e < nelem) {
      const int co = e / 576, rem = e - co * 576, tap = rem >> 6, ci = rem & 63;
      gw[((size_t)co * 64 + ci) * 9 + tap] = s;
    } else gb[e - nelem] = s;
  }
}

extern "C" int rumpy_tail_wgrad_reduce(const float* wslab, int32_t nslabs, int32_t C, float scale, float* gw, float* gb, void* stream) {
  if (!wslab || !gw || !gb || nslabs <= 0 || C < 1 || C > 4) { rumpy_set_error("rumpy_tail_wgrad_reduce: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(tail_wgrad_reduce_kernel, dim3((C * 576 + C + 15) / 16), dim3(256), 0, (hipStream_t)stream, wslab, nslabs, C, scale, gw, gb);
  return rumpy_check_launch("rumpy_tail_wgrad_reduce");
}

extern "C" int rumpy_tail_fwd_grid(int32_t N, int32_t H, int32_t W, int32_t grid_x) {
  const int ntiles = N * cdiv(W, TW) * cdiv(H, TH);
  int gx = grid_x > 0 ? grid_x : 2 * rumpy_device_cus();
  return gx > ntiles ? ntiles : gx;
}

extern "C" int rumpy_tail_fwd(const rumpy_tail_fwd_args* p, void* stream) {
  if (!p || !p->x || !p->w || !p->bias || !p->out) { rumpy_set_error("rumpy_tail_fwd: null pointer"); return RUMPY_E_ARG; }
  if (p->C < 1 || p->C > 4 || p->N <= 0 || p->H <= 0 || p->W <= 0) { rumpy_set_error("rumpy_tail_fwd: bad shape"); return RUMPY_E_ARG; }
  if ((p->dy4 || p->loss_partial || p->loss) && !p->target) { rumpy_set_error("rumpy_tail_fwd: loss outputs need target"); return RUMPY_E_ARG; }
  if (p->target && (!p->loss_partial || !p->loss)) { rumpy_set_error("rumpy_tail_fwd: target needs loss_partial and loss"); return RUMPY_E_ARG; }
  TailDev d;
  d.x = (const uint16_t*)p->x; d.w = (const uint4*)p->w; d.bias = p->bias; d.out = p->out; d.target = p->target; d.target_ind = p->target ? p->target_ind : nullptr;
  d.dy4 = (uint16_t*)p->dy4; d.loss_partial = p->loss_partial; d.wslab = p->wslab; d.N = p->N; d.C = p->C; d.H = p->H; d.W = p->W;
  if (p->wslab && !p->target) { rumpy_set_error("rumpy_tail_fwd: wslab needs target"); return RUMPY_E_ARG; }
  if (p->fmt != RUMPY_FMT_BF16 && !(p->fmt == RUMPY_FMT_F16 && !p->wslab && !p->dy4)) { rumpy_set_error("rumpy_tail_fwd: fmt %d goes without dy4 / wslab", p->fmt); return RUMPY_E_ARG; }
  d.nonfinite = p->nonfinite;
  d.tiles_x = cdiv(p->W, TW); d.tiles_y = cdiv(p->H, TH);
  const int ntiles = d.N * d.tiles_x * d.tiles_y;
  int gx = p->grid_x > 0 ? p->grid_x : 2 * rumpy_device_cus();
  if (gx > ntiles) gx = ntiles;
  hipStream_t s = (hipStream_t)stream;
  if (p->wslab) hipLaunchKernelGGL(tail_fwd_kernel<true>, dim3(gx), dim3(256), 0, s, d);
  else if (p->fmt == RUMPY_FMT_F16) hipLaunchKernelGGL((tail_fwd_kernel<false, RUMPY_FMT_F16>), dim3(gx), dim3(256), 0, s, d);
  else hipLaunchKernelGGL(tail_fwd_kernel<false>, dim3(gx), dim3(256), 0, s, d);
  if (p->target) {
    const float inv = 1.0f / ((float)p->N * p->C * p->H * p->W);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, p->loss_partial, gx, inv, p->loss);
  }
  return rumpy_check_launch("rumpy_tail_fwd");
}

extern "C" int rumpy_tail_dgrad(const rumpy_tail_dgrad_args* p, void* stream) {
  if (!p || !p->dy4 || !p->w || !p->dx || p->N <= 0 || p->H <= 0 || p->W <= 0) { rumpy_set_error("rumpy_tail_dgrad: bad argument"); return RUMPY_E_ARG; }
  TailDgradDev d;
  d.dy4 = (const uint16_t*)p->dy4; d.w = (const uint4*)p->w; d.dx = (uint16_t*)p->dx; d.N = p->N; d.H = p->H; d.W = p->W;
  d.tiles_x = cdiv(p->W, TW); d.tiles_y = cdiv(p->H, TH);
  hipLaunchKernelGGL(tail_dgrad_kernel, dim3(d.N * d.tiles_x * d.tiles_y), dim3(256), 0, (hipStream_t)stream, d);
  return rumpy_check_launch("rumpy_tail_dgrad");
}

extern "C" int rumpy_nchw_to_nhwc4(const rumpy_nchw_to_nhwc4_args* p, void* stream) {
  if (!p || !p->src || !p->dst || p->C < 1 || p->C > 4 || p->N <= 0 || p->H <= 0 || p->W <= 0) { rumpy_set_error("rumpy_nchw_to_nhwc4: bad argument"); return RUMPY_E_ARG; }
  const size_t total = (size_t)p->N * p->H * p->W;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p->src, (uint16_t*)p->dst, p->N, p->C, p->H, p->W);
  return rumpy_check_launch("rumpy_nchw_to_nhwc4");
}

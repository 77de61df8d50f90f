// The upsampler convolutions (64 -> 256 + PixelShuffle, common.py:23-48 Upsampler) as their own kernel: conv3x3_strip_kernel's work split
// (strip of 6 x 48 output pixels, wave (q, rh) = 16 output channels x 3 rows, filter slice stationary over a persistent loop of strips,
// next strip's input prefetched through registers into the other LDS stage) on the block kernels' LDS image (unpadded 128-byte pixels, 16-byte
// chunk index XOR pixel & 7: 51 KB per stage instead of 77) - which leaves room for the OUTPUT image of a strip (38 KB): the epilogue writes
// its packed values there and the strip leaves as whole 128-byte lines with non-temporal stores, one piece after every third MFMA group of
// the NEXT strip's sweep (block_common.hpp::strip_stage; conv_block.hip).  The MFMA epilogue's own layout would store 32-byte pieces of 32
// lines per wave-instruction: 1.3 TB/s on the 151 MB of the second upsampler conv (tests/tools/kbench.py up), with or without the shuffle.
#define BLOCK_PRIO 0       // no raised priority for this kernel's sweeps (block_common.hpp, round 6): the stores of the previous strip ride inside them - 54.6 against 51.4 us with it
#include "block_common.hpp"

struct UpDev {
  const uint16_t* x; const uint4* w; const uint4* w_lo; const float* bias; uint16_t* out;      // w_lo: NULL, or the filter's rounding-residual image (fp16 evaluation plans)
  const uint16_t* res1; const uint16_t* res2;      // optional residual operands, layout of out (out_mode 0)
  int N, H, W, cout_tiles, out_mode, sx_n, sy_n, relu; float scale;
};
constexpr int UPROWS = BSH + 2;                          // 8 input rows
constexpr int UPSTAGE = UPROWS * BCOLS * 128;           // 51200
constexpr int UPOUT = BSH * BCOLS * 128;                // 38400: output image, rows 0 .. 5, columns 1 .. 48 used
constexpr int UPPIECES = UPROWS * BCOLS * 8;            // 3200
constexpr int UPREGS = (UPPIECES + BTHREADS - 1) / BTHREADS;   // 7

struct UpCoord { int n, sy, sx; };
__device__ __forceinline__ UpCoord up_decode(int s, int sx_n, int sy_n) {
  UpCoord c;
  c.sx = s % sx_n;
  const int r = s / sx_n;
  c.sy = r % sy_n;
  c.n = r / sy_n;
  return c;
}
__device__ __forceinline__ void up_issue(uint4 (&R)[UPREGS], const uint16_t* __restrict__ x, UpCoord c, int H, int W, int tid) {
  const int y0 = c.sy * BSH - 1, x0 = c.sx * BSW - 1;
#pragma unroll
  for (int i = 0; i < UPREGS; ++i) {
    const int p = tid + BTHREADS * i;
    const int pix = p >> 3, part = p & 7;
    const int lr = pix / BCOLS, lc = pix - lr * BCOLS;
    const int y = y0 + lr, xx = x0 + lc;
    const bool ok = (p < UPPIECES) & ((unsigned)y < (unsigned)H) & ((unsigned)xx < (unsigned)W);
    const int e = ok ? ((c.n * H + y) * W + xx) * 64 + part * 8 : 0;
    uint4 v = *reinterpret_cast<const uint4*>(x + (unsigned)e);      // unconditional (clamped address): the waits stay counted
    if (!ok) v = make_uint4(0, 0, 0, 0);
    R[i] = v;
  }
}
__device__ __forceinline__ void up_write(const uint4 (&R)[UPREGS], unsigned char* stage, int tid) {
#pragma unroll
  for (int i = 0; i < UPREGS; ++i) {
    const int p = tid + BTHREADS * i;
    if (p < UPPIECES) *reinterpret_cast<uint4*>(stage + swz(p >> 3, p & 7)) = R[i];
  }
}
// element offset of piece i of strip c in the output tensor (0xffffffff outside the image): plain [N,H,W,64*tiles] or pixel-shuffled [N,2H,2W,64]
__device__ __forceinline__ unsigned up_piece_off(int i, int tid, UpCoord c, int ct, const UpDev& a) {
  const int p = tid + BTHREADS * i, pix = p >> 3, r = pix / BSW, col = pix - r * BSW;
  const int y = c.sy * BSH + r, xx = c.sx * BSW + col;
  if (!(p < STRIP_PIECES && y < a.H && xx < a.W)) return 0xffffffffu;
  if (a.out_mode == 0) return (unsigned)(((c.n * a.H + y) * a.W + xx) * (64 * a.cout_tiles) + ct * 64 + (p & 7) * 8);
  return (unsigned)(((c.n * 2 * a.H + 2 * y + (ct >> 1)) * (2 * a.W) + 2 * xx + (ct & 1)) * 64 + (p & 7) * 8);
}

// the same for the 3 rows of a row half (block_common.hpp::group_stage): piece p = tg + 256 i
__device__ __forceinline__ unsigned up_group_off(int i, int tg, int rh, UpCoord c, int ct, const UpDev& a) {
  const int p = tg + 256 * i, pix = p >> 3, r = pix / BSW, col = pix - r * BSW;
  const int y = c.sy * BSH + 3 * rh + r, xx = c.sx * BSW + col;
  if (!(p < GROUP_PIECES && y < a.H && xx < a.W)) return 0xffffffffu;
  if (a.out_mode == 0) return (unsigned)(((c.n * a.H + y) * a.W + xx) * (64 * a.cout_tiles) + ct * 64 + (p & 7) * 8);
  return (unsigned)(((c.n * 2 * a.H + 2 * y + (ct >> 1)) * (2 * a.W) + 2 * xx + (ct & 1)) * 64 + (p & 7) * 8);
}

// TWO: two filter images per launch (a.w_lo: fp16 evaluation plans; its own instantiation - the training kernel's registers stay as they were)
template <int FMT, bool TWO = false>
__global__ void __launch_bounds__(BTHREADS, 2) conv_up_kernel(UpDev a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * UPSTAGE + UPOUT];
  unsigned char* const ldo = lds + 2 * UPSTAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = wave >> 2;
  const int ct = blockIdx.y;
  const int nstrips = a.N * a.sy_n * a.sx_n;
  int strip = xcd_strip(blockIdx.x, gridDim.x);
  if (strip >= nstrips) return;
  uint4 R[UPREGS];
  up_issue(R, a.x, up_decode(strip, a.sx_n, a.sy_n), a.H, a.W, tid);
  bf16x8 F[18];
  // Two filter images per launch (a.w_lo, round 3): every strip is swept with the rounding-residual image first and with the filter
  // second, into the same fp32 accumulators - the slice in the registers is swapped twice per strip (18 L2 hits per wave, requested right
  // behind the sweep that read the old slice: they land under the partner wave's MFMAs / this wave's epilogue).
  constexpr bool two = TWO;
  auto fetch_filter = [&](const uint4* img) {
    asm volatile("" : "+s"(img));         // opaque: the two images are re-read every strip ON PURPOSE (hoisted out of the loop they would need 144 registers)
    const uint4* wp = img + ((size_t)(ct * 4 + q) * 18) * 64 + lane;
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
  };
  fetch_filter(two ? a.w_lo : a.w);
  const int c0 = 16 * q + 4 * g;
  const int gpair = 4 * (g & ~1);
  const int chunk8 = 2 * q + (gpair >> 3);
  f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (a.bias) { const float4 b4 = *reinterpret_cast<const float4*>(a.bias + ct * 64 + c0); bias4 = (f32x4){b4.x, b4.y, b4.z, b4.w}; }
  __shared__ unsigned gate[5];            // 0: waves that have written their share of an input stage; 1: waves that have finished a sweep;
  if (tid < 5) gate[tid] = 0u;            // 2 + rh: waves of a row half that have written their output rows; 4: (spare)
  __shared__ unsigned gate_rd[2];         // waves of a row half that have read their pieces of the output image back
  if (tid < 2) gate_rd[tid] = 0u;
  up_write(R, lds, tid);
  __syncthreads();                        // first stage + zeroed counters (the only workgroup barrier of the kernel)
  const int rhu = __builtin_amdgcn_readfirstlane(rh), tg = tid & 255;
  // the loads of the strip after the next one are issued BEFORE the stores of the current one (vector-memory operations return in issue
  // order: a stage write must not wait behind 5 non-temporal stores).  Measured neutral: what an iteration really costs beside its 162 MFMAs
  // per wave is the address arithmetic of loads / stores / LDS images (about 600 VALU instructions per wave and strip, issued beside the
  // partner wave's MFMAs: stamps of tests/tools/kbench.py up with a round-2 stamp build)
  {
    const int s2 = strip + (int)gridDim.x;
    up_issue(R, a.x, up_decode(s2 < nstrips ? s2 : strip, a.sx_n, a.sy_n), a.H, a.W, tid);
  }

  // Persistent loop WITHOUT workgroup barriers: the two row halves (waves 0-3 / 4-7) drift half an iteration apart - the older waves win the
  // matrix pipe, finish their sweep first and run their stage write / epilogue / stores under the other half's sweep (block_common.hpp:
  // row-half groups).  Every wait is for exactly what the next step touches:
  //   input stage of strip k complete (all 8 waves wrote their pieces)          -> gate[0] >= 8 k        (k counts strips of this workgroup from 1)
  //   both halves done with the sweep that read the stage about to be refilled  -> gate[1] >= 8 (k - 1)
  //   this half's output rows of strip k complete                               -> gate[2 + rh] >= 4 k
  //   this half's pieces of strip k - 1 read back (its rows may be overwritten) -> gate_rd[rh] >= 4 (k - 1)
  int buf = 0;
  unsigned k = 1;
  for (; strip < nstrips; strip += gridDim.x, ++k) {
    const UpCoord sc = up_decode(strip, a.sx_n, a.sy_n);
    const bool has_next = strip + (int)gridDim.x < nstrips;
    f32x4 acc[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = bias4;
    unsigned off[8][2];
    sweep_bases(off, (unsigned)(buf * UPSTAGE), 3 * rh, px, g);
    if (k > 1) gate_wait(&gate[0], 8u * (k - 1));          // (the first stage is behind the workgroup barrier)
    block_sweep<3, FMT>(acc, F, lds, off);
    if (two) {
      __builtin_amdgcn_sched_barrier(0);                   // (the new slice goes INTO the registers of the old one: not hoisted above the sweep's last reads)
      fetch_filter(a.w);                                   // residual image done: the filter itself, same accumulators
      __builtin_amdgcn_sched_barrier(0);
      block_sweep<3, FMT>(acc, F, lds, off);
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) fetch_filter(a.w_lo);                  // for the next strip: lands under the epilogue below
      __builtin_amdgcn_sched_barrier(0);
    }
    gate_arrive(&gate[1], lane);
    if (has_next) {
      gate_wait(&gate[1], 8u * (k - 1));                   // nobody sweeps over the other stage any more
      up_write(R, lds + (buf ^ 1) * UPSTAGE, tid);
      gate_arrive(&gate[0], lane);
    }
    {   // the strip after the next one (past the end: this strip again, unused): issued on every path, before this strip's stores
      const int s2 = strip + 2 * (int)gridDim.x;
      up_issue(R, a.x, up_decode(s2 < nstrips ? s2 : strip, a.sx_n, a.sy_n), a.H, a.W, tid);
    }
    // epilogue: the packed values go to this half's rows of the output image (pairs k < 3: (row k, col tile 0 | 1); 3: (rows 0 | 1, col tile 2); single: (2, 2))
    gate_wait(&gate_rd[rhu], 4u * (k - 1));
    const bool post = a.relu || a.scale != 1.0f || a.res1 || a.res2;      // uniform: the upsampler convs skip all of it
    auto own = [&](float x) -> float { if (a.relu) x = relu_f32(x); return x * a.scale; };
    auto res_off = [&](int r, int xx, int ch) -> unsigned {                 // element offset in a residual tensor, 0xffffffff outside the image
      const int y = sc.sy * BSH + 3 * rh + r, xg = sc.sx * BSW + xx;
      return (y < a.H && xg < a.W) ? (unsigned)(((sc.n * a.H + y) * a.W + xg) * (64 * a.cout_tiles) + ct * 64 + ch) : 0xffffffffu;
    };
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 tx = (kk < 3) ? acc[kk < 3 ? kk : 0][0] : acc[0][2];
      const f32x4 ty = (kk < 3) ? acc[kk < 3 ? kk : 0][1] : acc[1][2];
      float v[8];
      pair_up(tx, ty, g, v);
      const int r = (kk < 3) ? kk : (g & 1), c = (kk < 3) ? (g & 1) : 2;
      if (post) {
        const unsigned o = res_off(r, 16 * c + px, 16 * q + gpair);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = own(v[j]);
        if (o != 0xffffffffu) {
          float m[8];
          if (a.res1) {
            unpack8<FMT>(*reinterpret_cast<const uint4*>(a.res1 + o), m);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += m[j];
          }
          if (a.res2) {
            unpack8<FMT>(*reinterpret_cast<const uint4*>(a.res2 + o), m);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += m[j];
          }
        }
      }
      const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
      *reinterpret_cast<uint4*>(ldo + swz((3 * rh + r) * BCOLS + 16 * c + px + 1, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    {
      float v[4] = {acc[2][2][0], acc[2][2][1], acc[2][2][2], acc[2][2][3]};
      if (post) {
        const unsigned o = res_off(2, 32 + px, c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = own(v[j]);
        if (o != 0xffffffffu) {
          float m[4];
          if (a.res1) {
            unpack4<FMT>(*reinterpret_cast<const uint2*>(a.res1 + o), m);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += m[j];
          }
          if (a.res2) {
            unpack4<FMT>(*reinterpret_cast<const uint2*>(a.res2 + o), m);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += m[j];
          }
        }
      }
      *reinterpret_cast<uint2*>(ldo + swz((3 * rh + 2) * BCOLS + 32 + px + 1, 2 * q + (g >> 1)) + (g & 1) * 8) = pack4<FMT>(v[0], v[1], v[2], v[3]);
    }
    gate_arrive(&gate[2 + rhu], lane);
    gate_wait(&gate[2 + rhu], 4u * k);
    {
      uint4 S[GROUP_REGS];
      group_stage<0>(S, ldo, tg, rhu);
#pragma unroll
      for (int i = 0; i < GROUP_REGS; ++i) {
        const unsigned o = up_group_off(i, tg, rhu, sc, ct, a);
        if (o != 0xffffffffu) st16_nt(a.out + o, S[i]);      // whole lines, non-temporal; they drain under the sweeps
      }
    }
    gate_arrive(&gate_rd[rhu], lane);     // (release: behind the reads above)
    buf ^= 1;
  }
}

int rumpy_conv_up_launch(const rumpy_conv_args* p, hipStream_t s) {
  UpDev d;
  d.x = (const uint16_t*)p->x; d.w = (const uint4*)p->w; d.w_lo = (const uint4*)p->w_lo; d.bias = p->bias; d.out = (uint16_t*)p->out;
  d.res1 = (const uint16_t*)p->res1; d.res2 = (const uint16_t*)p->res2; d.relu = p->relu; d.scale = p->scale;
  d.N = p->N; d.H = p->H; d.W = p->W; d.cout_tiles = p->cout_tiles; d.out_mode = p->out_mode;
  d.sx_n = (p->W + BSW - 1) / BSW; d.sy_n = (p->H + BSH - 1) / BSH;
  const int nstrips = d.N * d.sx_n * d.sy_n;
  int gx = p->grid_x;
  if (gx <= 0) {
    const int slots = rumpy_device_cus() / p->cout_tiles > 0 ? rumpy_device_cus() / p->cout_tiles : 1;
    const int rounds = (nstrips + slots - 1) / slots;
    gx = (nstrips + rounds - 1) / rounds;
  }
  if (gx > nstrips) gx = nstrips;
  if (p->fmt == RUMPY_FMT_F16 && p->w_lo) hipLaunchKernelGGL((conv_up_kernel<RUMPY_FMT_F16, true>), dim3(gx, p->cout_tiles), dim3(BTHREADS), 0, s, d);
  else if (p->fmt == RUMPY_FMT_F16) hipLaunchKernelGGL(conv_up_kernel<RUMPY_FMT_F16>, dim3(gx, p->cout_tiles), dim3(BTHREADS), 0, s, d);
  else hipLaunchKernelGGL(conv_up_kernel<RUMPY_FMT_BF16>, dim3(gx, p->cout_tiles), dim3(BTHREADS), 0, s, d);
  return 0;
}

// EXPERIMENTAL (librumpy_exp.so, not the product library; measured NOT to pay: DESIGN.md 4.2 item 10).
// One residual block per launch on HALF strips, two launch chains side by side (round 3).
//
// conv_block.hip's workgroup (6 output rows, 512 threads, 115 KB of LDS, 2 x 252 registers per SIMD lane) fills a CU alone: the boundary
// between two dependent launches (2.9 us), the tile that comes back from the memory side because the previous launch has just written it
// (+2.6 us) and the store tail are paid with an idle matrix pipe (tests/tools/overlap/overlap_probe.hip).  This kernel is the same block on
// strips of THREE output rows by FOUR waves: 7 input rows + 5 rows of the intermediate activation = 76.8 KB, so TWO workgroups fit a CU - and
// rumpy_conv_block_split launches the two halves of the batch as two chains on two streams, whose launches run out of phase: one chain's
// boundary, tile load and epilogues under the other chain's sweeps.  Price: T on 5 rows for 3 output rows (5/3 instead of 8/6 halo
// recompute): 12.5 % more MFMAs.
//
// Everything else is conv_block.hip with the row halves removed: wave q = output channels 16q.. of ALL rows of the strip, filter slice
// stationary in 72 VGPRs, block_common.hpp's swizzled LDS images / pipelined sweep / paired-tile epilogues / whole-line non-temporal stores.
// First phase in two sweeps (T rows 0..2, then 3..4: 36 + 24 accumulator registers instead of 60 at once), second phase one sweep.  Per
// accumulator the MFMA order is conv_block.hip's, and so is every epilogue operation: OUT, T and the mask bytes are BITWISE those of
// rumpy_conv_block (tests/tools/chain_tests.py::test_half_strip_block_launches_are_bitwise_the_whole_strip_launches).
// Forms: FORM 1 (forward: ReLU, mask bytes written) and FORM 3 (data gradient: * scale1, mask bytes read), bf16, W <= 48 - what the
// training step of the 64-feature EDSR launches.
#include "block_common.hpp"
#include "rumpy_experimental.h"
#include <mutex>

constexpr int HSH = 3;                                  // output rows of a half strip
constexpr int HXROWS = HSH + 4, HTROWS = HSH + 2;       // 7 input rows, 5 T rows
constexpr int HXC = BCOLS, HTC = BCOLS;                 // one zero halo column per side (the strip spans the image)
constexpr int HXBYTES = HXROWS * HXC * 128;             // 44,800
constexpr int HTBYTES = HTROWS * HTC * 128;             // 32,000
constexpr int HTHREADS = 256;
constexpr int HXPIECES = HXROWS * HXC * 8;              // 2,800 16-byte pieces of the input tile
constexpr int HXREGS = (HXPIECES + HTHREADS - 1) / HTHREADS;   // 11
constexpr int HSPIECES = HSH * BSW * 8;                 // 1,152 pieces of the strip's own rows
constexpr int HSREGS = (HSPIECES + HTHREADS - 1) / HTHREADS;   // 5 (the last one half used)

template <int FORM, int FMT = RUMPY_FMT_BF16>
__global__ void __launch_bounds__(HTHREADS, 2) conv_hblock_kernel(BlockDev a) {
  static_assert(FORM == 1 || FORM == 3, "forward (1) and mask-byte data-gradient (3) forms");
  __shared__ __attribute__((aligned(16))) unsigned char lds[HXBYTES + HTBYTES];
  unsigned char* const ldx = lds;
  unsigned char* const ldt = lds + HXBYTES;
  const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, g = lane >> 4;
  const int strip = xcd_strip(blockIdx.x, gridDim.x);
  const int n = strip / a.sy_n, sy = strip - n * a.sy_n;        // a.sy_n = ceil(H / 3)

  // ---- phase 0: input rows 3sy-2 .. 3sy+4, columns -1 .. 48 -> LDS (branch-free loads, zero outside the image) ----
  bf16x8 F[18];
  {
    uint4 R[HXREGS];
    const int y0 = sy * HSH - 2;
#pragma unroll
    for (int i = 0; i < HXREGS; ++i) {
      const int p = tid + HTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      const int lr = pix / HXC, lc = pix - lr * HXC;
      const int y = y0 + lr, x = lc - 1;
      const bool ok = (p < HXPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
      const int e = ok ? ((n * a.H + y) * a.W + x) * 64 + part * 8 : 0;
      uint4 v = *reinterpret_cast<const uint4*>(a.x + (unsigned)e);
      if (!ok) v = make_uint4(0, 0, 0, 0);
      R[i] = v;
    }
    {                                        // first filter behind the tile's requests (returns in order): under the tile's latency
      const uint4* wp = a.w1 + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
    }
    if (tid < HTROWS * 2 * 8) {              // border columns of the T image: convB's zero padding, never written by the epilogue
      const int row = tid >> 4, side = (tid >> 3) & 1, chunk = tid & 7;
      *reinterpret_cast<uint4*>(ldt + swz(row * HTC + side * (HTC - 1), chunk)) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < HXREGS; ++i) {
      const int p = tid + HTHREADS * i;
      if (p < HXPIECES) *reinterpret_cast<uint4*>(ldx + swz(p >> 3, p & 7)) = R[i];
    }
  }
  const int c0 = 16 * q + 4 * g;
  const int gpair = 4 * (g & ~1);
  const int chunk8 = 2 * q + (gpair >> 3);          // 16-byte chunk of this lane's 8 channels in the paired layout
  __syncthreads();

  // ---- phase 1: T rows j = 0 .. 4 (image rows 3sy-1+j) from input rows j .. j+2, as two sweeps: rows 0 .. 2, then rows 3 .. 4 ----
  // tiles of the first sweep as pairs (conv_block.hip's second-phase pattern): k < 3: X = (row k, col tile 0), Y = (row k, col tile 1);
  // k = 3: X = (0, 2), Y = (1, 2); single: (2, 2).  Second sweep: k < 2: X = (3 + k, 0), Y = (3 + k, 1); k = 2: X = (3, 2), Y = (4, 2).
  // Element offset of a lane's pixel (channel 16q + gpair) in an [N,H,W,64] tensor, or "outside the image" (T is zero there: convB's padding)
  auto t_off = [&](int j, int xx, int ch) -> unsigned {
    const int y = sy * HSH - 1 + j;
    return (((unsigned)y < (unsigned)a.H) & (xx < a.W)) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + ch) : 0xffffffffu;
  };
  unsigned moffA[4], moffB[3], moffS;
  unsigned MBA[FORM == 3 ? 4 : 1], MBB[FORM == 3 ? 3 : 1], MBS = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int jr = (k < 3) ? k : (g & 1), c = (k < 3) ? (g & 1) : 2;
    moffA[k] = t_off(jr, 16 * c + px, 16 * q + gpair);
    if (FORM == 3) MBA[FORM == 3 ? k : 0] = a.mbits[(moffA[k] != 0xffffffffu ? moffA[k] : 0u) >> 3];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int jr = 3 + ((k < 2) ? k : (g & 1)), c = (k < 2) ? (g & 1) : 2;
    moffB[k] = t_off(jr, 16 * c + px, 16 * q + gpair);
    if (FORM == 3) MBB[FORM == 3 ? k : 0] = a.mbits[(moffB[k] != 0xffffffffu ? moffB[k] : 0u) >> 3];
  }
  moffS = t_off(2, 32 + px, c0);
  if (FORM == 3) MBS = a.mbits[(moffS != 0xffffffffu ? moffS : 0u) >> 3];

  f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (a.b1) { const float4 t = *reinterpret_cast<const float4*>(a.b1 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
  auto post1 = [&](f32x4 t) -> f32x4 {
    if (FORM == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = relu_f32(t[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] *= a.scale1;
    }
    return t;
  };
  // one paired tile -> the T image: even-g lanes hold 8 channels of tile X's pixel, odd-g lanes of tile Y's (T row j, column xx of the lane)
  auto t_pair = [&](const f32x4& ax, const f32x4& ay, int j, int xx, unsigned mo, unsigned mb) {
    float v[8];
    pair_up(post1(ax), post1(ay), g, v);
    uint4 o = make_uint4(0, 0, 0, 0);
    if (mo != 0xffffffffu) {
      const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
      o = make_uint4(lo.x, lo.y, hi.x, hi.y);
      if (FORM == 3) o = relu_mask_bits(o, mb);
    }
    *reinterpret_cast<uint4*>(ldt + swz(j * HTC + xx + 1, chunk8)) = o;
  };
  unsigned off[8][2];
  {
    f32x4 acc[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    sweep_bases<HXC>(off, 0u, 0, px, g);
    block_sweep<3, FMT, NoHook, 3, HXC>(acc, F, lds, off);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int jr = (k < 3) ? k : (g & 1), c = (k < 3) ? (g & 1) : 2;
      t_pair((k < 3) ? acc[k < 3 ? k : 0][0] : acc[0][2], (k < 3) ? acc[k < 3 ? k : 0][1] : acc[1][2], jr, 16 * c + px, moffA[k], MBA[FORM == 3 ? k : 0]);
    }
    {                                        // the single tile (row 2, col tile 2): 4 channels c0 .. c0+3 per lane
      const f32x4 t = post1(acc[2][2]);
      uint2 o = make_uint2(0, 0);
      if (moffS != 0xffffffffu) {
        o = pack4<FMT>(t[0], t[1], t[2], t[3]);
        if (FORM == 3) {
          const uint4 m4 = relu_mask_bits(make_uint4(o.x, o.y, 0, 0), MBS >> (4 * (g & 1)));
          o = make_uint2(m4.x, m4.y);
        }
      }
      *reinterpret_cast<uint2*>(ldt + swz(2 * HTC + 32 + px + 1, 2 * q + (g >> 1)) + (g & 1) * 8) = o;
    }
  }
  {
    f32x4 acc[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    sweep_bases<HXC>(off, 0u, 3, px, g);
    block_sweep<2, FMT, NoHook, 3, HXC>(acc, F, lds, off);
    {                                        // second filter: L2 hits that land under the epilogue
      const uint4* wp = a.w2 + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int jr = 3 + ((k < 2) ? k : (g & 1)), c = (k < 2) ? (g & 1) : 2;
      t_pair((k < 2) ? acc[k < 2 ? k : 0][0] : acc[0][2], (k < 2) ? acc[k < 2 ? k : 0][1] : acc[1][2], jr, 16 * c + px, moffB[k], MBB[FORM == 3 ? k : 0]);
    }
  }
  // element offsets of this thread's pieces of the strip's own 3 rows (T and OUT stores): piece p = tid + 256 i = chunk p & 7 of pixel p >> 3
  unsigned soff[HSREGS];
#pragma unroll
  for (int i = 0; i < HSREGS; ++i) {
    const int p = tid + HTHREADS * i, pix = p >> 3, r = pix / BSW, col = pix - r * BSW, y = sy * HSH + r;
    soff[i] = (p < HSPIECES && y < a.H && col < a.W) ? (unsigned)(((n * a.H + y) * a.W + col) * 64 + (p & 7) * 8) : 0xffffffffu;
  }
  __syncthreads();                           // every wave's 16 channels of the 5 T rows are in LDS
  // The strip's own rows of T (and their ReLU mask bytes) go to HBM from the finished LDS image: whole lines, non-temporal, one piece after
  // every third MFMA group of the second sweep
  uint4 S[HSREGS];
  const bool t_out = a.t != nullptr;
  if (t_out) {
#pragma unroll
    for (int i = 0; i < HSREGS; ++i) {
      const int p = tid + HTHREADS * i, pix = (p < HSPIECES ? p : 0) >> 3, r = pix / BSW, col = pix - r * BSW;
      S[i] = *reinterpret_cast<const uint4*>(ldt + swz((r + 1) * HTC + col + 1, p & 7));
    }
  }
  auto t_store = [&](int grp) {             // grp is a constant after unrolling: piece i after the MFMAs of group 3 i
    if (grp % 3 == 0 && grp / 3 < HSREGS) {
      const int i = grp / 3 < HSREGS ? grp / 3 : 0;
      if (t_out && soff[i] != 0xffffffffu) {
        st16_nt(a.t + soff[i], S[i]);
        if (FORM == 1 && a.mbits) a.mbits[soff[i] >> 3] = (unsigned char)relu_bits(S[i]);
      }
    }
  };

  // ---- phase 2: output rows 0 .. 2 of the strip from T rows r .. r+2 ; OUT = X + scale2 * (convB(T) + b2) [+ res2] ----
  {
    f32x4 acc[3][3];
    f32x4 c4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.b2) { const float4 t = *reinterpret_cast<const float4*>(a.b2 + c0); c4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = c4;
    sweep_bases<HTC>(off, (unsigned)HXBYTES, 0, px, g);
    block_sweep<3, FMT, decltype(t_store), 3, HTC>(acc, F, lds, off, t_store);
    // pairs k < 3: X = (row k, col 0), Y = (row k, col 1); k = 3: X = (0, 2), Y = (1, 2); single: (2, 2)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 tx = (k < 3) ? acc[k < 3 ? k : 0][0] : acc[0][2];
      const f32x4 ty = (k < 3) ? acc[k < 3 ? k : 0][1] : acc[1][2];
      float v[8], m[8];
      pair_up(tx, ty, g, v);
      const int r = (k < 3) ? k : (g & 1), c = (k < 3) ? (g & 1) : 2;
      const int y = sy * HSH + r, xx = 16 * c + px;
      if (y < a.H && xx < a.W) {
        unpack8<FMT>(*reinterpret_cast<const uint4*>(ldx + swz((r + 2) * HXC + xx + 1, chunk8)), m);   // residual = the input tile
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a.scale2, m[j]);
        if (a.res2) {
          const unsigned o = (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair);
          unpack8<FMT>(*reinterpret_cast<const uint4*>(a.res2 + o), m);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += m[j];
        }
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<uint4*>(ldx + swz((r + 2) * HXC + xx + 1, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);   // OUT image, in place of the input pixel
      }
    }
    {
      const int y = sy * HSH + 2, xx = 32 + px;
      if (y < a.H && xx < a.W) {
        float v[4] = {acc[2][2][0], acc[2][2][1], acc[2][2][2], acc[2][2][3]};
        float m[4];
        unpack4<FMT>(*reinterpret_cast<const uint2*>(ldx + swz((2 + 2) * HXC + xx + 1, 2 * q + (g >> 1)) + (g & 1) * 8), m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], a.scale2, m[j]);
        if (a.res2) {
          const unsigned o = (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0);
          unpack4<FMT>(*reinterpret_cast<const uint2*>(a.res2 + o), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        *reinterpret_cast<uint2*>(ldx + swz((2 + 2) * HXC + xx + 1, 2 * q + (g >> 1)) + (g & 1) * 8) = pack4<FMT>(v[0], v[1], v[2], v[3]);
      }
    }
  }
  // ---- OUT: the strip's 3 rows now sit in LDS in place of the input tile's centre rows -> whole lines to HBM, non-temporal ----
  __syncthreads();
#pragma unroll
  for (int i = 0; i < HSREGS; ++i) {
    const int p = tid + HTHREADS * i, pix = (p < HSPIECES ? p : 0) >> 3, r = pix / BSW, col = pix - r * BSW;
    S[i] = *reinterpret_cast<const uint4*>(ldx + swz((r + 2) * HXC + col + 1, p & 7));
  }
#pragma unroll
  for (int i = 0; i < HSREGS; ++i)
    if (soff[i] != 0xffffffffu) st16_nt(a.out + soff[i], S[i]);
}

// ---- the second stream of a device and the two events of a fork / join (created once, outside any stream capture: the first call of a
// process is an eager one - the engine's warm-up pass) ----
namespace {
struct SplitCtx { hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
SplitCtx g_split[64];
std::mutex g_split_mu;
SplitCtx* split_ctx() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(g_split_mu);
  SplitCtx& c = g_split[dev];
  if (!c.side) {
    if (hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess) { c.side = nullptr; return nullptr; }
    if (hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c.join, hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return &c;
}
}  // namespace

// phase offset between the two chains: both start from one fork and their launches take the same time, so without it they stay IN phase -
// both wait for their tiles, then share the matrix pipe - which is what co-residence must avoid (overlap_probe: 512 half workgroups in one
// launch 15.6-16.4 us against 12.4-12.8 for chains that run out of phase).  One wave that sleeps for `ticks` of the 100 MHz counter.
__global__ void __launch_bounds__(64) split_delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static void hblock_launch(const rumpy_block_args* p, int n0, int cnt, hipStream_t s) {
  BlockDev d;
  const size_t img = (size_t)p->H * p->W * 64;
  d.x = (const uint16_t*)p->x + n0 * img; d.w1 = (const uint4*)p->w1; d.b1 = p->b1; d.w2 = (const uint4*)p->w2; d.b2 = p->b2;
  d.mask = nullptr; d.res2 = p->res2 ? (const uint16_t*)p->res2 + n0 * img : nullptr;
  d.t = p->t ? (uint16_t*)p->t + n0 * img : nullptr; d.out = (uint16_t*)p->out + n0 * img;
  d.N = cnt; d.H = p->H; d.W = p->W; d.sy_n = (p->H + HSH - 1) / HSH; d.relu1 = p->relu1; d.scale1 = p->scale1; d.scale2 = p->scale2;
  d.res_mode = 0; d.res1 = nullptr; d.pool = nullptr; d.ct_n = 1;
  d.mbits = p->maskbits ? (unsigned char*)p->maskbits + n0 * (img / 8) : nullptr;
  const dim3 grid(cnt * d.sy_n);
  if (p->relu1) RUMPY_LAUNCH_PROBED(5, (conv_hblock_kernel<1, RUMPY_FMT_BF16>), grid, dim3(HTHREADS), s, d);
  else RUMPY_LAUNCH_PROBED(5, (conv_hblock_kernel<3, RUMPY_FMT_BF16>), grid, dim3(HTHREADS), s, d);
}

// what rumpy_conv_block does for the two ResBlock forms of a training step, as half-strip launches: images [0, ceil(N/2)) on `stream`, the
// rest on the device's second stream.  flags: RUMPY_SPLIT_FORK = the second stream first waits for everything queued on `stream` so far (first
// block of a run of split launches), RUMPY_SPLIT_JOIN = `stream` afterwards waits for the second stream (last block of the run); in between
// the two chains only follow their own stream's order - that is the point.  flags & RUMPY_SPLIT_ONE_STREAM: both halves on `stream` (A/B, tests).
extern "C" int rumpy_conv_block_split(const rumpy_block_split_args* sp, void* stream) {
  if (!sp) { rumpy_set_error("rumpy_conv_block_split: null pointer"); return RUMPY_E_ARG; }
  const rumpy_block_args* p = &sp->block;
  if (!p->x || !p->w1 || !p->w2 || !p->out) { rumpy_set_error("rumpy_conv_block_split: null pointer"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->W > BSW) { rumpy_set_error("rumpy_conv_block_split: bad shape (N=%d H=%d W=%d; W <= 48)", p->N, p->H, p->W); return RUMPY_E_ARG; }
  if ((int64_t)p->N * p->H * p->W * 64 >= (int64_t)0xffffffffu) { rumpy_set_error("rumpy_conv_block_split: tensor beyond 32-bit element offsets"); return RUMPY_E_ARG; }
  const bool fwd = p->relu1 && p->scale1 == 1.0f && !p->mask, bwd = !p->relu1 && p->maskbits && !p->mask;
  if (p->res_mode != 0 || p->pool || p->fmt != RUMPY_FMT_BF16 || p->col_tile || !(fwd || bwd)) {
    rumpy_set_error("rumpy_conv_block_split: the ResBlock forward form (relu1, scale1 = 1) or the mask-byte data-gradient form, bf16"); return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const int na = (p->N + 1) / 2, nb = p->N - na;
  if ((sp->flags & RUMPY_SPLIT_ONE_STREAM) || nb == 0) {
    hblock_launch(p, 0, na, s);
    if (nb) hblock_launch(p, na, nb, s);
    return rumpy_check_launch("rumpy_conv_block_split");
  }
  SplitCtx* c = split_ctx();
  if (!c) { rumpy_set_error("rumpy_conv_block_split: no second stream"); return RUMPY_E_LAUNCH; }
  if (sp->flags & RUMPY_SPLIT_FORK) {
    if (hipEventRecord(c->fork, s) != hipSuccess || hipStreamWaitEvent(c->side, c->fork, 0) != hipSuccess) {
      rumpy_set_error("rumpy_conv_block_split: fork failed"); return RUMPY_E_LAUNCH; }
    static const int delay_ticks = getenv("RUMPY_SPLIT_DELAY_US") ? (int)(atof(getenv("RUMPY_SPLIT_DELAY_US")) * 100.0) : 500;    // default 5 us: half a launch
    if (delay_ticks > 0) hipLaunchKernelGGL(split_delay_kernel, dim3(1), dim3(64), 0, c->side, (unsigned long long)delay_ticks);
  }
  hblock_launch(p, 0, na, s);
  hblock_launch(p, na, nb, c->side);
  if (sp->flags & RUMPY_SPLIT_JOIN) {
    if (hipEventRecord(c->join, c->side) != hipSuccess || hipStreamWaitEvent(s, c->join, 0) != hipSuccess) {
      rumpy_set_error("rumpy_conv_block_split: join failed"); return RUMPY_E_LAUNCH; }
  }
  return rumpy_check_launch("rumpy_conv_block_split");
}

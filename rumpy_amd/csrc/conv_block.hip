// Two 3x3 convolutions 64 -> 64 with the activation between them kept in LDS: one residual block per launch.
//
//   forward  (EDSR ResBlock, architectures.py:24-44 / common.py ResBlock):  t = relu(conv1(x) + b1) ;  out = x + s * (conv2(t) + b2)
//   backward (its data gradient):                                           gt = mask(t) . s * conv2^T(g) ;  gx = g + conv1^T(gt)
// both are   T = post1(convA(X)) ,  OUT = X + scale2 * convB(T)   with post1 = [+bias] [ReLU] [* scale1] [ReLU mask].
//
// Why: at 288 pixels per CU a 64 -> 64 layer launched alone is bounded by what it pays once per launch (kernel boundary
// incl. the 9.4 MB write-back, first HBM loads, store drain: DESIGN.md 4.1), not by its MFMAs.  A block in one launch pays
// them once for two layers, never re-reads T from HBM and takes the residual operand from the input tile that is already in
// LDS; the price is the halo recompute: T is needed on 8 rows (6 + one halo row each side) -> 4/3 of convA's MFMAs.
// No inter-workgroup communication: a strip spans the image width (W <= 48), so T needs no column halo; rows outside the
// image are the zero padding of convB and are forced to zero (convA evaluated there would not be).
//
// Geometry = conv_strip.hip: strip of 6 output rows x 48 columns per 512-thread workgroup, wave (q, rh) = output channels
// 16q.. of one row half, filter slice stationary in 72 VGPRs (re-fetched between the phases), MFMA 16x16x32 bf16, reads of
// group i+1 issued ahead of the MFMAs of group i, paired-tile 16-byte epilogues.  LDS holds the 10 x 50 input pixels and the
// 8 x 50 T pixels UNPADDED (128 B per pixel, 115.2 KB together; the 96-byte-stride layout of conv_strip would need 172.8 KB)
// with the 16-byte chunk index XOR-ed with (linear pixel index & 7): conflict-free for every ds_read_b128 fragment read
// (checked by enumeration against the 4 x 16-lane groups of MI355X_MICROARCH.md, LDS table).  The XOR term of a read depends
// on lane constants and on (2*row + tap column) & 7 only (50 = 2 mod 8, 16 = 0 mod 8): 8 x 2 per-lane base addresses are
// prepared per phase and every read is base + immediate.
//
// Round 3: images wider than one strip.  G = BlockGeo<NC, true> (block_common.hpp) cuts the image into column tiles of 16 NC output
// columns (NC = 2 or 3; 64 x 64 crops - the reference's shipped training shape - are two tiles of 32): the input image in LDS then has
// two halo columns per side, and T on the one halo column per side that convB needs is one more MFMA tile per wave (its 4 T rows x 2
// columns = 8 of the tile's 16 pixels; halo_sweep), computed before the main sweep of the first phase.  Everything else - work split,
// sweeps, row-half gates, whole-line stores - is the W <= 48 kernel with the geometry as a template parameter; per accumulator the MFMA
// order is unchanged, so the results are bitwise those of the two-launch path (test_conv_block_column_tiles_*).
#include "block_common.hpp"
#include <cstdlib>
#include <cstdio>

// (The timing-experiment builds of rounds 1-3 - phase stamps, pieces compiled out, alternative prologue orders - live as a patch under
// tests/tools/patches/abl_r03.patch; this file holds the product kernel only.)

// GEN = false: the ResBlock form (residual operand = block input, read from LDS).  GEN = true: the general form used for
// RCABs - res_mode 1 (no residual) or 2 (residual operand res1 from HBM, prefetched under the second sweep) and the
// per-(strip, row half) channel sums of scale2 * (convB(T) + b2) for the channel-attention pool.
// FORM: 0 = post1 flags read at run time; 1 = forward form (ReLU, no scale1, no mask; stores the mask bytes if asked to); 2 = data-gradient
// form (no ReLU, * scale1, mask = a bf16 activation); 3 = data-gradient form with the mask as bytes (block_common.hpp::relu_bits):
// the two forms the engine launches, without the per-value selects and branches of the generic epilogue.
// FMT: element format (RUMPY_FMT_F16 is instantiated for the ResBlock forward form only: evaluation plans)
template <bool GEN, int FORM = 0, int FMT = RUMPY_FMT_BF16, class G = GeoL>
__global__ void __launch_bounds__(BTHREADS, 2) conv_block_kernel(BlockDev a) {
  constexpr int NC = G::NC, XC = G::XC, TC = G::TC, XH = G::XH;
  constexpr int SH = G::SH, OR = G::OR, TR = G::TR;       // strip rows, output rows / T rows per row half
  constexpr int NP1 = NC == 3 ? 6 : TR;    // paired tiles of the first phase (TR rows x NC column tiles; NC = 2: one pair per row)
  constexpr int NP2 = NC == 3 ? 4 : OR;    // ... of the second phase (OR rows x NC; NC = 3 leaves one single tile)
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::XBYTES + G::TBYTES];
  __shared__ unsigned gate[8];             // waves of row half 0 / 1 that have written their T rows [0, 1], their OUT rows [2, 3] (block_common.hpp::gate_*);
                                           // [4] waves that have written the early part of the input tile, [5] row half 1's waves: the late part
  unsigned char* const ldx = lds;
  unsigned char* const ldt = lds + G::XBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = __builtin_amdgcn_readfirstlane(wave >> 2), tg = tid & 255;
  const int strip = xcd_strip(blockIdx.x, gridDim.x);
  // strip -> (image, column tile, strip row), strip rows fastest: vertical neighbours (4 shared input rows of 10) sit next to each other
  int n, sy, ct = 0;
  if (G::CT) { sy = strip % a.sy_n; const int r = strip / a.sy_n; ct = r % a.ct_n; n = r / a.ct_n; }
  else { n = strip / a.sy_n; sy = strip - n * a.sy_n; }
  const int x0 = ct * G::OW;               // image column of the strip's first output column

  // ---- phase 0: input rows 6sy-2 .. 6sy+7, columns x0-XH .. x0+OW+XH-1 -> LDS (branch-free loads, zero outside the image) ----
  // Round 3: the tile arrives bandwidth-priced (every CU asks for its 64 KB at the same moment: 2.8 us), and row half 0 only sweeps
  // over input rows 0 .. 5.  The pieces of those rows (R0 rounds of 512) are loaded by all threads first and announced on their own LDS counter;
  // the rest is loaded by row half 1's threads (row half 0 issues as many loads of one cached line instead: every wave runs the same,
  // unconditional load sequence, so the compiler's waits stay counted).  Row half 0 starts its first sweep when rows 0 .. 5 are in LDS, row half
  // 1 when everything is; the first conv's filter is requested before the tile, so its (L2-hit) latency lies under the tile's.
  constexpr int R0 = ((TR + 2) * XC * 8 + BTHREADS - 1) / BTHREADS;                        // rounds that cover input rows 0 .. TR + 1 (row half 0's window)
  constexpr int LATE = G::XPIECES - R0 * BTHREADS > 0 ? G::XPIECES - R0 * BTHREADS : 0;
  constexpr int R1 = (LATE + 255) / 256;                                                     // rounds of row half 1's 256 threads for the rest
  if (tid < 8) gate[tid] = 0u;
  __syncthreads();                         // (nothing is in flight yet: a bare s_barrier) the counters are zero before anybody arrives
  bf16x8 F[18];
  auto fetch_filter = [&]() {
    const uint4* wp = a.w1 + (size_t)q * 18 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
  };
  {
    uint4 R[R0], Rl[R1 > 0 ? R1 : 1];
    const int y0 = sy * SH - 2;
    auto fetch = [&](int p, bool live) -> uint4 {
      const int pix = p >> 3, part = p & 7;
      const int lr = pix / XC, lc = pix - lr * XC;
      const int y = y0 + lr, x = x0 - XH + lc;
      const bool ok = live & (p < G::XPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
      const int e = ok ? ((n * a.H + y) * a.W + x) * 64 + part * 8 : 0;
      uint4 v = *reinterpret_cast<const uint4*>(a.x + (unsigned)e);
      if (!ok) v = make_uint4(0, 0, 0, 0);
      return v;
    };
#pragma unroll
    for (int i = 0; i < R0; ++i) R[i] = fetch(tid + BTHREADS * i, true);
#pragma unroll
    for (int i = 0; i < R1; ++i) Rl[i] = fetch(R0 * BTHREADS + tg + 256 * i, rh == 1);
    fetch_filter();                          // behind the tile's requests (returns in order): under the tile's latency, not in front of it
    // border columns of the T image: convB's zero padding, never written by the epilogue (column tiles: real T values, written by the halo tile)
    if (!G::CT && tid < G::TROWS * 2 * 8) {
      const int row = tid >> 4, side = (tid >> 3) & 1, chunk = tid & 7;
      *reinterpret_cast<uint4*>(ldt + swz(row * TC + side * (TC - 1), chunk)) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < R0; ++i) {
      const int p = tid + BTHREADS * i;
      if (p < G::XPIECES) *reinterpret_cast<uint4*>(ldx + swz(p >> 3, p & 7)) = R[i];
    }
    gate_arrive(&gate[4], lane);           // this wave's pieces of input rows 0 .. 5 (and a bit) are in LDS
    if (R1 > 0 && rh == 1) {
#pragma unroll
      for (int i = 0; i < R1; ++i) {
        const int p = R0 * BTHREADS + tg + 256 * i;
        if (p < G::XPIECES) *reinterpret_cast<uint4*>(ldx + swz(p >> 3, p & 7)) = Rl[i];
      }
      gate_arrive(&gate[5], lane);
    }
  }
  const int c0 = 16 * q + 4 * g;
  const int gpair = 4 * (g & ~1);
  const int chunk8 = 2 * q + (gpair >> 3);          // 16-byte chunk of this lane's 8 channels in the paired layout
  gate_wait(&gate[4], 8u);                 // input rows 0 .. 5: all eight waves' early pieces
  if (R1 > 0 && rh == 1) gate_wait(&gate[5], 4u);   // rows 6 .. 9: row half 1's own late pieces

  // ---- phase 1: T rows j = 4rh .. 4rh+3 (image rows 6sy-1+j) from input rows j .. j+2 ----
  // tile pairs: k < 4: X = (row k, col tile 0), Y = (row k, col tile 1); k = 4: X = (0, 2), Y = (1, 2); k = 5: X = (2, 2), Y = (3, 2)
  unsigned moff[NP1];
  uint4 M[(FORM == 1 || FORM == 3) ? 1 : NP1];
  unsigned MB[FORM == 3 ? NP1 : 1];
#pragma unroll
  for (int k = 0; k < NP1; ++k) {
    const int jr = (k < TR) ? k : (2 * (k - TR) + (g & 1)), c = (k < TR) ? (g & 1) : 2;
    const int y = sy * SH - 1 + TR * rh + jr, xx = x0 + 16 * c + px;
    const bool in = ((unsigned)y < (unsigned)a.H) & (xx < a.W);
    moff[k] = in ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
    if (FORM == 0 || FORM == 2) {
      M[(FORM == 1 || FORM == 3) ? 0 : k] = make_uint4(0, 0, 0, 0);
      if (FORM == 2 || a.mask) M[(FORM == 1 || FORM == 3) ? 0 : k] = *reinterpret_cast<const uint4*>(a.mask + (in ? moff[k] : 0u));
    }
    if (FORM == 3) MB[FORM == 3 ? k : 0] = a.mbits[(in ? moff[k] : 0u) >> 3];
  }
  // column tiles: this lane's pixel of the halo tile = T row hj, halo side px & 1 (lanes px >= 8 repeat lanes px - 8 and store nothing)
  const int hj = TR * rh + ((px >> 1) < TR ? (px >> 1) : TR - 1), htc = (px & 1) ? TC - 1 : 0;      // (lanes px >= 2 TR repeat the last row and store nothing)
  unsigned hoffe = 0xffffffffu;            // element offset of (that pixel, channel c0) in an [N,H,W,64] tensor, or outside the image
  uint2 HM = make_uint2(0, 0);
  unsigned HB = 0;
  if (G::CT) {
    const int y = sy * SH - 1 + hj, xx = x0 - 1 + htc;
    if (((unsigned)y < (unsigned)a.H) & ((unsigned)xx < (unsigned)a.W)) hoffe = (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0);
    if (FORM == 2 || (FORM == 0 && a.mask)) HM = *reinterpret_cast<const uint2*>(a.mask + (hoffe != 0xffffffffu ? hoffe : 0u));
    if (FORM == 3) HB = a.mbits[(hoffe != 0xffffffffu ? hoffe : 0u) >> 3];
  }
  {
    f32x4 acc[TR][NC];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.b1) { const float4 t = *reinterpret_cast<const float4*>(a.b1 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[r][c] = b4;
    unsigned off[8][2];
    auto post1 = [&](f32x4 t) -> f32x4 {
      if (FORM == 1 || (FORM == 0 && a.relu1)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = relu_f32(t[j]);
      }
      if (FORM == 2 || FORM == 3 || (FORM == 0 && a.scale1 != 1.0f)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] *= a.scale1;
      }
      return t;
    };
    auto t_pair = [&](int k) {          // k is a constant after unrolling
      const f32x4 tx = post1((k < TR) ? acc[k < TR ? k : 0][0] : acc[2 * (k < TR ? 0 : k - TR)][NC - 1]);
      const f32x4 ty = post1((k < TR) ? acc[k < TR ? k : 0][1] : acc[2 * (k < TR ? 0 : k - TR) + 1][NC - 1]);
      float v[8];
      pair_up(tx, ty, g, v);
      const int jr = (k < TR) ? k : (2 * (k - TR) + (g & 1)), c = (k < TR) ? (g & 1) : 2;
      const int j = TR * rh + jr, xx = 16 * c + px;
      uint4 o = make_uint4(0, 0, 0, 0);                      // outside the image: convB's zero padding
      if (moff[k] != 0xffffffffu) {
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (FORM == 2 || (FORM == 0 && a.mask)) o = relu_mask_packed(o, M[(FORM == 1 || FORM == 3) ? 0 : k]);
        if (FORM == 3) o = relu_mask_bits(o, MB[FORM == 3 ? k : 0]);
      }
      *reinterpret_cast<uint4*>(ldt + swz(j * TC + xx + 1, chunk8)) = o;
    };
    if (G::CT) {
      // T on the halo columns first (its accumulator is dead before the main sweep's 4 x NC tiles are live)
      sweep_bases<XC>(off, 0u, hj, 0, g, htc);
      const f32x4 th = post1(halo_sweep<FMT, XC>(b4, F, lds, off));
      uint2 o = make_uint2(0, 0);
      if (hoffe != 0xffffffffu) {
        o = pack4<FMT>(th[0], th[1], th[2], th[3]);
        if (FORM == 2 || (FORM == 0 && a.mask)) o = make_uint2(o.x & relu_keep(HM.x), o.y & relu_keep(HM.y));
        if (FORM == 3) {
          const unsigned b = HB >> (4 * (g & 1));
          const uint4 m4 = relu_mask_bits(make_uint4(o.x, o.y, 0, 0), b);
          o = make_uint2(m4.x, m4.y);
        }
      }
      if (px < 2 * TR) *reinterpret_cast<uint2*>(ldt + swz(hj * TC + htc, 2 * q + (g >> 1)) + (g & 1) * 8) = o;
    }
    sweep_bases<XC>(off, 0u, TR * rh, px, g, G::CT ? 1 : 0);
    block_sweep<TR, FMT, NoHook, NC, XC>(acc, F, lds, off);
    // second filter: L2 hits that land under the epilogue
    {
      const uint4* wp = a.w2 + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
    }
#pragma unroll
    for (int k = 0; k < NP1; ++k) t_pair(k);
    gate_arrive(&gate[rh], lane);          // this wave's 16 channels of T rows 4rh .. 4rh+3 are in LDS
  }
  unsigned soff[G::GREGS];                 // element offsets of this thread's pieces of its row half's 3 strip rows (T and OUT stores)
#pragma unroll
  for (int i = 0; i < G::GREGS; ++i) soff[i] = group_piece_off<G>(i, tg, rh, n, sy, a.H, a.W, x0);
  // No workgroup barrier between the phases: output rows 0, 1 need T rows 0 .. 3 (row half 0's own), output row 2 also row 4, output rows
  // 3 .. 5 T rows 3 .. 7 - each row half waits for exactly the T rows it reads (block_common.hpp: row-half groups).  The input image is
  // overwritten (OUT, in place) only behind a wait for the OTHER half's T rows, i.e. when nobody sweeps over it any more.
  gate_wait(&gate[rh], 4u);
  if (rh == 1) gate_wait(&gate[0], 4u);
  // The row half's own strip rows of T (and their ReLU mask bytes) go to HBM from the finished LDS image: whole lines, non-temporal, one
  // piece after every third MFMA group of the second sweep (block_common.hpp::strip_stage).
  uint4 S[G::GREGS];
  const bool t_out = a.t != nullptr;
  if (t_out) group_stage<1, G>(S, ldt, tg, rh);
  auto t_store = [&](int grp) {           // grp is a constant after unrolling: piece i after the MFMAs of group 3 i
    if (grp % 3 == 0 && grp / 3 < G::GREGS) {
      const int i = grp / 3 < G::GREGS ? grp / 3 : 0;
      if (t_out && soff[i] != 0xffffffffu) {
        st16_nt(a.t + soff[i], S[i]);
        if (FORM == 1 && a.mbits) a.mbits[soff[i] >> 3] = (unsigned char)relu_bits(S[i]);
      }
    }
  };

  // ---- phase 2: output rows 3rh .. 3rh+2 of the strip from T rows r .. r+2 ; OUT = X + scale2 * (convB(T) + b2) [+ res2] ----
  {
    f32x4 acc[OR][NC];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.b2) { const float4 t = *reinterpret_cast<const float4*>(a.b2 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < OR; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[r][c] = b4;
    // GEN, res_mode 2: the residual vectors are requested before the sweep and land under it
    unsigned roff[NP2], rsoff = 0xffffffffu;
    uint4 P1p[GEN ? NP2 : 1];
    uint2 P1s = make_uint2(0, 0);
    if (GEN) {
#pragma unroll
      for (int k = 0; k < NP2; ++k) {
        const int r = (k < OR) ? k : (g & 1), c = (k < OR) ? (g & 1) : 2;
        const int y = sy * SH + OR * rh + r, xx = x0 + 16 * c + px;
        roff[k] = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
        P1p[k] = make_uint4(0, 0, 0, 0);
        if (a.res_mode == 2) P1p[k] = *reinterpret_cast<const uint4*>(a.res1 + (roff[k] != 0xffffffffu ? roff[k] : 0u));
      }
      if (NC == 3) {
        const int y = sy * SH + OR * rh + 2, xx = x0 + 32 + px;
        rsoff = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0) : 0xffffffffu;
        if (a.res_mode == 2) P1s = *reinterpret_cast<const uint2*>(a.res1 + (rsoff != 0xffffffffu ? rsoff : 0u));
      }
    }
    unsigned off[8][2];
    if (rh == 0) {
      sweep_bases<TC>(off, (unsigned)G::XBYTES, 0, px, g);
      block_sweep<OR - 1, FMT, decltype(t_store), NC, TC>(*reinterpret_cast<f32x4(*)[OR - 1][NC]>(&acc[0]), F, lds, off, t_store);   // output rows 0 .. OR-2 <- T rows 0 .. OR (this half's own)
      gate_wait(&gate[1], 4u);
      sweep_bases<TC>(off, (unsigned)G::XBYTES, OR - 1, px, g);
      block_sweep<1, FMT, NoHook, NC, TC>(*reinterpret_cast<f32x4(*)[1][NC]>(&acc[OR - 1]), F, lds, off);                              // output row OR-1 <- T rows OR-1 .. OR+1
    } else {
      sweep_bases<TC>(off, (unsigned)G::XBYTES, OR, px, g);
      block_sweep<OR, FMT, decltype(t_store), NC, TC>(acc, F, lds, off, t_store);                                                     // output rows OR .. SH-1 <- T rows OR .. SH+1
    }
    float ps[4] = {0.f, 0.f, 0.f, 0.f};                         // GEN pool sums: single tile, channels 4g .. 4g+3 of the wave's 16
    float ps8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};    //                paired tiles, channels 4(g&~1) .. +7
    // pairs k < 3: X = (row k, col 0), Y = (row k, col 1); k = 3: X = (0, 2), Y = (1, 2); single: (2, 2)
#pragma unroll
    for (int k = 0; k < NP2; ++k) {
      const f32x4 tx = (k < OR) ? acc[k < OR ? k : 0][0] : acc[0][NC - 1];
      const f32x4 ty = (k < OR) ? acc[k < OR ? k : 0][1] : acc[1][NC - 1];
      float v[8], m[8];
      pair_up(tx, ty, g, v);
      const int r = (k < OR) ? k : (g & 1), c = (k < OR) ? (g & 1) : 2;
      const int srow = OR * rh + r, y = sy * SH + srow, xx = 16 * c + px;
      if (y < a.H && x0 + xx < a.W) {
        if (!GEN) {
          unpack8<FMT>(*reinterpret_cast<const uint4*>(ldx + swz((srow + 2) * XC + xx + XH, chunk8)), m);   // residual = the input tile
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a.scale2, m[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= a.scale2;
          if (a.pool) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ps8[j] += v[j];
          }
          if (a.res_mode == 2) {
            unpack8<FMT>(P1p[GEN ? k : 0], m);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += m[j];
          }
        }
        if (a.res2) {
          const unsigned o = (unsigned)(((n * a.H + y) * a.W + x0 + xx) * 64 + 16 * q + gpair);
          unpack8<FMT>(*reinterpret_cast<const uint4*>(a.res2 + o), m);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += m[j];
        }
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<uint4*>(ldx + swz((srow + 2) * XC + xx + XH, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);   // OUT image, in place of the input pixel
      }
    }
    if (NC == 3) {
      const int srow = OR * rh + 2, y = sy * SH + srow, xx = 32 + px;
      if (y < a.H && x0 + xx < a.W) {
        float v[4] = {acc[2][NC - 1][0], acc[2][NC - 1][1], acc[2][NC - 1][2], acc[2][NC - 1][3]};
        float m[4];
        if (!GEN) {
          unpack4<FMT>(*reinterpret_cast<const uint2*>(ldx + swz((srow + 2) * XC + xx + XH, 2 * q + (g >> 1)) + (g & 1) * 8), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], a.scale2, m[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= a.scale2;
          if (a.pool) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ps[j] += v[j];
          }
          if (a.res_mode == 2) {
            unpack4<FMT>(P1s, m);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += m[j];
          }
        }
        if (a.res2) {
          const unsigned o = (unsigned)(((n * a.H + y) * a.W + x0 + xx) * 64 + c0);
          unpack4<FMT>(*reinterpret_cast<const uint2*>(a.res2 + o), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        *reinterpret_cast<uint2*>(ldx + swz((srow + 2) * XC + xx + XH, 2 * q + (g >> 1)) + (g & 1) * 8) = pack4<FMT>(v[0], v[1], v[2], v[3]);
      }
    }
    if (GEN && a.pool) {
      // per-(strip, row half) channel sums: pool[n][2*sy + rh][channel] (the layout of conv_strip.hip with one column strip):
      // reduce over the 16 pixel lanes, fold the odd-g lanes into the even ones, add the single tile's 4+4 channels
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = row16_sum(ps8[j]);                    // (common.hpp: the butterflies on the VALU, same bits)
        t += lane_xor16(t, g);
        ps8[j] = t;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = row16_sum(ps[j]);
        const float up = lane_xor16(t, g);              // the same sums of lane group g ^ 1
        ps8[j] += (g & 1) ? up : t;
        ps8[4 + j] += (g & 1) ? t : up;
      }
      if (px == 0 && !(g & 1)) {
        // one row of partial sums per (column tile, strip row, row half) of an image
        float* pp = a.pool + ((size_t)((n * a.ct_n + ct) * a.sy_n * 2 + 2 * sy + rh)) * 64 + 16 * q + 4 * g;
        *reinterpret_cast<float4*>(pp) = make_float4(ps8[0], ps8[1], ps8[2], ps8[3]);
        *reinterpret_cast<float4*>(pp + 4) = make_float4(ps8[4], ps8[5], ps8[6], ps8[7]);
      }
    }
  }
  // ---- OUT: the row half's 3 rows now sit in LDS in place of the input tile's centre rows (each lane replaced exactly the input values it
  // had read as its residual operand; nothing else reads the input tile in phase 2) -> whole lines to HBM, non-temporal ----
  gate_arrive(&gate[2 + rh], lane);
  gate_wait(&gate[2 + rh], 4u);
  group_stage<2, G>(S, ldx, tg, rh);
#pragma unroll
  for (int i = 0; i < G::GREGS; ++i)
    if (soff[i] != 0xffffffffu) st16_nt(a.out + soff[i], S[i]);
}

template <class G>
static void block_launch(const rumpy_block_args* p, const BlockDev& d, hipStream_t s) {
  const dim3 grid(d.N * d.sy_n * d.ct_n);
  if (p->res_mode == 0 && !p->pool) {
    if (p->fmt == RUMPY_FMT_F16) RUMPY_LAUNCH_PROBED(5, (conv_block_kernel<false, 1, RUMPY_FMT_F16, G>), grid, dim3(BTHREADS), s, d);
    else if (p->relu1 && p->scale1 == 1.0f && !p->mask) RUMPY_LAUNCH_PROBED(5, (conv_block_kernel<false, 1, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
    else if (!p->relu1 && p->maskbits) RUMPY_LAUNCH_PROBED(5, (conv_block_kernel<false, 3, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
    else if (!p->relu1 && p->mask) RUMPY_LAUNCH_PROBED(5, (conv_block_kernel<false, 2, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
    else RUMPY_LAUNCH_PROBED(5, (conv_block_kernel<false, 0, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
  } else if (p->fmt == RUMPY_FMT_F16) {
    RUMPY_LAUNCH_PROBED(5, (conv_block_kernel<true, 0, RUMPY_FMT_F16, G>), grid, dim3(BTHREADS), s, d);
  } else {
    RUMPY_LAUNCH_PROBED(5, (conv_block_kernel<true, 0, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
  }
}

// Round 4: strips of 4 / 8 rows x 32-column tiles, for the ResBlock forms the engine launches (forward bf16 / fp16, mask-byte data gradient)
template <int SH>
static void block_launch_rows(const rumpy_block_args* p, const BlockDev& d, hipStream_t s) {
  typedef BlockGeo<2, true, SH> G;
  const dim3 grid(d.N * d.sy_n * d.ct_n);
  if (p->fmt == RUMPY_FMT_F16) RUMPY_LAUNCH_PROBED(5, (conv_block_kernel<false, 1, RUMPY_FMT_F16, G>), grid, dim3(BTHREADS), s, d);
  else if (p->relu1) RUMPY_LAUNCH_PROBED(5, (conv_block_kernel<false, 1, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
  else RUMPY_LAUNCH_PROBED(5, (conv_block_kernel<false, 3, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
}

// Geometry of a ResBlock-form launch on [N, H, W] (W > 48): strip rows SH in {4, 6, 8} x column tiles of 16 NC columns.  A workgroup is alone on
// its CU, so a launch costs (rounds of workgroups on the CUs) x (time of one workgroup); one workgroup = (2 SH + 2 rows of MFMA work: the first
// conv also on the two halo rows; + about 6 rows' worth of what is paid once: prologue, gates, epilogue tails) x (NC column tiles + 1: the halo
// tile and the two extra input columns per side).  The count lands on or just under a multiple of the CUs where it can: 16 x 64 x 64 ->
// 8 rows x 32 columns = 256 workgroups (one round; 6 rows: 352 = two rounds), 339 x 510 -> 8 x 32 (688 workgroups, three rounds of less).
// RUMPY_BLOCK_GEO="SH,NC" forces one (A/B runs).
static void block_geometry(int N, int H, int W, bool rows_ok, int* sh, int* nc, int* ct_n) {
  block_col_tiles(W, nc, ct_n);
  *sh = BSH;
  if (!rows_ok || W <= BSW) return;
  const char* force = getenv("RUMPY_BLOCK_GEO");          // (read per call: the tests toggle it)
  int fh = 0, fc = 0;                                      // "SH,NC": one of the candidates below, or it is ignored with a message (it used to be read as two characters)
  if (force && (sscanf(force, "%d,%d", &fh, &fc) != 2 || !((fh == BSH && (fc == 2 || fc == 3)) || ((fh == 4 || fh == 8) && fc == 2)))) {
    static bool told = false;
    if (!told) { fprintf(stderr, "rumpy_amd: RUMPY_BLOCK_GEO=\"%s\" is not one of 6,3 6,2 8,2 4,2 - ignored\n", force); told = true; }
    fh = fc = 0;
  }
  const int cus = rumpy_device_cus();
  long best = -1;
  const int cand[4][2] = {{BSH, *nc}, {BSH, 2}, {8, 2}, {4, 2}};
  for (int i = 0; i < 4; ++i) {
    const int h = cand[i][0], c = cand[i][1];
    if (fh && (fh != h || fc != c)) continue;
    const int ct = (W + 16 * c - 1) / (16 * c);
    const long wgs = (long)N * ((H + h - 1) / h) * ct;
    const long cost = ((wgs + cus - 1) / cus) * (2 * h + 2 + 6) * (c + 1);
    if (best < 0 || cost < best) { best = cost; *sh = h; *nc = c; *ct_n = ct; }
  }
}

// rows of per-image pool partial sums a rumpy_conv_block launch with `pool` writes: 2 per (strip row, column tile)
extern "C" int rumpy_block_pool_tiles(int32_t H, int32_t W) {
  int nc, ct_n;
  block_col_tiles(W, &nc, &ct_n);
  return 2 * ((H + BSH - 1) / BSH) * ct_n;
}

int rumpy_conv_block_fp8_launch(const rumpy_block_args* p, hipStream_t s);      // conv_block_fp8.hip

extern "C" int rumpy_conv_block(const rumpy_block_args* p, void* stream) {
  if (p && p->w1_f8) {
    if (!p->x || !p->out || p->N <= 0 || p->H <= 0 || p->W <= 0) { rumpy_set_error("rumpy_conv_block: null pointer / bad shape"); return RUMPY_E_ARG; }
    if ((int64_t)p->N * p->H * p->W * 64 >= (int64_t)0xffffffffu) { rumpy_set_error("rumpy_conv_block: tensor beyond 32-bit element offsets"); return RUMPY_E_ARG; }
    const int rc = rumpy_conv_block_fp8_launch(p, (hipStream_t)stream);
    return rc ? rc : rumpy_check_launch("rumpy_conv_block");
  }
  if (!p || !p->x || !p->w1 || !p->w2 || !p->out) { rumpy_set_error("rumpy_conv_block: null pointer"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0) { rumpy_set_error("rumpy_conv_block: bad shape (N=%d H=%d W=%d)", p->N, p->H, p->W); return RUMPY_E_ARG; }
  if ((int64_t)p->N * p->H * p->W * 64 >= (int64_t)0xffffffffu) { rumpy_set_error("rumpy_conv_block: tensor beyond 32-bit element offsets"); return RUMPY_E_ARG; }
  BlockDev d;
  d.x = (const uint16_t*)p->x; d.w1 = (const uint4*)p->w1; d.b1 = p->b1; d.w2 = (const uint4*)p->w2; d.b2 = p->b2;
  d.mask = (const uint16_t*)p->mask; d.res2 = (const uint16_t*)p->res2; d.t = (uint16_t*)p->t; d.out = (uint16_t*)p->out;
  d.N = p->N; d.H = p->H; d.W = p->W; d.relu1 = p->relu1; d.scale1 = p->scale1; d.scale2 = p->scale2;
  d.res_mode = p->res_mode; d.res1 = (const uint16_t*)p->res1; d.pool = p->pool; d.mbits = (unsigned char*)p->maskbits;
  // the forms with kernels at every strip height: ResBlock forward (bf16 / fp16) and its mask-byte data gradient
  const bool rows_ok = p->res_mode == 0 && !p->pool && !p->col_tile &&
                       ((p->relu1 && p->scale1 == 1.0f && !p->mask) || (!p->relu1 && p->maskbits && p->fmt == RUMPY_FMT_BF16));
  int nc, sh;
  block_geometry(p->N, p->H, p->W, rows_ok, &sh, &nc, &d.ct_n);
  d.sy_n = (p->H + sh - 1) / sh;
  bool tiled = p->W > BSW;
  if (p->col_tile) {
    if ((p->col_tile != 2 && p->col_tile != 3) || p->pool) { rumpy_set_error("rumpy_conv_block: col_tile is 0, 2 or 3 (and not with pool)"); return RUMPY_E_ARG; }
    nc = p->col_tile; d.ct_n = (p->W + 16 * nc - 1) / (16 * nc); tiled = true;
  }
  if (p->maskbits && !(p->res_mode == 0 && !p->pool && ((p->relu1 && p->scale1 == 1.0f && !p->mask) || !p->relu1))) {
    rumpy_set_error("rumpy_conv_block: maskbits goes with the ResBlock forward form (written) or a data-gradient form (read)"); return RUMPY_E_ARG; }
  if (p->res_mode < 0 || p->res_mode > 2 || (p->res_mode == 2 && !p->res1)) { rumpy_set_error("rumpy_conv_block: bad res_mode / res1"); return RUMPY_E_ARG; }
  if (p->fmt != RUMPY_FMT_BF16 && p->fmt != RUMPY_FMT_F16) { rumpy_set_error("rumpy_conv_block: fmt %d", p->fmt); return RUMPY_E_ARG; }
  if (p->fmt == RUMPY_FMT_F16 && p->res_mode == 0 && !p->pool && !(p->relu1 && p->scale1 == 1.0f && !p->mask)) {
    rumpy_set_error("rumpy_conv_block: fmt %d goes with the forward forms only", p->fmt); return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  if (!tiled) block_launch<GeoL>(p, d, s);
  else if (sh == 8) block_launch_rows<8>(p, d, s);
  else if (sh == 4) block_launch_rows<4>(p, d, s);
  else if (nc == 3) block_launch<BlockGeo<3, true> >(p, d, s);
  else block_launch<BlockGeo<2, true> >(p, d, s);
  return rumpy_check_launch("rumpy_conv_block");
}

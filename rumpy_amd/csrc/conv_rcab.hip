// A whole residual channel-attention block (RCAB, rumpy/SISR/models/advanced/architectures.py:60-84; QRCAB,
// attention_manipulators/architectures.py:154-228) per launch, forward and backward:
//
//   forward :  t1 = relu(conv1(x) + b1) ; t2 = conv2(t1) + b2 ; gate = CA(mean_hw(t2)) [* gate_q] ; out = x + gate * t2
//   backward:  ds = sum_hw(dy * t2) -> (dz, dh, dp) through the squeeze-excite MLP ; d_t2 = dy * gate + dp / HW ;
//              gt1 = [t1 > 0] . conv2^T(d_t2) ; dx = dy + conv1^T(gt1)
//
// conv_block.hip already keeps the activation between the two convs in LDS but has to stop at the channel attention, because its gate
// needs the mean over the WHOLE image: three more streaming launches per block (gate * t2 + x ; sum dy * t2 ; dy * gate + dp) that
// re-read and re-write 9.4 MB tensors at 2-3 TB/s - 31 % of an RCAN step.  Here the strips of one image exchange their 64 partial sums
// through HBM instead (one 8-byte record {fp32 value, tag} per channel and strip, written with write-through stores and polled with
// system-coherent loads: the data is the signal, as in conv_block_chain.hip), every workgroup evaluates the 600-flop MLP itself, and the
// gate is applied to the tile that is still on chip.  Tags are (epoch << 12) + launch sequence number, epoch read from device memory
// and advanced once per pass by the host side, so a record of an earlier launch or step can never be taken for a current one and ONE
// exchange buffer serves every block of the network.
// Needs every strip of an image resident at the same time: strips of an image have consecutive workgroup ids and ids are dispatched
// in order, so the oldest unfinished image always has all of its strips on the chip as long as ceil(H/6) <= CUs (checked by the host).
// A poll that does not complete within ~0.1 s stores a code in *status and gives up (wrong numbers, reported by the host; no hang).
// Sums over strips are taken in strip order by every workgroup: all strips of an image use bit-identical gates, run to run.
//
// Round 3: images wider than 48 pixels run as column tiles (block_common.hpp::BlockGeo<NC, true>, conv_block.hip): a "strip" of the
// exchange is then a (column tile, strip row) pair, ns = ceil(H/6) * column tiles of them per image, all of which must be resident
// together (ns <= CUs; the reference's 64 x 64 training crops are 22).
#include "rcab_common.hpp"
#include <cstdlib>
#include <cstdio>

// MB (backward only): the ReLU mask comes as bytes (written by the forward launch) instead of the bf16 activation
// FMT: element format (RUMPY_FMT_F16 is instantiated for the forward launch only: evaluation plans)
template <bool BWD, bool MB = false, int FMT = RUMPY_FMT_BF16, class G = GeoL>
__global__ void __launch_bounds__(BTHREADS, 2) rcab_kernel(RcabDev a) {
  constexpr int NC = G::NC, XC = G::XC, TC = G::TC, XH = G::XH, OW = G::OW;
  constexpr int SH = G::SH, OR = G::OR, TR = G::TR;       // strip rows, output rows / T rows per row half (block_common.hpp::BlockGeo)
  constexpr int NP1 = NC == 3 ? 6 : TR;    // paired tiles of the first phase (TR rows x NC column tiles; NC = 2: one pair per row)
  constexpr int NP2 = NC == 3 ? 4 : OR;    // ... of the second phase (OR rows x NC; NC = 3 leaves one single tile)
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::XBYTES + G::TBYTES];
  __shared__ float sx[8 * 64];
  __shared__ float spool[2 * 64];
  __shared__ __attribute__((aligned(16))) float sgate[64];
  __shared__ __attribute__((aligned(16))) float sdp[64];
  // the squeeze-excite MLP's operands, fetched with the first loads of the kernel: its 2 * cr dependent steps then read LDS, not L2
  // (one L2 round trip per step was 4-5 us of the first version)
  __shared__ float sw1[RC_MAXR * 64];      // [r][c] = conv_du.0.weight
  __shared__ float sw2t[RC_MAXR * 64];     // [r][c] = conv_du.2.weight[c][r]
  __shared__ unsigned gate[4];              // row-half groups (block_common.hpp::gate_*): waves that have written their T rows / OUT rows
  __shared__ float svec[4 * 64];           // [0] conv_du.0.bias (cr) | [1] conv_du.2.bias | [2] q gate | [3] bwd: forward gate ; hidden at [0][32..]
  unsigned char* const ldx = lds;
  unsigned char* const ldt = lds + G::XBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = __builtin_amdgcn_readfirstlane(wave >> 2), tg = tid & 255;
  if (tid < 4) gate[tid] = 0u;              // (ordered before every use by the barrier behind the tile load)
  // XCD-aware strip order when every XCD gets whole images (then the strips of an image are consecutive in ONE XCD's dispatch order and
  // the exchange argument at the top of this file holds per XCD: needs ceil(H/6) <= 32 CUs); identity otherwise
  const int nwg = gridDim.x;
  const bool remap = ((nwg & 7) == 0) && (((nwg >> 3) % a.ns) == 0) && (a.ns <= 32);
  const int strip = remap ? xcd_strip(blockIdx.x, nwg) : (int)blockIdx.x;
  // strip -> (image, strip of the image) -> (column tile, strip row), strip rows fastest (conv_block.hip)
  const int n = strip / a.ns, si = strip - n * a.ns;
  int sy = si, ct = 0;
  if (G::CT) { ct = si / a.sy_n; sy = si - ct * a.sy_n; }
  const int x0 = ct * OW;                  // image column of the strip's first output column
  const unsigned tag = (*a.epoch << 12) + a.seq;

  // ---- phase 0: input rows 6sy-2 .. 6sy+7, columns -1 .. 48 -> LDS (branch-free loads, zero outside the image) ----
  uint4 T2[BWD ? G::SREGS : 1];
  float mw1[2], mw2[2], mv = 0.f;          // MLP operands of this thread: requested first, written to LDS behind the tile
  {
    const int mtot = a.cr * 64;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + BTHREADS * k;     // cr <= 16: at most 1024 weights per matrix
      mw1[k] = a.cw1[i < mtot ? i : 0];
      mw2[k] = a.cw2[i < mtot ? i : 0];
    }
    // one unconditional load per thread (wave-uniform source select, clamped index): conditional loads would turn later waits into vmcnt(0)
    const int which = tid >> 6, c = tid & 63;
    const int cr_c = c < a.cr ? c : 0;
    const float* src = a.cb2; int idx = c;
    if (which == 0) { src = a.cb1; idx = cr_c; }
    else if (which == 2 && a.qgate) { src = a.qgate; idx = n * 64 + c; }
    else if (which == 3) { src = a.gate; idx = n * 64 + c; }             // forward: not used (the launch's own output buffer)
    else if (which == 4) { src = a.hidden; idx = n * a.cr + cr_c; }
    mv = src[idx];
    if (which == 2 && !a.qgate) mv = 1.f;
  }
  bf16x8 F[18];
  {
    uint4 R[G::XREGS];
    const int y0 = sy * SH - 2;
#pragma unroll
    for (int i = 0; i < G::XREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      const int lr = pix / XC, lc = pix - lr * XC;
      const int y = y0 + lr, x = x0 - XH + lc;
      const bool ok = (p < G::XPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
      const int e = ok ? ((n * a.H + y) * a.W + x) * 64 + part * 8 : 0;
      uint4 v = *reinterpret_cast<const uint4*>(a.x + (unsigned)e);
      v = keep_if(v, ok);
      R[i] = v;
    }
    if (BWD) {   // the strip's own rows of the forward conv2 output: piece p = tid + 512 i -> (pixel p >> 3 of 6 x OW, chunk tid & 7)
#pragma unroll
      for (int i = 0; i < G::SREGS; ++i) {
        const int p = tid + BTHREADS * i;
        const int pix = p >> 3, r = pix / OW, col = pix - r * OW;
        const int y = sy * SH + r;
        const bool ok = (p < G::SPIECES) & (y < a.H) & (x0 + col < a.W);
        const int e = ok ? ((n * a.H + y) * a.W + x0 + col) * 64 + (p & 7) * 8 : 0;
        uint4 v = *reinterpret_cast<const uint4*>(a.t2_in + (unsigned)e);
        v = keep_if(v, ok);
        T2[i] = v;
      }
    }
    // the first conv's filter: requested behind the tile (loads return in order) and in front of the tile's waits - its L2-hit latency lies
    // under the tile's (round 5: it used to be requested behind the LDS writes, i.e. after the whole tile had arrived)
    {
      const uint4* wp = a.w1 + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
    }
    if (!G::CT && tid < G::TROWS * 2 * 8) {       // (column tiles: the halo tile of the first phase writes these columns)
      const int row = tid >> 4, side = (tid >> 3) & 1, chunk = tid & 7;
      *reinterpret_cast<uint4*>(ldt + swz(row * TC + side * (TC - 1), chunk)) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < G::XREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      if (p < G::XPIECES) *reinterpret_cast<uint4*>(ldx + swz(pix, part)) = R[i];
    }
    {
      const int mtot = a.cr * 64;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = tid + BTHREADS * k;
        if (i < mtot) {
          sw1[i] = mw1[k];                                  // [r][c] as stored
          sw2t[(i % a.cr) * 64 + i / a.cr] = mw2[k];        // [c][r] -> [r][c]
        }
      }
      const int which = tid >> 6, c = tid & 63;
      if (which == 0) { if (c < RC_MAXR) svec[c] = mv; }      // row 0: bias of conv_du.0 in [0, 16), hidden (backward) in [32, 48)
      else if (which < 4) svec[which * 64 + c] = mv;
      else if (which == 4 && c < RC_MAXR) svec[32 + c] = mv;      // hidden[n][r] behind the (<= 16) bias entries of row 0
    }
  }
  const int c0 = 16 * q + 4 * g;
  const int gpair = 4 * (g & ~1);
  const int chunk8 = 2 * q + (gpair >> 3);
  __syncthreads();

  if (BWD) {
    // ---- phase 0b: ds = sum over the strip of dy * t2 per channel -> all strips of the image -> MLP backward -> d_t2 in place ----
    float part8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < G::SREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, r = pix / OW, col = pix - r * OW;
      if (p < G::SPIECES) {
        float d[8], t[8];
        unpack8<FMT>(*reinterpret_cast<const uint4*>(ldx + swz((r + 2) * XC + col + XH, tid & 7)), d);
        unpack8<FMT>(T2[i], t);
#pragma unroll
        for (int j = 0; j < 8; ++j) part8[j] = fmaf(d[j], t[j], part8[j]);
      }
    }
    // threads with the same chunk (tid & 7) hold partial sums of the same 8 channels: [k = tid >> 3][chunk][8] in the (still unused)
    // T image, then 256 threads add 16 k's each, then 64 threads add the 4 parts - fixed order
    float* red = reinterpret_cast<float*>(ldt);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(tid >> 3) * 64 + (tid & 7) * 8 + j] = part8[j];
    __syncthreads();
    if (tid < 256) {
      const int c = tid & 63, part = tid >> 6;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += red[(part * 16 + k) * 64 + c];
      red[64 * 64 + part * 64 + c] = s;
    }
    __syncthreads();
    float mine = 0.f;
    if (tid < 64) mine = (red[64 * 64 + tid] + red[64 * 64 + 64 + tid]) + (red[64 * 64 + 128 + tid] + red[64 * 64 + 192 + tid]);
    __syncthreads();                                        // red is dead: the border columns of the T image are rewritten below
    if (!G::CT && tid < G::TROWS * 2 * 8) {
      const int row = tid >> 4, side = (tid >> 3) & 1, chunk = tid & 7;
      *reinterpret_cast<uint4*>(ldt + swz(row * TC + side * (TC - 1), chunk)) = make_uint4(0, 0, 0, 0);
    }
    const float ds = strip_allsum(a, mine, n, si, tid, tag, sx);
    if (tid < 64) {
      const int c = tid;
      const float s = svec[3 * 64 + c];
      const float gq = svec[2 * 64 + c];
      const float dz = (ds * gq) * s * (1.f - s);
      float dp = 0.f;
      for (int r0 = 0; r0 < a.cr; r0 += 4) {          // four hidden units per round, as in the forward launch
        float dhs[4], w1[4], hid[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = (r0 + i < a.cr) ? r0 + i : r0;
          dhs[i] = sw2t[r * 64 + c] * dz; w1[i] = sw1[r * 64 + c]; hid[i] = svec[32 + r];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dhs[i] = wave_sum(dhs[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (r0 + i < a.cr) {
            const float dh = (hid[i] > 0.f) ? dhs[i] : 0.f;
            dp = fmaf(w1[i], dh, dp);
          }
        }
      }
      sgate[c] = s * gq;
      sdp[c] = dp * a.inv_hw;
      if (si == 0) {
        a.dz[n * 64 + c] = dz;
        if (a.dzq) a.dzq[n * 64 + c] = (ds * s) * gq * (1.f - gq);
      }
    }
    __syncthreads();
    // d_t2 = dy * gate + dp / HW on every pixel of the tile that lies inside the image (outside stays the zero padding);
    // the strip's own rows also go to HBM: conv2's weight gradient reads them
    {
      const int y0 = sy * SH - 2;
      const float4 ga = *reinterpret_cast<const float4*>(sgate + (tid & 7) * 8), gb = *reinterpret_cast<const float4*>(sgate + (tid & 7) * 8 + 4);
      const float4 pa = *reinterpret_cast<const float4*>(sdp + (tid & 7) * 8), pb = *reinterpret_cast<const float4*>(sdp + (tid & 7) * 8 + 4);
#pragma unroll
      for (int i = 0; i < G::XREGS; ++i) {
        const int p = tid + BTHREADS * i;
        const int pix = p >> 3, part = p & 7;
        const int lr = pix / XC, lc = pix - lr * XC;
        const int y = y0 + lr, x = x0 - XH + lc;
        const bool ok = (p < G::XPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
        if (ok) {
          uint4* cell = reinterpret_cast<uint4*>(ldx + swz(pix, part));
          float d[8];
          unpack8<FMT>(*cell, d);
          const uint2 lo = pack4<FMT>(fmaf(d[0], ga.x, pa.x), fmaf(d[1], ga.y, pa.y), fmaf(d[2], ga.z, pa.z), fmaf(d[3], ga.w, pa.w));
          const uint2 hi = pack4<FMT>(fmaf(d[4], gb.x, pb.x), fmaf(d[5], gb.y, pb.y), fmaf(d[6], gb.z, pb.z), fmaf(d[7], gb.w, pb.w));
          const uint4 o = make_uint4(lo.x, lo.y, hi.x, hi.y);
          *cell = o;
          if (lr >= 2 && lr < 2 + SH && (!G::CT || (lc >= XH && lc < XH + OW)))      // the strip's own pixels (halo columns belong to the neighbours)
            st16_nt(a.t2 + (unsigned)(((n * a.H + y) * a.W + x) * 64 + part * 8), o);     // 8 lanes per pixel: whole lines
        }
      }
    }
    __syncthreads();
  }

  // ---- phase 1: T rows j = 4rh .. 4rh+3 (image rows 6sy-1+j) from input rows j .. j+2 ----
  unsigned moff[NP1];
  uint4 M[(BWD && !MB) ? NP1 : 1];
  unsigned MBY[(BWD && MB) ? NP1 : 1];
#pragma unroll
  for (int k = 0; k < NP1; ++k) {
    const int jr = (k < TR) ? k : (2 * (k - TR) + (g & 1)), c = (k < TR) ? (g & 1) : 2;
    const int y = sy * SH - 1 + TR * rh + jr, xx = x0 + 16 * c + px;
    const bool in = ((unsigned)y < (unsigned)a.H) & (xx < a.W);
    moff[k] = in ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
    if (BWD && !MB) M[(BWD && !MB) ? k : 0] = *reinterpret_cast<const uint4*>(a.mask + (in ? moff[k] : 0u));
    if (BWD && MB) MBY[(BWD && MB) ? k : 0] = a.mbits[(in ? moff[k] : 0u) >> 3];
  }
  // column tiles: this lane's pixel of the halo tile = T row hj, halo side px & 1 (conv_block.hip)
  const int hj = TR * rh + ((px >> 1) < TR ? (px >> 1) : TR - 1), htc = (px & 1) ? TC - 1 : 0;
  unsigned hoffe = 0xffffffffu;
  uint2 HM = make_uint2(0, 0);
  unsigned HB = 0;
  if (G::CT) {
    const int y = sy * SH - 1 + hj, xx = x0 - 1 + htc;
    if (((unsigned)y < (unsigned)a.H) & ((unsigned)xx < (unsigned)a.W)) hoffe = (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0);
    if (BWD && !MB) HM = *reinterpret_cast<const uint2*>(a.mask + (hoffe != 0xffffffffu ? hoffe : 0u));
    if (BWD && MB) HB = a.mbits[(hoffe != 0xffffffffu ? hoffe : 0u) >> 3];
  }
  {
    f32x4 acc[TR][NC];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!BWD) { const float4 t = *reinterpret_cast<const float4*>(a.b1 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[r][c] = b4;
    unsigned off[8][2];
    if (G::CT) {
      sweep_bases<XC>(off, 0u, hj, 0, g, htc);
      f32x4 th = halo_sweep<FMT, XC>(b4, F, lds, off);
      if (!BWD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) th[j] = relu_f32(th[j]);
      }
      uint2 o = make_uint2(0, 0);
      if (hoffe != 0xffffffffu) {
        o = pack4<FMT>(th[0], th[1], th[2], th[3]);
        if (BWD && !MB) o = make_uint2(o.x & relu_keep(HM.x), o.y & relu_keep(HM.y));
        if (BWD && MB) {
          const uint4 m4 = relu_mask_bits(make_uint4(o.x, o.y, 0, 0), HB >> (4 * (g & 1)));
          o = make_uint2(m4.x, m4.y);
        }
      }
      if (px < 2 * TR) *reinterpret_cast<uint2*>(ldt + swz(hj * TC + htc, 2 * q + (g >> 1)) + (g & 1) * 8) = o;
    }
    sweep_bases<XC>(off, 0u, TR * rh, px, g, G::CT ? 1 : 0);
    block_sweep<TR, FMT, NoHook, NC, XC>(acc, F, lds, off);
    {
      const uint4* wp = a.w2 + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
    }
#pragma unroll
    for (int k = 0; k < NP1; ++k) {
      const f32x4 tx = (k < TR) ? acc[k < TR ? k : 0][0] : acc[2 * (k < TR ? 0 : k - TR)][NC - 1];
      const f32x4 ty = (k < TR) ? acc[k < TR ? k : 0][1] : acc[2 * (k < TR ? 0 : k - 4) + 1][NC - 1];
      float v[8];
      pair_up(tx, ty, g, v);
      if (!BWD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = relu_f32(v[j]);
      }
      const int jr = (k < TR) ? k : (2 * (k - TR) + (g & 1)), c = (k < TR) ? (g & 1) : 2;
      const int j = TR * rh + jr, xx = 16 * c + px;
      uint4 o = make_uint4(0, 0, 0, 0);
      if (moff[k] != 0xffffffffu) {
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (BWD && !MB) o = relu_mask_packed(o, M[(BWD && !MB) ? k : 0]);
        if (BWD && MB) o = relu_mask_bits(o, MBY[(BWD && MB) ? k : 0]);
      }
      *reinterpret_cast<uint4*>(ldt + swz(j * TC + xx + 1, chunk8)) = o;
    }
    gate_arrive(&gate[rh], lane);          // this wave's 16 channels of T rows 4rh .. 4rh+3 are in LDS
  }
  unsigned soffg[G::GREGS];                // element offsets of this thread's 16-byte pieces of its row half's 3 strip rows (T; backward: OUT)
#pragma unroll
  for (int i = 0; i < G::GREGS; ++i) soffg[i] = group_piece_off<G>(i, tg, rh, n, sy, a.H, a.W, x0);
  // no workgroup barrier between the phases: each row half waits for exactly the T rows it reads (conv_block.hip, block_common.hpp)
  gate_wait(&gate[rh], 4u);
  if (rh == 1) gate_wait(&gate[0], 4u);
  // the row half's own strip rows of T (forward: + their ReLU mask bytes) go to HBM from the finished LDS image: whole lines, non-temporal,
  // one piece after every third MFMA group of the second sweep (block_common.hpp::strip_stage; conv_block.hip)
  static_assert(G::SREGS >= G::GREGS, "S holds a row half's pieces and, later, the whole strip's");
  uint4 S[G::SREGS];
  unsigned soff[G::SREGS];                 // (forward) element offsets of this thread's pieces of the whole strip: t2 and OUT stores
  const bool t_out = a.t != nullptr;
  if (t_out) group_stage<1, G>(*reinterpret_cast<uint4(*)[G::GREGS]>(&S[0]), ldt, tg, rh);
  auto t_store = [&](int grp) {           // grp is a constant after unrolling
    if (grp % 3 == 0 && grp / 3 < G::GREGS) {
      const int i = grp / 3 < G::GREGS ? grp / 3 : 0;
      if (t_out && soffg[i] != 0xffffffffu) {
        st16_nt(a.t + soffg[i], S[i]);
        if (!BWD && a.mbits) a.mbits[soffg[i] >> 3] = (unsigned char)relu_bits(S[i]);
      }
    }
  };

  // ---- phase 2: rows 3rh .. 3rh+2 of the strip from T rows r .. r+2 ----
  {
    f32x4 acc[OR][NC];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!BWD) { const float4 t = *reinterpret_cast<const float4*>(a.b2 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < OR; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[r][c] = b4;
    // backward: the residual operand dy (its tile in LDS now holds d_t2) is requested before the sweep and lands under it.  (Taking it
    // from the LDS tile before the transform - 18 more live registers - measured 0.7 % slower on the RCAN step: this read hits L2 / MALL.)
    unsigned ooff[NP2], osoff = 0xffffffffu;
    uint4 P1p[BWD ? NP2 : 1];
    uint2 P1s = make_uint2(0, 0);
#pragma unroll
    for (int k = 0; k < NP2; ++k) {
      const int r = (k < OR) ? k : (g & 1), c = (k < OR) ? (g & 1) : 2;
      const int y = sy * SH + OR * rh + r, xx = x0 + 16 * c + px;
      ooff[k] = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
      if (BWD) P1p[k] = *reinterpret_cast<const uint4*>(a.x + (ooff[k] != 0xffffffffu ? ooff[k] : 0u));
    }
    if (NC == 3) {
      const int y = sy * SH + OR * rh + 2, xx = x0 + 32 + px;
      osoff = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0) : 0xffffffffu;
      if (BWD) P1s = *reinterpret_cast<const uint2*>(a.x + (osoff != 0xffffffffu ? osoff : 0u));
    }
    unsigned off[8][2];
    if (rh == 0) {
      sweep_bases<TC>(off, (unsigned)G::XBYTES, 0, px, g);
      block_sweep<OR - 1, FMT, decltype(t_store), NC, TC>(*reinterpret_cast<f32x4(*)[OR - 1][NC]>(&acc[0]), F, lds, off, t_store);   // rows 0, 1 <- T rows 0 .. 3 (this half's own)
      gate_wait(&gate[1], 4u);
      sweep_bases<TC>(off, (unsigned)G::XBYTES, OR - 1, px, g);
      block_sweep<1, FMT, NoHook, NC, TC>(*reinterpret_cast<f32x4(*)[1][NC]>(&acc[OR - 1]), F, lds, off);                        // row 2 <- T rows 2 .. 4
    } else {
      sweep_bases<TC>(off, (unsigned)G::XBYTES, OR, px, g);
      block_sweep<OR, FMT, decltype(t_store), NC, TC>(acc, F, lds, off, t_store);                                           // rows 3 .. 5 <- T rows 3 .. 7
    }
    // pairs k < 3: X = (row k, col 0), Y = (row k, col 1); k = 3: X = (0, 2), Y = (1, 2); single: (2, 2)
    float V[NP2][8], vs[4];
#pragma unroll
    for (int k = 0; k < NP2; ++k) {
      const f32x4 tx = (k < OR) ? acc[k < OR ? k : 0][0] : acc[0][NC - 1];
      const f32x4 ty = (k < OR) ? acc[k < OR ? k : 0][1] : acc[1][NC - 1];
      pair_up(tx, ty, g, V[k]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) vs[j] = acc[NC == 3 ? 2 : 0][NC - 1][j];      // (NC = 3 only: the single tile; osoff stays "outside" otherwise)

    if (BWD) {
      // dx = dy + conv1^T(gt1)
#pragma unroll
      for (int k = 0; k < NP2; ++k) {
        if (ooff[k] != 0xffffffffu) {
          float m[8];
          unpack8<FMT>(P1p[BWD ? k : 0], m);
          if (a.res2) {
            float e[8];
            unpack8<FMT>(*reinterpret_cast<const uint4*>(a.res2 + ooff[k]), e);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] += e[j];
          }
          const uint2 lo = pack4<FMT>(V[k][0] + m[0], V[k][1] + m[1], V[k][2] + m[2], V[k][3] + m[3]);
          const uint2 hi = pack4<FMT>(V[k][4] + m[4], V[k][5] + m[5], V[k][6] + m[6], V[k][7] + m[7]);
          const int r = (k < OR) ? k : (g & 1), c = (k < OR) ? (g & 1) : 2;
          *reinterpret_cast<uint4*>(ldx + swz((OR * rh + r + 2) * XC + 16 * c + px + XH, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);   // dx image in place of the
        }                                                                                                                                // d_t2 tile (dead in phase 2)
      }
      if (osoff != 0xffffffffu) {
        float m[4];
        unpack4<FMT>(P1s, m);
        if (a.res2) {
          float e[4];
          unpack4<FMT>(*reinterpret_cast<const uint2*>(a.res2 + osoff), e);
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] += e[j];
        }
        *reinterpret_cast<uint2*>(ldx + swz((OR * rh + 2 + 2) * XC + 32 + px + XH, 2 * q + (g >> 1)) + (g & 1) * 8) =
            pack4<FMT>(vs[0] + m[0], vs[1] + m[1], vs[2] + m[2], vs[3] + m[3]);
      }
    } else {
      // t2 = conv2(t1) + b2: channel sums of the strip for the attention pool, t2 itself to HBM when training
      float ps8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NP2; ++k) {
        if (ooff[k] != 0xffffffffu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) ps8[j] += V[k][j];
        }
      }
      if (osoff != 0xffffffffu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ps[j] += vs[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = row16_sum(ps8[j]);
        t += lane_xor16(t, g);
        ps8[j] = t;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = row16_sum(ps[j]);
        const float up = lane_xor16(t, g);
        ps8[j] += (g & 1) ? up : t;
        ps8[4 + j] += (g & 1) ? t : up;
      }
      if (px == 0 && !(g & 1)) {
        float* pp = spool + rh * 64 + 16 * q + 4 * g;
        *reinterpret_cast<float4*>(pp) = make_float4(ps8[0], ps8[1], ps8[2], ps8[3]);
        *reinterpret_cast<float4*>(pp + 4) = make_float4(ps8[4], ps8[5], ps8[6], ps8[7]);
      }
      __syncthreads();                     // every wave has finished its second sweep: the T image is dead
#pragma unroll
      for (int i = 0; i < G::SREGS; ++i) soff[i] = strip_piece_off<G>(i, tid, n, sy, a.H, a.W, x0);
      if (a.t2) {                          // training: t2 = conv2(t1) + b2 goes to HBM through the T image's rows 1 .. 6 (whole lines, below)
#pragma unroll
        for (int k = 0; k < NP2; ++k) {
          if (ooff[k] != 0xffffffffu) {
            const int r = (k < OR) ? k : (g & 1), c = (k < OR) ? (g & 1) : 2;
            const uint2 lo = pack4<FMT>(V[k][0], V[k][1], V[k][2], V[k][3]), hi = pack4<FMT>(V[k][4], V[k][5], V[k][6], V[k][7]);
            *reinterpret_cast<uint4*>(ldt + swz((OR * rh + r + 1) * TC + 16 * c + px + 1, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
        }
        if (osoff != 0xffffffffu)
          *reinterpret_cast<uint2*>(ldt + swz((OR * rh + 2 + 1) * TC + 32 + px + 1, 2 * q + (g >> 1)) + (g & 1) * 8) = pack4<FMT>(vs[0], vs[1], vs[2], vs[3]);
      }
      const float mine = (tid < 64) ? spool[tid] + spool[64 + tid] : 0.f;
      const float tot = strip_allsum(a, mine, n, si, tid, tag, sx);      // (its barriers also complete the t2 image)
      if (a.t2) strip_stage<1, G>(S, ldt, tid);
      if (tid < 64) {
        const int c = tid;
        const float mean = tot * a.inv_hw;
        float z = svec[64 + c];
        for (int r0 = 0; r0 < a.cr; r0 += 4) {        // four hidden units per round (cr = 4 for the reference's reduction 16): their LDS operands are
          float hs[4], w2[4], b1[4];                    // requested together and their four wave sums are independent instruction chains
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = (r0 + i < a.cr) ? r0 + i : r0;
            hs[i] = sw1[r * 64 + c] * mean; w2[i] = sw2t[r * 64 + c]; b1[i] = svec[r];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) hs[i] = wave_sum(hs[i]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (r0 + i < a.cr) {
              const float h = fmaxf(hs[i] + b1[i], 0.f);
              z = fmaf(w2[i], h, z);
              if (si == 0 && c == 0) a.hidden[n * a.cr + r0 + i] = h;
            }
          }
        }
        const float gt = 1.f / (1.f + expf(-z));
        sgate[c] = gt * svec[2 * 64 + c];
        if (si == 0) { a.mean[n * 64 + c] = mean; a.gate[n * 64 + c] = gt; }
      }
      __syncthreads();
      if (a.t2) {
#pragma unroll
        for (int i = 0; i < G::SREGS; ++i)
          if (soff[i] != 0xffffffffu) st16_nt(a.t2 + soff[i], S[i]);
      }
      // out = x + gate * t2, the residual operand from the input tile in LDS; the result replaces it there
      const float4 ga = *reinterpret_cast<const float4*>(sgate + 16 * q + gpair), gb = *reinterpret_cast<const float4*>(sgate + 16 * q + gpair + 4);
#pragma unroll
      for (int k = 0; k < NP2; ++k) {
        if (ooff[k] != 0xffffffffu) {
          const int r = (k < OR) ? k : (g & 1), c = (k < OR) ? (g & 1) : 2;
          const int srow = OR * rh + r, xx = 16 * c + px;
          float m[8];
          unpack8<FMT>(*reinterpret_cast<const uint4*>(ldx + swz((srow + 2) * XC + xx + XH, chunk8)), m);
          const uint2 lo = pack4<FMT>(fmaf(V[k][0], ga.x, m[0]), fmaf(V[k][1], ga.y, m[1]), fmaf(V[k][2], ga.z, m[2]), fmaf(V[k][3], ga.w, m[3]));
          const uint2 hi = pack4<FMT>(fmaf(V[k][4], gb.x, m[4]), fmaf(V[k][5], gb.y, m[5]), fmaf(V[k][6], gb.z, m[6]), fmaf(V[k][7], gb.w, m[7]));
          *reinterpret_cast<uint4*>(ldx + swz((srow + 2) * XC + xx + XH, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      }
      if (osoff != 0xffffffffu) {
        const int srow = OR * rh + 2, xx = 32 + px;
        float m[4];
        unpack4<FMT>(*reinterpret_cast<const uint2*>(ldx + swz((srow + 2) * XC + xx + XH, 2 * q + (g >> 1)) + (g & 1) * 8), m);
        const float4 gs = *reinterpret_cast<const float4*>(sgate + c0);
        *reinterpret_cast<uint2*>(ldx + swz((srow + 2) * XC + xx + XH, 2 * q + (g >> 1)) + (g & 1) * 8) =
            pack4<FMT>(fmaf(vs[0], gs.x, m[0]), fmaf(vs[1], gs.y, m[1]), fmaf(vs[2], gs.z, m[2]), fmaf(vs[3], gs.w, m[3]));
      }
    }
  }
  // ---- OUT (forward: x + gate * t2; backward: dx): the image sits in LDS in place of the input tile's centre rows -> whole lines, non-temporal ----
  if (BWD) {                               // per row half: its 3 rows are complete when its 4 waves have arrived
    gate_arrive(&gate[2 + rh], lane);
    gate_wait(&gate[2 + rh], 4u);
    group_stage<2, G>(*reinterpret_cast<uint4(*)[G::GREGS]>(&S[0]), ldx, tg, rh);
#pragma unroll
    for (int i = 0; i < G::GREGS; ++i)
      if (soffg[i] != 0xffffffffu) st16_nt(a.out + soffg[i], S[i]);
  } else {
    __syncthreads();
    strip_stage<2, G>(S, ldx, tid);
#pragma unroll
    for (int i = 0; i < G::SREGS; ++i)
      if (soff[i] != 0xffffffffu) st16_nt(a.out + soff[i], S[i]);
  }
}

__global__ void rcab_epoch_kernel(unsigned* epoch) { *epoch += 1u; }

template <class G>
static void rcab_dispatch(const rumpy_rcab_args* p, const RcabDev& d, hipStream_t s, bool bwd) {
  const dim3 grid(d.N * d.ns);
  if (bwd && p->maskbits) RUMPY_LAUNCH_PROBED(5, (rcab_kernel<true, true, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
  else if (bwd) RUMPY_LAUNCH_PROBED(5, (rcab_kernel<true, false, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
  else if (p->fmt == RUMPY_FMT_F16) RUMPY_LAUNCH_PROBED(5, (rcab_kernel<false, false, RUMPY_FMT_F16, G>), grid, dim3(BTHREADS), s, d);
  else RUMPY_LAUNCH_PROBED(5, (rcab_kernel<false, false, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
}

// Geometry of a launch on [N, H, W] (round 4; conv_block.hip::block_geometry): strips of 4 / 6 / 8 rows x column tiles, chosen so that the
// workgroup count fits the CUs - 8 crops of 64 x 64 (the reference's div2k/rcan.toml batch) are 8 x 16 x 2 = 256 workgroups of 4 rows, where
// 6-row strips leave 80 CUs idle.  A workgroup's fixed part (exchange, MLP, gate, two more barriers) is priced at 12 rows' worth of MFMA work.
static void rcab_geometry(int N, int H, int W, bool rows_ok, int* sh, int* nc, int* ct_n) {
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
    if (((H + h - 1) / h) * ct > cus) continue;           // every strip of an image has to be resident
    const long wgs = (long)N * ((H + h - 1) / h) * ct;
    const long cost = ((wgs + cus - 1) / cus) * (2 * h + 2 + 12) * (c + 1);
    if (best < 0 || cost < best) { best = cost; *sh = h; *nc = c; *ct_n = ct; }
  }
}

template <int SH>
static void rcab_dispatch_rows(const rumpy_rcab_args* p, const RcabDev& d, hipStream_t s, bool bwd) {
  typedef BlockGeo<2, true, SH> G;
  const dim3 grid(d.N * d.ns);
  if (bwd) RUMPY_LAUNCH_PROBED(5, (rcab_kernel<true, true, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
  else if (p->fmt == RUMPY_FMT_F16) RUMPY_LAUNCH_PROBED(5, (rcab_kernel<false, false, RUMPY_FMT_F16, G>), grid, dim3(BTHREADS), s, d);
  else RUMPY_LAUNCH_PROBED(5, (rcab_kernel<false, false, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
}

// upper bound of the strips (workgroups) of one image over the geometries a launch may take: strip rows x column tiles; all of them exchange
// their pool sums, so all must be resident together (the host sizes the exchange buffer and checks residency with it)
extern "C" int rumpy_rcab_strips(int32_t H, int32_t W) {
  int nc, ct_n;
  block_col_tiles(W, &nc, &ct_n);
  int n = ((H + BSH - 1) / BSH) * ct_n;
  if (W > BSW) {
    const int ct2 = (W + 31) / 32;
    n = max(n, ((H + BSH - 1) / BSH) * ct2);
  }
  return n;
}

int rumpy_rcab_fp8_launch(const rumpy_rcab_args* p, const RcabDev& d0, hipStream_t s, bool bwd, const char* what);      // conv_rcab_fp8.hip

static int rcab_launch(const rumpy_rcab_args* p, void* stream, bool bwd, const char* what) {
  if (!p || !p->x || !p->w1 || !p->w2 || !p->out || !p->ca_w1 || !p->ca_b1 || !p->ca_w2 || !p->ca_b2 || !p->hidden || !p->gate ||
      !p->xchg || !p->epoch || !p->status) { rumpy_set_error("%s: null pointer", what); return RUMPY_E_ARG; }
  if (!bwd && (!p->b1 || !p->b2 || !p->mean)) { rumpy_set_error("%s: forward needs b1, b2, mean", what); return RUMPY_E_ARG; }
  if (bwd && (!p->t2_in || (!p->mask && !p->maskbits) || !p->t2 || !p->dz)) { rumpy_set_error("%s: backward needs t2_in, mask, t2 (d_t2 out), dz", what); return RUMPY_E_ARG; }
  if (bwd && p->dzq && !p->qgate) { rumpy_set_error("%s: dzq without qgate", what); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0) { rumpy_set_error("%s: bad shape", what); return RUMPY_E_ARG; }
  // strips of 4 / 8 rows: forward (bf16 / fp16) and the mask-byte backward form, when the 4-row strips of an image still fit the CUs and the
  // exchange buffer (sized for rumpy_rcab_strips) holds them
  int nc, ct_n, sh;
  const int ns4 = ((p->H + 3) / 4) * ((p->W + 31) / 32);
  const bool rows_ok = !p->w1_f8 && (!bwd || p->maskbits) && ns4 <= rumpy_device_cus() && (int64_t)p->N * ns4 * 64 * 8 <= p->xchg_bytes;
  rcab_geometry(p->N, p->H, p->W, rows_ok, &sh, &nc, &ct_n);
  const int sy_n = (p->H + sh - 1) / sh;
  const int ns = sy_n * ct_n;
  if (p->cr <= 0 || p->cr > RC_MAXR || ns > rumpy_device_cus() || p->seq >= 4096u || (int64_t)p->N * p->H * p->W * 64 >= (int64_t)0xffffffffu) {
    rumpy_set_error("%s: needs strips per image = ceil(H/6) * column tiles <= CUs, 0 < Cr <= 16, seq < 4096 (W=%d H=%d strips=%d Cr=%d seq=%u)", what, p->W, p->H, ns, p->cr, p->seq); return RUMPY_E_ARG; }
  if (p->fmt != RUMPY_FMT_BF16 && !(p->fmt == RUMPY_FMT_F16 && !bwd)) { rumpy_set_error("%s: fmt %d is a forward-only format", what, p->fmt); return RUMPY_E_ARG; }
  const int64_t need = (int64_t)p->N * ns * 64 * 8;
  if (p->xchg_bytes < need) { rumpy_set_error("%s: exchange buffer too small (%lld < %lld)", what, (long long)p->xchg_bytes, (long long)need); return RUMPY_E_ARG; }
  RcabDev d;
  d.x = (const uint16_t*)p->x; d.w1 = (const uint4*)p->w1; d.b1 = p->b1; d.w2 = (const uint4*)p->w2; d.b2 = p->b2;
  d.t = (uint16_t*)p->t; d.t2 = (uint16_t*)p->t2; d.t2_in = (const uint16_t*)p->t2_in; d.mask = (const uint16_t*)p->mask; d.res2 = (const uint16_t*)p->res2; d.out = (uint16_t*)p->out;
  d.N = p->N; d.H = p->H; d.W = p->W; d.sy_n = sy_n; d.ct_n = ct_n; d.ns = ns;
  d.cw1 = p->ca_w1; d.cb1 = p->ca_b1; d.cw2 = p->ca_w2; d.cb2 = p->ca_b2; d.cr = p->cr; d.inv_hw = 1.0f / ((float)p->H * (float)p->W);
  d.mean = p->mean; d.hidden = p->hidden; d.gate = p->gate; d.qgate = p->qgate; d.dz = p->dz; d.dzq = p->dzq;
  d.xchg = (unsigned long long*)p->xchg; d.xchg_bytes = (unsigned)need; d.epoch = (const unsigned*)p->epoch; d.seq = p->seq; d.status = (unsigned*)p->status; d.mbits = (unsigned char*)p->maskbits;
  hipStream_t s = (hipStream_t)stream;
  if (p->w1_f8) {                          // precision 'fp8' (conv_rcab_fp8.hip)
    const int rc = rumpy_rcab_fp8_launch(p, d, s, bwd, what);
    return rc ? rc : rumpy_check_launch(what);
  }
  if (sh == 8) rcab_dispatch_rows<8>(p, d, s, bwd);
  else if (sh == 4) rcab_dispatch_rows<4>(p, d, s, bwd);
  else if (p->W <= BSW) rcab_dispatch<GeoL>(p, d, s, bwd);
  else if (nc == 3) rcab_dispatch<BlockGeo<3, true> >(p, d, s, bwd);
  else rcab_dispatch<BlockGeo<2, true> >(p, d, s, bwd);
  return rumpy_check_launch(what);
}

extern "C" int64_t rumpy_rcab_xchg_bytes(int32_t N, int32_t H, int32_t W) {
  const int64_t ns4 = W > BSW ? (int64_t)((H + 3) / 4) * ((W + 31) / 32) : 0;          // room for the 4-row geometry as well
  return (int64_t)N * max((int64_t)rumpy_rcab_strips(H, W), ns4) * 64 * 8;
}
extern "C" int rumpy_rcab_fwd(const rumpy_rcab_args* p, void* stream) { return rcab_launch(p, stream, false, "rumpy_rcab_fwd"); }
extern "C" int rumpy_rcab_bwd(const rumpy_rcab_args* p, void* stream) { return rcab_launch(p, stream, true, "rumpy_rcab_bwd"); }
extern "C" int rumpy_rcab_epoch_advance(void* epoch, void* stream) {
  if (!epoch) { rumpy_set_error("rumpy_rcab_epoch_advance: null pointer"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(rcab_epoch_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned*)epoch);
  return rumpy_check_launch("rumpy_rcab_epoch_advance");
}

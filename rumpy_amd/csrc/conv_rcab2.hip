// A residual channel-attention block (RCAB, rumpy/SISR/models/advanced/architectures.py:60-84; QRCAB,
// attention_manipulators/architectures.py:154-228) per launch WITHOUT any exchange between the workgroups of a launch (round 5).
//
// The channel attention gate of block k, s_k = CA(mean_hw(U_k)) with U_k = conv2(relu(conv1(x_k))), needs a sum over the whole image.
// conv_rcab.hip takes that sum inside the launch that produces U_k: the strips of an image post their partial sums in HBM and poll for
// each other's, which makes the launch depend on all strips of an image being resident at once (a watchdog, a shared-GPU fall-back, a
// hazard next to the kernels of a collective) and puts two workgroup barriers, an LDS all-gather and a one-wave MLP behind the second
// sweep.  Nothing forces the sum to be CONSUMED where it is produced:
//
//   forward launch k   : [x_k = x_{k-1} + s_{k-1} * U_{k-1}, formed while the tile is staged; its own rows stored once]
//                        t1 = relu(conv1(x_k) + b1) ; U_k = conv2(t1) + b2 -> HBM UNGATED, + one row of 64 pool partial sums per
//                        (strip, row half), plain stores.  s_{k-1} is evaluated from the PREVIOUS launch's partial rows in the prologue.
//   backward launch k  : G = dL/dx_{k+1}.  ds_k = sum_hw G * U_k arrives as partial rows written by the launch that PRODUCED G (the backward
//                        launch of block k+1: it reduces dx * U_k in its epilogue, U_k's own rows fetched by LDS-DMA at kernel start);
//                        MLP backward in the prologue, dU = G * s_k + dp / HW formed while the tile is staged, then
//                        gt1 = [t1 > 0] . conv2^T(dU) ; dx = G + conv1^T(gt1) [+ res2], and the partial rows sum dx * U_{k-1} for launch k-1.
//
// Every launch is the residual-block kernel (conv_block.hip: row-half groups on LDS counters, whole-line non-temporal stores, no workgroup
// barrier behind the prologue) plus a per-wave, register-only 600-flop MLP that runs while the tile is in flight.  Partial rows are summed
// in row order by every workgroup: all strips of an image use bit-identical gates, run to run.  Chain ends (first / last block of a
// ResidualGroup) use the streaming kernels of ca.hip: rumpy_ca_fwd_fused forms x + s * U for the group's conv, rumpy_ca_bwd_reduce the
// partial rows of the group conv's data gradient.
// One more rounding than conv_rcab.hip: the residual branch is gated from the STORED (bf16 / fp16) U, not from the fp32 accumulators.
#include "rcab_common.hpp"
#include <cstdlib>
#include <cstdio>

struct Rcab2Dev {
  const uint16_t* x;         // fwd: x_{k-1} (or the block input itself when u_in is NULL); bwd: G
  const uint16_t* u_in;      // fwd: U_{k-1} (pending branch) or NULL; bwd: U_{k-1} of the forward pass (own rows only) or NULL
  const float* part_in; int np_in;   // partial rows [N][np_in][64] of the gate evaluated in this launch (fwd: pool sums of u_in; bwd: sum G * U_k)
  float* part_out;           // fwd: pool sums of u_out; bwd: sum dx * u_in - [N][2 * strips][64], or NULL
  const uint4* w1; const float* b1; const uint4* w2; const float* b2;
  uint16_t* x_out;           // fwd: x_k (own rows; with u_in); bwd: dU
  uint16_t* t;               // fwd: t1 or NULL; bwd: gt1
  uint16_t* u_out;           // fwd: U_k; bwd: dx
  const uint16_t* res2; unsigned char* mbits;
  int N, H, W, sy_n, ct_n;
  const float* cw1; const float* cb1; const float* cw2; const float* cb2; int cr; float inv_hw;
  float* mean; float* hidden; float* gate; const float* qgate; float* dz; float* dzq;
};

typedef __attribute__((address_space(3))) unsigned char* r2_lds_u8;
// one LDS-DMA wave-instruction (wgrad_dma.hip::dma16): lane l copies 16 B from its own global address to LDS byte (lds_dst + 16 l)
__device__ __forceinline__ void r2_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// -DRCAB2_STAMPS (measurement builds only, tests/tools/r05_stamps.sh): phase time stamps of every wave (s_memrealtime, 100 MHz) into a buffer
// handed over by rumpy_debug_rcab2_stamps; the results of the launch are unchanged
#ifdef RCAB2_STAMPS
__device__ unsigned long long* g_r2_stamps;
#define R2_STAMP(k) do { if ((threadIdx.x & 63) == 0 && g_r2_stamps) g_r2_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int rumpy_debug_rcab2_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_r2_stamps), &buf, sizeof(buf)); }
#else
#define R2_STAMP(k) do { } while (0)
#endif

template <bool BWD, int FMT = RUMPY_FMT_BF16, class G = GeoL>
__global__ void __launch_bounds__(BTHREADS, 2) rcab2_kernel(Rcab2Dev a) {
  constexpr int NC = G::NC, XC = G::XC, TC = G::TC, XH = G::XH, OW = G::OW;
  constexpr int SH = G::SH, OR = G::OR, TR = G::TR;
  constexpr int NP1 = NC == 3 ? 6 : TR;
  constexpr int NP2 = NC == 3 ? 4 : OR;
  constexpr int UBYTES = BWD ? 2 * G::GREGS * 256 * 16 : 16;      // bwd: the strip's own rows of U_{k-1}, piece (rh, i, tg) at ((rh GREGS + i) 256 + tg) 16
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::XBYTES + G::TBYTES];
  __shared__ __attribute__((aligned(16))) unsigned char ldu[UBYTES];
  __shared__ float red[8 * 64];            // prologue: the waves' shares of the partial rows; bwd tail: [rh][q][64] product sums
  __shared__ unsigned gate[12];            // waves that have written: [0, 1] T rows of row half 0 / 1, [2, 3] OUT rows, [4] the early part of the input tile, [5] row half
                                           // 1: the late part, [6] their share of the partial rows, [8, 9] their product sums (bwd)
  unsigned char* const ldx = lds;
  unsigned char* const ldt = lds + G::XBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = __builtin_amdgcn_readfirstlane(wave >> 2), tg = tid & 255;
  const int strip = xcd_strip(blockIdx.x, gridDim.x);
  int n, sy, ct = 0;
  if (G::CT) { sy = strip % a.sy_n; const int r = strip / a.sy_n; ct = r % a.ct_n; n = r / a.ct_n; }
  else { n = strip / a.sy_n; sy = strip - n * a.sy_n; }
  const int si = ct * a.sy_n + sy;         // strip of the image: partial rows 2 si, 2 si + 1
  const int np_out = 2 * a.sy_n * a.ct_n;
  const int x0 = ct * OW;
  const bool pend = BWD || a.u_in != nullptr;      // a gate is evaluated and applied in this launch
  if (tid < 12) gate[tid] = 0u;
  __syncthreads();                         // (nothing is in flight yet: a bare s_barrier) the counters are zero before anybody arrives
  R2_STAMP(0);

  // ---- phase 0a: the oldest requests of the launch - this wave's share of the partial rows, the MLP's operands of channel `lane` ----
  float pv[8], mw1[4], mw2[4], mu[4] = {0.f, 0.f, 0.f, 0.f}, mb2 = 0.f, mgq = 1.f, msg = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) pv[k] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { mw1[i] = 0.f; mw2[i] = 0.f; }
  if (pend) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {          // rows wave, wave + 8, ... (np_in <= 64, checked by the host); rows beyond np_in are masked where they are added
      const int j = wave + 8 * k;
      pv[k] = a.part_in[((size_t)n * a.np_in + (j < a.np_in ? j : 0)) * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // hidden units 0 .. cr - 1 (cr <= 4)
      const int r = i < a.cr ? i : 0;
      mw1[i] = a.cw1[r * 64 + lane];
      mw2[i] = a.cw2[lane * a.cr + r];
    }
    mb2 = a.cb2[lane];
    if (a.qgate) mgq = a.qgate[n * 64 + lane];
    if (BWD) msg = a.gate[n * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) mu[i] = BWD ? a.hidden[n * a.cr + (i < a.cr ? i : 0)] : a.cb1[i < a.cr ? i : 0];      // (uniform: bias of conv_du.0 | its hidden unit)
  }
  // ---- phase 0b: tile requests: input rows SH sy - 2 .. SH sy + SH + 1, columns x0 - XH .. x0 + OW + XH - 1 (zero outside the image) ----
  // conv_block.hip's split: the pieces of input rows 0 .. TR + 1 (row half 0's window; R0 rounds of 512) are requested by all threads first and
  // announced on their own LDS counter, the rest by row half 1's threads (row half 0 issues as many loads of one cached line instead: every wave
  // runs the same, unconditional load sequence, so the compiler's waits stay counted).  Row half 0 starts its first sweep when rows 0 .. TR + 1 are
  // in LDS, row half 1 when everything is.  With a pending branch every piece is two loads (x and u).
  constexpr int R0 = ((TR + 2) * XC * 8 + BTHREADS - 1) / BTHREADS;
  constexpr int LATE = G::XPIECES - R0 * BTHREADS > 0 ? G::XPIECES - R0 * BTHREADS : 0;
  constexpr int R1 = (LATE + 255) / 256;
  constexpr int RA = R0 + (R1 > 0 ? R1 : 0);
  uint4 R[RA], R2[(!BWD) ? RA : 1];      // (R2: requested and read with a pending branch only)
  const int y0 = sy * SH - 2;
  auto piece_of = [&](int i) -> int { return i < R0 ? tid + BTHREADS * i : R0 * BTHREADS + tg + 256 * (i - R0); };
  // (masks, not selects or branches: with `if (!ok) v = 0` the compiler built an exec-masked block holding an s_waitcnt vmcnt(0) behind EVERY load -
  // taken by every strip, because the halo columns of a W <= 48 image always lie outside it - and the tile arrived one load latency at a time)
  auto piece_elem = [&](int p, bool live, unsigned& m) -> unsigned {
    const int pix = p >> 3, part = p & 7;
    const int lr = pix / XC, lc = pix - lr * XC;
    const int y = y0 + lr, x = x0 - XH + lc;
    const bool ok = live & (p < G::XPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
    m = 0u - (unsigned)ok;
    return (unsigned)(((n * a.H + y) * a.W + x) * 64 + part * 8) & m;
  };
  unsigned RM[RA];                         // all ones where the piece lies inside the image
  if (!BWD && pend) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const unsigned e = piece_elem(piece_of(i), i < R0 || rh == 1, RM[i]);
      R[i] = *reinterpret_cast<const uint4*>(a.x + e);
      R2[(!BWD) ? i : 0] = *reinterpret_cast<const uint4*>(a.u_in + e);
    }
  } else {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const unsigned e = piece_elem(piece_of(i), i < R0 || rh == 1, RM[i]);
      R[i] = *reinterpret_cast<const uint4*>(a.x + e);
    }
  }
  bf16x8 F[18];
  {
    const uint4* wp = a.w1 + (size_t)q * 18 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
  }

  // ---- phase 0c: the gate (every wave by itself, lane = channel; all in registers).  The waves' shares of the partial rows meet through LDS and a
  // counter, not through __syncthreads: that is a full s_waitcnt vmcnt(0) - it would hold the gate back until the whole tile has arrived ----
  float sA[8], pA[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sA[j] = 0.f; pA[j] = 0.f; }
  __builtin_amdgcn_sched_barrier(0);       // every request above is issued before the first wait below (left alone the scheduler put the partial rows' adds - a wait - in front of the tile)
  if (pend) {
#pragma unroll
    for (int k = 0; k < 8; ++k) pv[k] = (wave + 8 * k < a.np_in) ? pv[k] : 0.f;
    red[wave * 64 + lane] = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
    gate_arrive(&gate[6], lane);
    gate_wait(&gate[6], 8u);
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += red[k * 64 + lane];
    const bool wr = (si == 0) && (wave == 0);
    float sg, dpv = 0.f;
    // straight-line over the (at most 4) hidden units, operands requested at kernel start.  Cr <= 4 (the reference's reduction 16 of 64 features) is
    // all this kernel takes: a loop over further units with loads in it made the compiler drain EVERY outstanding load in front of the gate (a full
    // wait at the loop head; and a register that such a load may still target costs the straight-line path an s_waitcnt vmcnt(0) where it is
    // reused).  Wider attention MLPs run conv_rcab.hip (the engine picks the form per block).
    if (!BWD) {
      const float mean = tot * a.inv_hw;
      float z = mb2;
      {
        float hs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) hs[i] = wave_sum(mw1[i] * mean);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < a.cr) {
            const float h = fmaxf(hs[i] + mu[i], 0.f);
            z = fmaf(mw2[i], h, z);
            if (wr && lane == 0) a.hidden[n * a.cr + i] = h;
          }
        }
      }
      const float gt = 1.f / (1.f + expf(-z));
      sg = gt * mgq;
      if (wr) { a.mean[n * 64 + lane] = mean; a.gate[n * 64 + lane] = gt; }
    } else {
      const float ds = tot, s = msg, gq = mgq;
      const float dz = (ds * gq) * s * (1.f - s);
      float dp = 0.f;
      {
        float dhs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dhs[i] = wave_sum(mw2[i] * dz);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < a.cr) {
            const float dh = (mu[i] > 0.f) ? dhs[i] : 0.f;
            dp = fmaf(mw1[i], dh, dp);
          }
        }
      }
      sg = s * gq;
      dpv = dp * a.inv_hw;
      if (wr) {
        a.dz[n * 64 + lane] = dz;
        if (a.dzq) a.dzq[n * 64 + lane] = (ds * s) * gq * (1.f - gq);
      }
    }
    // this thread's pieces are chunk tid & 7 of their pixels: channels 8 (tid & 7) .. + 7
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sA[j] = __shfl(sg, 8 * (tid & 7) + j);
      if (BWD) pA[j] = __shfl(dpv, 8 * (tid & 7) + j);
    }
  }
  R2_STAMP(1);
  // ---- phase 0d: the tile -> LDS, gated on the way; its own pixels -> HBM (whole lines: 8 lanes per pixel) ----
  if (!G::CT && tid < G::TROWS * 2 * 8) {
    const int row = tid >> 4, side = (tid >> 3) & 1, chunk = tid & 7;
    *reinterpret_cast<uint4*>(ldt + swz(row * TC + side * (TC - 1), chunk)) = make_uint4(0, 0, 0, 0);
  }
  auto stage_piece = [&](int i) {           // i is a constant after unrolling
    const int p = piece_of(i);
    const int pix = p >> 3, part = p & 7;
    const unsigned m = RM[i];
    uint4 o = make_uint4(R[i].x & m, R[i].y & m, R[i].z & m, R[i].w & m);
    if (pend) {
      unsigned m2;
      const unsigned e = piece_elem(p, true, m2);
      if (m) {
        float d[8];
        unpack8<FMT>(R[i], d);
        if (!BWD) {
          float u[8];
          unpack8<FMT>(R2[(!BWD) ? i : 0], u);
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j] = fmaf(u[j], sA[j], d[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j] = fmaf(d[j], sA[j], pA[j]);
        }
        const uint2 lo = pack4<FMT>(d[0], d[1], d[2], d[3]), hi = pack4<FMT>(d[4], d[5], d[6], d[7]);
        o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        const int lr = pix / XC, lc = pix - lr * XC;
        if (lr >= 2 && lr < 2 + SH && lc >= XH && lc < XH + OW) st16_nt(a.x_out + e, o);
      }
    }
    if (p < G::XPIECES) *reinterpret_cast<uint4*>(ldx + swz(pix, part)) = o;
  };
#pragma unroll
  for (int i = 0; i < R0; ++i) stage_piece(i);
  gate_arrive(&gate[4], lane);             // this wave's pieces of input rows 0 .. TR + 1 (and a bit) are in LDS
  if (R1 > 0 && rh == 1) {
#pragma unroll
    for (int i = R0; i < RA; ++i) stage_piece(i);
    gate_arrive(&gate[5], lane);
  }
  const int c0 = 16 * q + 4 * g;
  const int gpair = 4 * (g & ~1);
  const int chunk8 = 2 * q + (gpair >> 3);
  unsigned soff[G::GREGS];                 // element offsets of this thread's pieces of its row half's strip rows (T, OUT stores; bwd: the U_{k-1} pieces)
#pragma unroll
  for (int i = 0; i < G::GREGS; ++i) soff[i] = group_piece_off<G>(i, tg, rh, n, sy, a.H, a.W, x0);
  const bool prod = BWD && a.u_in != nullptr && a.part_out != nullptr;
  R2_STAMP(2);
  gate_wait(&gate[4], 8u);                 // input rows 0 .. TR + 1: all eight waves' early pieces
  if (R1 > 0 && rh == 1) gate_wait(&gate[5], 4u);   // the rest: row half 1's own late pieces
  R2_STAMP(3);
  if (BWD && prod) {
    // the forward pass's U_{k-1}, own rows: needed behind the last sweep - by LDS-DMA, no registers; requested here (not in front of the tile: it
    // would delay it) and complete at the start of phase 2 (the wait for the second filter, which is younger, covers it)
    const unsigned ubase = (unsigned)(size_t)(r2_lds_u8)ldu;
#pragma unroll
    for (int i = 0; i < G::GREGS; ++i)
      r2_dma16((const void*)(a.u_in + (soff[i] != 0xffffffffu ? soff[i] : 0u)), __builtin_amdgcn_readfirstlane(ubase + ((rh * G::GREGS + i) * 256 + 64 * q) * 16));
  }

  // ---- phase 1: T rows TR rh .. TR rh + TR - 1 (image rows SH sy - 1 + j) from input rows j .. j + 2 ----
  unsigned moff[NP1];
  unsigned MB[BWD ? NP1 : 1];
#pragma unroll
  for (int k = 0; k < NP1; ++k) {
    const int jr = (k < TR) ? k : (2 * (k - TR) + (g & 1)), c = (k < TR) ? (g & 1) : 2;
    const int y = sy * SH - 1 + TR * rh + jr, xx = x0 + 16 * c + px;
    const bool in = ((unsigned)y < (unsigned)a.H) & (xx < a.W);
    moff[k] = in ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
    if (BWD) MB[BWD ? k : 0] = a.mbits[(in ? moff[k] : 0u) >> 3];
  }
  const int hj = TR * rh + ((px >> 1) < TR ? (px >> 1) : TR - 1), htc = (px & 1) ? TC - 1 : 0;
  unsigned hoffe = 0xffffffffu;
  unsigned HB = 0;
  if (G::CT) {
    const int y = sy * SH - 1 + hj, xx = x0 - 1 + htc;
    if (((unsigned)y < (unsigned)a.H) & ((unsigned)xx < (unsigned)a.W)) hoffe = (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0);
    if (BWD) HB = a.mbits[(hoffe != 0xffffffffu ? hoffe : 0u) >> 3];
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
        if (BWD) {
          const uint4 m4 = relu_mask_bits(make_uint4(o.x, o.y, 0, 0), HB >> (4 * (g & 1)));
          o = make_uint2(m4.x, m4.y);
        }
      }
      if (px < 2 * TR) *reinterpret_cast<uint2*>(ldt + swz(hj * TC + htc, 2 * q + (g >> 1)) + (g & 1) * 8) = o;
    }
    sweep_bases<XC>(off, 0u, TR * rh, px, g, G::CT ? 1 : 0);
    block_sweep<TR, FMT, NoHook, NC, XC>(acc, F, lds, off);
    R2_STAMP(4);
    {
      const uint4* wp = a.w2 + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
    }
#pragma unroll
    for (int k = 0; k < NP1; ++k) {
      const f32x4 tx = (k < TR) ? acc[k < TR ? k : 0][0] : acc[2 * (k < TR ? 0 : k - TR)][NC - 1];
      const f32x4 ty = (k < TR) ? acc[k < TR ? k : 0][1] : acc[2 * (k < TR ? 0 : k - TR) + 1][NC - 1];
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
        if (BWD) o = relu_mask_bits(o, MB[BWD ? k : 0]);
      }
      *reinterpret_cast<uint4*>(ldt + swz(j * TC + xx + 1, chunk8)) = o;
    }
    gate_arrive(&gate[rh], lane);
  }
  R2_STAMP(5);
  gate_wait(&gate[rh], 4u);
  if (rh == 1) gate_wait(&gate[0], 4u);
  if (BWD && prod) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the U_{k-1} pieces are in LDS (only the second filter is younger, and it is needed now)
  uint4 S[G::GREGS];
  const bool t_out = a.t != nullptr;
  if (t_out) group_stage<1, G>(S, ldt, tg, rh);
  auto t_store = [&](int grp) {
    if (grp % 3 == 0 && grp / 3 < G::GREGS) {
      const int i = grp / 3 < G::GREGS ? grp / 3 : 0;
      if (t_out && soff[i] != 0xffffffffu) {
        st16_nt(a.t + soff[i], S[i]);
        if (!BWD && a.mbits) a.mbits[soff[i] >> 3] = (unsigned char)relu_bits(S[i]);
      }
    }
  };

  // ---- phase 2: the row half's OR strip rows from T rows r .. r + 2: forward U = conv2(T) + b2 (ungated), backward dx = G + conv1^T(gt1) [+ res2] ----
  {
    f32x4 acc[OR][NC];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!BWD) { const float4 t = *reinterpret_cast<const float4*>(a.b2 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < OR; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[r][c] = b4;
    // backward: the residual operand G (its tile in LDS now holds dU) is requested before the sweep and lands under it (an L2 / MALL hit)
    unsigned ooff[NP2], osoff = 0xffffffffu;
    uint4 P1p[BWD ? NP2 : 1];
    uint2 P1s = make_uint2(0, 0);
#pragma unroll
    for (int k = 0; k < NP2; ++k) {
      const int r = (k < OR) ? k : (g & 1), c = (k < OR) ? (g & 1) : 2;
      const int y = sy * SH + OR * rh + r, xx = x0 + 16 * c + px;
      ooff[k] = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
      if (BWD) P1p[BWD ? k : 0] = *reinterpret_cast<const uint4*>(a.x + (ooff[k] != 0xffffffffu ? ooff[k] : 0u));
    }
    if (NC == 3) {
      const int y = sy * SH + OR * rh + 2, xx = x0 + 32 + px;
      osoff = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0) : 0xffffffffu;
      if (BWD) P1s = *reinterpret_cast<const uint2*>(a.x + (osoff != 0xffffffffu ? osoff : 0u));
    }
    unsigned off[8][2];
    if (rh == 0) {
      sweep_bases<TC>(off, (unsigned)G::XBYTES, 0, px, g);
      block_sweep<OR - 1, FMT, decltype(t_store), NC, TC>(*reinterpret_cast<f32x4(*)[OR - 1][NC]>(&acc[0]), F, lds, off, t_store);
      gate_wait(&gate[1], 4u);
      sweep_bases<TC>(off, (unsigned)G::XBYTES, OR - 1, px, g);
      block_sweep<1, FMT, NoHook, NC, TC>(*reinterpret_cast<f32x4(*)[1][NC]>(&acc[OR - 1]), F, lds, off);
    } else {
      sweep_bases<TC>(off, (unsigned)G::XBYTES, OR, px, g);
      block_sweep<OR, FMT, decltype(t_store), NC, TC>(acc, F, lds, off, t_store);
    }
    R2_STAMP(6);
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    float ps8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NP2; ++k) {
      const f32x4 tx = (k < OR) ? acc[k < OR ? k : 0][0] : acc[0][NC - 1];
      const f32x4 ty = (k < OR) ? acc[k < OR ? k : 0][1] : acc[1][NC - 1];
      float v[8];
      pair_up(tx, ty, g, v);
      if (ooff[k] != 0xffffffffu) {
        if (BWD) {                         // (conv1^T(gt1) + G) + res2: the order of the block kernel and of the two-launch path (bitwise the same dx)
          float m[8];
          unpack8<FMT>(P1p[BWD ? k : 0], m);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += m[j];
          if (a.res2) {
            unpack8<FMT>(*reinterpret_cast<const uint4*>(a.res2 + ooff[k]), m);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += m[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) ps8[j] += v[j];
        }
        const int r = (k < OR) ? k : (g & 1), c = (k < OR) ? (g & 1) : 2;
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<uint4*>(ldx + swz((OR * rh + r + 2) * XC + 16 * c + px + XH, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);   // OUT image in place of the
      }                                                                                                                            // input tile's centre rows
    }
    if (NC == 3 && osoff != 0xffffffffu) {
      float v[4] = {acc[NC == 3 ? 2 : 0][NC - 1][0], acc[NC == 3 ? 2 : 0][NC - 1][1], acc[NC == 3 ? 2 : 0][NC - 1][2], acc[NC == 3 ? 2 : 0][NC - 1][3]};
      if (BWD) {
        float m[4];
        unpack4<FMT>(P1s, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += m[j];
        if (a.res2) {
          unpack4<FMT>(*reinterpret_cast<const uint2*>(a.res2 + osoff), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) ps[j] += v[j];
      }
      *reinterpret_cast<uint2*>(ldx + swz((OR * rh + 2 + 2) * XC + 32 + px + XH, 2 * q + (g >> 1)) + (g & 1) * 8) = pack4<FMT>(v[0], v[1], v[2], v[3]);
    }
    if (!BWD && a.part_out) {
      // pool sums of the fp32 accumulators: one row of 64 per (strip, row half) - conv_block.hip's layout
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
        float* pp = a.part_out + ((size_t)n * np_out + 2 * si + rh) * 64 + 16 * q + 4 * g;
        *reinterpret_cast<float4*>(pp) = make_float4(ps8[0], ps8[1], ps8[2], ps8[3]);
        *reinterpret_cast<float4*>(pp + 4) = make_float4(ps8[4], ps8[5], ps8[6], ps8[7]);
      }
    }
  }
  // ---- OUT: the row half's rows sit in LDS in place of the input tile's centre rows -> whole lines, non-temporal ----
  gate_arrive(&gate[2 + rh], lane);
  R2_STAMP(7);
  gate_wait(&gate[2 + rh], 4u);
  group_stage<2, G>(S, ldx, tg, rh);
#pragma unroll
  for (int i = 0; i < G::GREGS; ++i)
    if (soff[i] != 0xffffffffu) st16_nt(a.u_out + soff[i], S[i]);
  R2_STAMP(8);
  if (BWD && prod) {
    // partial rows of sum_hw dx * U_{k-1} for the backward launch of block k - 1: this thread's dx pieces against the U pieces its own wave
    // fetched at kernel start; lanes with the same chunk hold the same 8 channels
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < G::GREGS; ++i) {
      if (soff[i] != 0xffffffffu) {
        float d[8], u[8];
        unpack8<FMT>(S[i], d);
        unpack8<FMT>(*reinterpret_cast<const uint4*>(ldu + ((rh * G::GREGS + i) * 256 + tg) * 16), u);
#pragma unroll
        for (int j = 0; j < 8; ++j) a8[j] = fmaf(d[j], u[j], a8[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = a8[j];
      t += __shfl_xor(t, 8);
      t += lane_xor16(t, g);
      t += lane_xor32(t, lane);
      a8[j] = t;
    }
    if (lane < 8) {
      float* rp = red + (rh * 4 + q) * 64 + 8 * lane;
      *reinterpret_cast<float4*>(rp) = make_float4(a8[0], a8[1], a8[2], a8[3]);
      *reinterpret_cast<float4*>(rp + 4) = make_float4(a8[4], a8[5], a8[6], a8[7]);
    }
    gate_arrive(&gate[8 + rh], lane);
    if (q == 0) {
      gate_wait(&gate[8 + rh], 4u);
      const float* rp = red + rh * 4 * 64 + lane;
      a.part_out[((size_t)n * np_out + 2 * si + rh) * 64 + lane] = (rp[0] + rp[64]) + (rp[128] + rp[192]);
    }
    R2_STAMP(9);
  }
}

// partial rows [N][np][64] -> [N][1][64] (images with more than 64 partial rows: whole-image evaluation), fixed order
__global__ void __launch_bounds__(512) rcab2_reduce_kernel(const float* __restrict__ in, int np, float* __restrict__ out) {
  __shared__ float red[8 * 64];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  int j = wave;
  for (; j + 56 < np; j += 64) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = in[((size_t)n * np + j + 8 * k) * 64 + lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; j < np; j += 8) s += in[((size_t)n * np + j) * 64 + lane];
  red[wave * 64 + lane] = s;
  __syncthreads();
  if (wave == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k * 64 + lane];
    out[(size_t)n * 64 + lane] = t;
  }
}

template <class G>
static void rcab2_dispatch(const rumpy_rcab2_args* p, const Rcab2Dev& d, hipStream_t s, bool bwd) {
  const dim3 grid(d.N * d.sy_n * d.ct_n);
  if (bwd) RUMPY_LAUNCH_PROBED(5, (rcab2_kernel<true, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
  else if (p->fmt == RUMPY_FMT_F16) RUMPY_LAUNCH_PROBED(5, (rcab2_kernel<false, RUMPY_FMT_F16, G>), grid, dim3(BTHREADS), s, d);
  else RUMPY_LAUNCH_PROBED(5, (rcab2_kernel<false, RUMPY_FMT_BF16, G>), grid, dim3(BTHREADS), s, d);
}

// Geometry of a launch on [N, H, W]: conv_block.hip::block_geometry's cost (rounds of workgroups on the CUs x time of one workgroup); no strip of
// an image has to be resident with any other, so every candidate is admissible at every image size
static void rcab2_geometry(int N, int H, int W, int* sh, int* nc, int* ct_n) {
  block_col_tiles(W, nc, ct_n);
  *sh = BSH;
  if (W <= BSW) return;
  const char* force = getenv("RUMPY_BLOCK_GEO");
  int fh = 0, fc = 0;
  if (force && (sscanf(force, "%d,%d", &fh, &fc) != 2 || !((fh == BSH && (fc == 2 || fc == 3)) || ((fh == 4 || fh == 8) && fc == 2)))) { fh = 0; fc = 0; }   // (conv_block.hip says so once)
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

// partial rows per image a launch on [N, H, W] writes (and the merge / next launch reads): 2 per workgroup of the image
extern "C" int rumpy_rcab2_partials(int32_t N, int32_t H, int32_t W) {
  int sh, nc, ct_n;
  rcab2_geometry(N, H, W, &sh, &nc, &ct_n);
  return 2 * ((H + sh - 1) / sh) * ct_n;
}

static int rcab2_launch(const rumpy_rcab2_args* p, void* stream, bool bwd, const char* what) {
  if (!p || !p->x || !p->w1 || !p->w2 || !p->u_out) { rumpy_set_error("%s: null pointer", what); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || (int64_t)p->N * p->H * p->W * 64 >= (int64_t)0xffffffffu) { rumpy_set_error("%s: bad shape", what); return RUMPY_E_ARG; }
  const bool gated = bwd || p->u_in;
  if (!bwd && (!p->b1 || !p->b2)) { rumpy_set_error("%s: forward needs b1, b2", what); return RUMPY_E_ARG; }
  if (!bwd && p->u_in && !p->x_out) { rumpy_set_error("%s: a pending branch (u_in) needs x_out", what); return RUMPY_E_ARG; }
  if (bwd && (!p->x_out || !p->t || !p->maskbits || !p->dz)) { rumpy_set_error("%s: backward needs x_out (dU), t (gt1), maskbits, dz", what); return RUMPY_E_ARG; }
  if (bwd && p->u_in && !p->part_out) { rumpy_set_error("%s: backward with u_in needs part_out", what); return RUMPY_E_ARG; }
  if (gated && p->np_in > 64 && !p->part_scratch) { rumpy_set_error("%s: %d partial rows per image need part_scratch", what, p->np_in); return RUMPY_E_ARG; }
  if (gated && (!p->part_in || p->np_in <= 0 || !p->ca_w1 || !p->ca_b1 || !p->ca_w2 || !p->ca_b2 || !p->hidden || !p->gate ||
                p->cr <= 0 || p->cr > 4 || (!bwd && !p->mean))) {
    rumpy_set_error("%s: a gate needs part_in (%d rows), the attention MLP, hidden, gate%s, 0 < Cr <= 4", what, p->np_in, bwd ? "" : ", mean"); return RUMPY_E_ARG; }
  if (bwd && p->dzq && !p->qgate) { rumpy_set_error("%s: dzq without qgate", what); return RUMPY_E_ARG; }
  if (p->fmt != RUMPY_FMT_BF16 && !(p->fmt == RUMPY_FMT_F16 && !bwd)) { rumpy_set_error("%s: fmt %d is a forward-only format", what, p->fmt); return RUMPY_E_ARG; }
  int sh, nc, ct_n;
  rcab2_geometry(p->N, p->H, p->W, &sh, &nc, &ct_n);
  Rcab2Dev d;
  d.x = (const uint16_t*)p->x; d.u_in = (const uint16_t*)p->u_in; d.part_in = p->part_in; d.np_in = p->np_in; d.part_out = p->part_out;
  d.w1 = (const uint4*)p->w1; d.b1 = p->b1; d.w2 = (const uint4*)p->w2; d.b2 = p->b2;
  d.x_out = (uint16_t*)p->x_out; d.t = (uint16_t*)p->t; d.u_out = (uint16_t*)p->u_out; d.res2 = (const uint16_t*)p->res2; d.mbits = (unsigned char*)p->maskbits;
  d.N = p->N; d.H = p->H; d.W = p->W; d.sy_n = (p->H + sh - 1) / sh; d.ct_n = ct_n;
  d.cw1 = p->ca_w1; d.cb1 = p->ca_b1; d.cw2 = p->ca_w2; d.cb2 = p->ca_b2; d.cr = p->cr; d.inv_hw = 1.0f / ((float)p->H * (float)p->W);
  d.mean = p->mean; d.hidden = p->hidden; d.gate = p->gate; d.qgate = p->qgate; d.dz = p->dz; d.dzq = p->dzq;
  hipStream_t s = (hipStream_t)stream;
  if (gated && p->np_in > 64) {            // whole-image evaluation: hundreds of rows per image - folded into one row first (fixed order)
    hipLaunchKernelGGL(rcab2_reduce_kernel, dim3(p->N), dim3(512), 0, s, p->part_in, p->np_in, p->part_scratch);
    d.part_in = p->part_scratch; d.np_in = 1;
  }
  if (p->W <= BSW) rcab2_dispatch<GeoL>(p, d, s, bwd);
  else if (sh == 8) rcab2_dispatch<BlockGeo<2, true, 8> >(p, d, s, bwd);
  else if (sh == 4) rcab2_dispatch<BlockGeo<2, true, 4> >(p, d, s, bwd);
  else if (nc == 3) rcab2_dispatch<BlockGeo<3, true> >(p, d, s, bwd);
  else rcab2_dispatch<BlockGeo<2, true> >(p, d, s, bwd);
  return rumpy_check_launch(what);
}

extern "C" int rumpy_rcab2_fwd(const rumpy_rcab2_args* p, void* stream) { return rcab2_launch(p, stream, false, "rumpy_rcab2_fwd"); }
extern "C" int rumpy_rcab2_bwd(const rumpy_rcab2_args* p, void* stream) { return rcab2_launch(p, stream, true, "rumpy_rcab2_bwd"); }

// A run of residual channel-attention blocks (the RCABs of one ResidualGroup: rumpy/SISR/models/advanced/architectures.py:60-84, :107-124) in ONE
// persistent launch, forward or backward (round 5): conv_chain.hip's loop - the strip stays in LDS from block to block, halo rows come from the
// vertical neighbours, strips are claimed per XCD - around conv_rcab.hip's block: the strips of an image exchange their 64 pool sums inside the launch
// (every workgroup is resident by construction of the chain), the gate is applied on chip.
//
// Why: an RCAB launch at 32 x 48 x 48 is ~20 us of which the matrix pipe is busy 7; 3 us are the launch boundary and 2.4 us the 10-row tile load, paid 400
// times per RCAN step.  In the chain a block reads NOTHING it has not produced itself except two halo rows per side (the forward pass's t2 rows in the
// backward direction: prefetched by LDS-DMA under the previous block).  What a block still stores is what the weight gradients read: t1, t2, the block
// output (forward); d_t2, gt1, dx (backward).
//
// Forward block:   t1 = relu(conv1(x) + b1) ; t2 = conv2(t1) + b2 ; gate = CA(mean_hw(t2)) [* qgate] ; x' = x + gate * t2            (conv_rcab.hip)
// Backward block:  ds = sum_hw(dy * t2) -> MLP backward -> d_t2 = dy * gate + dp / HW ; gt1 = [t1 > 0] . conv2^T(d_t2) ; dx = dy + conv1^T(gt1) [+ res2]
// Arithmetic, summation orders and roundings are those of rumpy_rcab_fwd / rumpy_rcab_bwd: the chain is bitwise the per-block launches
// (tests/test_chain_gpu.py).  W <= 48 (a strip spans the image), N * ceil(H/6) <= CUs, Cr <= 4 (the backward direction keeps t2's rows in LDS beside
// the two images: 152 KB + the attention MLP's operands), bf16.
#include "chain_common.hpp"
#include "rcab_common.hpp"
#include "rumpy_experimental.h"
#include <cstdlib>

struct RcBlk {
  const uint16_t* x; const uint4* w1; const float* b1; const uint4* w2; const float* b2;
  uint16_t* t; uint16_t* t2; const uint16_t* t2_in; const uint16_t* res2; uint16_t* out; unsigned char* mbits;
  const float* cw1; const float* cb1; const float* cw2; const float* cb2;
  float* mean; float* hidden; float* gate; const float* qgate; float* dz; float* dzq;
};
static_assert(sizeof(RcBlk) == sizeof(rumpy_rcab_chain_block), "rumpy_rcab_chain_block is the device-side block record");
struct RcChainDev {
  const RcBlk* blk; int nblk, N, H, W, sy_n, cr; float inv_hw;
  unsigned* work; unsigned long long* xchg; unsigned xchg_bytes; unsigned* status; int nxcd, fake_xcc, force_sc1;
};
constexpr int RCC_R = 4;               // hidden units of the attention MLP this kernel holds in LDS

// all-gather of one fp32 per (strip, channel) among the strips of image n (rcab_common.hpp::strip_allsum with the chain's tags); returns (threads < 64:
// channel tid) the sum over strips in strip order.  Called by all 512 threads.
__device__ __forceinline__ void rcc_post(const RcChainDev& a, float mine, int n, int si, int tid, unsigned tag) {
  const rc_rsrc rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.xchg, 0, a.xchg_bytes, 0x00020000);
  if (tid < 64) __builtin_amdgcn_raw_buffer_store_b64((rc_u32x2){__float_as_uint(mine), tag}, rr, (unsigned)(((n * a.sy_n + si) * 64 + (tid & 63)) * 8), 0, RC_SC1);
}
__device__ __forceinline__ float rcc_gather(const RcChainDev& a, int n, int tid, unsigned tag, float* sx) {
  const rc_rsrc rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.xchg, 0, a.xchg_bytes, 0x00020000);
  const int c = tid & 63, w = tid >> 6;
  float total = 0.f;
  for (int s0 = 0; s0 < a.sy_n; s0 += 8) {
    const int s = s0 + w;
    float val = 0.f;
    if (s < a.sy_n) {
      const unsigned byte = (unsigned)(((n * a.sy_n + s) * 64 + c) * 8);
      rc_u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rr, byte, 0, RC_SC1);
      unsigned spins = 0;
      while (!__all(r.y == tag)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > RC_SPIN) { if (c == 0) atomicExch(a.status, 0x600u + (tag & 255u)); break; }
        r = __builtin_amdgcn_raw_buffer_load_b64(rr, byte, 0, RC_SC1);
      }
      val = __uint_as_float(r.x);
    }
    __syncthreads();
    sx[w * 64 + c] = val;
    __syncthreads();
    if (tid < 64) {
#pragma unroll
      for (int k = 0; k < 8; ++k) total += sx[k * 64 + c];
    }
  }
  return total;
}

typedef __attribute__((address_space(3))) unsigned char* rcc_lds_u8;
__device__ __forceinline__ void rcc_dma16(const void* gsrc, unsigned lds_dst) {      // wgrad_dma.hip::dma16
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// -DRCC_STAMPS (measurement builds only): phase time stamps (s_memrealtime, 100 MHz) of every wave in the chain's MIDDLE block
#ifdef RCC_STAMPS
__device__ unsigned long long* g_rcc_stamps;
#define RCC_STAMP(k) do { if (b == a.nblk / 2 && (threadIdx.x & 63) == 0 && g_rcc_stamps) g_rcc_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int rumpy_debug_rcc_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rcc_stamps), &buf, sizeof(buf)); }
#else
#define RCC_STAMP(k) do { } while (0)
#endif

template <bool BWD>
__global__ void __launch_bounds__(BTHREADS, 2) rcab_chain_kernel(RcChainDev a) {
  constexpr int UBYTES = BWD ? STRIP_REGS * BTHREADS * 16 : 16;      // bwd: the forward pass's t2, the strip's own rows, piece p = tid + 512 i at 16 p
  __shared__ __attribute__((aligned(16))) unsigned char lds[BXBYTES + BTBYTES];
  __shared__ __attribute__((aligned(16))) unsigned char ldu[UBYTES];
  __shared__ float sx[8 * 64];
  __shared__ float spool[2 * 64];
  __shared__ __attribute__((aligned(16))) float sgate[64];
  __shared__ __attribute__((aligned(16))) float sdp[64];
  __shared__ float sw1[RCC_R * 64];        // [r][c] = conv_du.0.weight
  __shared__ float sw2t[RCC_R * 64];       // [r][c] = conv_du.2.weight[c][r]
  __shared__ float svec[4 * 64];           // [0] conv_du.0.bias (cr) | [1] conv_du.2.bias | [2] q gate | [3] bwd: forward gate ; hidden at [0][32..]
  __shared__ unsigned gate[8];             // per row half: T rows written [0,1], OUT rows written [2,3], halo rows in LDS [4,5], stores acknowledged [6,7]
  __shared__ int claim[4];
  unsigned char* const ldx = lds;
  unsigned char* const ldt = lds + BXBYTES;
  const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = wave & 3, rh = wave >> 2;
  const ChainPlace place = chain_claim(a.work, a.N, a.sy_n, a.nxcd, a.fake_xcc, claim);
  const unsigned epoch = place.epoch;
  const int strip = place.strip;
  const int n = strip / a.sy_n, sy = strip - n * a.sy_n;
  const bool has_nb = (rh == 0) ? (sy > 0) : (sy + 1 < a.sy_n);
  const int nb_strip = (rh == 0) ? strip - 1 : strip + 1;
  unsigned* const flags = chain_flags(a.work, gridDim.x);
  const bool local = has_nb && !a.force_sc1 && chain_same_xcd(a.work, epoch, nb_strip, place.xcc, a.status);
  const RcBlk b0 = a.blk[0];
  // backward: the forward pass's t2, the strip's own rows, by LDS-DMA (no registers): piece p = tid + 512 i lands at byte 16 p of ldu
  auto t2_dma = [&](const uint16_t* src, int tid) {
    const unsigned ubase = (unsigned)(size_t)(rcc_lds_u8)ldu;
#pragma unroll
    for (int i = 0; i < STRIP_REGS; ++i) {
      const unsigned so = strip_piece_off(i, tid, n, sy, a.H, a.W);
      rcc_dma16((const void*)(src + (so != 0xffffffffu ? so : 0u)), __builtin_amdgcn_readfirstlane(ubase + (BTHREADS * i + 64 * wave) * 16));
    }
  };

  // ---- block 0: input rows 6sy-2 .. 6sy+7, columns -1 .. 48 -> LDS ----
  if (BWD) t2_dma(b0.t2_in, tid);
  {
    uint4 R[BREGS];
    const int y0 = sy * BSH - 2;
#pragma unroll
    for (int i = 0; i < BREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      const int lr = pix / BCOLS, lc = pix - lr * BCOLS;
      const int y = y0 + lr, x = lc - 1;
      const bool ok = (p < BPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
      const int e = ok ? ((n * a.H + y) * a.W + x) * 64 + part * 8 : 0;
      R[i] = keep_if(*reinterpret_cast<const uint4*>(b0.x + (unsigned)e), ok);
    }
    if (tid < 8) gate[tid] = 0u;
    if (tid < BTROWS * 2 * 8) {
      const int row = tid >> 4, side = (tid >> 3) & 1, chunk = tid & 7;
      *reinterpret_cast<uint4*>(ldt + swz(row * BCOLS + side * (BCOLS - 1), chunk)) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BREGS; ++i) {
      const int p = tid + BTHREADS * i;
      if (p < BPIECES) *reinterpret_cast<uint4*>(ldx + swz(p >> 3, p & 7)) = R[i];
    }
  }
  bf16x8 F[18];
  if (!BWD) {
    const uint4* wp = b0.w1 + (size_t)q * 18 * 64 + lane0;
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
  }
  __syncthreads();

  for (int b = 0; b < a.nblk; ++b) {
    const RcBlk blk = a.blk[b];
    const unsigned tag = (epoch << 8) + (unsigned)b;
    const unsigned done = 4u * (unsigned)b;              // gate counts at the end of block b - 1
    RCC_STAMP(0);
    int lane = lane0, tidb = tid;
    asm volatile("" : "+v"(lane), "+v"(tidb));           // (per-lane geometry is recomputed per block from opaque copies: hoisted out of the loop it spills, conv_chain.hip)
    const int px = lane & 15, g = lane >> 4, tg = 64 * q + lane;
    const int c0 = 16 * q + 4 * g;
    const int gpair = 4 * (g & ~1);
    const int chunk8 = 2 * q + (gpair >> 3);
    // the attention MLP's operands of this block: requested here, written to LDS behind the first conv (the previous block's MLP is long done: every
    // wave has passed the barriers of its tail)
    float mw1 = 0.f, mw2 = 0.f, mv;
    {
      const int mtot = a.cr * 64;
      if (tidb < RCC_R * 64) { mw1 = blk.cw1[tidb < mtot ? tidb : 0]; mw2 = blk.cw2[tidb < mtot ? tidb : 0]; }
      const int which = tidb >> 6, c = tidb & 63;
      const int cr_c = c < a.cr ? c : 0;
      const float* src = blk.cb2; int idx = c;
      if (which == 0) { src = blk.cb1; idx = cr_c; }
      else if (which == 2 && blk.qgate) { src = blk.qgate; idx = n * 64 + c; }
      else if (which == 3 && BWD) { src = blk.gate; idx = n * 64 + c; }
      else if (which == 4 && BWD) { src = blk.hidden; idx = n * a.cr + cr_c; }
      mv = src[idx];
      if (which == 2 && !blk.qgate) mv = 1.f;
    }
    // lane geometry of the epilogues and stores (offsets in a [N,H,W,64] tensor): filled where the direction needs it - the backward direction's prologue
    // (product, reduce, exchange, transform) runs without these 20 registers
    unsigned moff[6], soff[GROUP_REGS];
    unsigned MB[BWD ? 6 : 1];
    auto geometry = [&]() {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int jr = (k < 4) ? k : (2 * (k - 4) + (g & 1)), c = (k < 4) ? (g & 1) : 2;
        const int y = sy * BSH - 1 + 4 * rh + jr, xx = 16 * c + px;
        const bool in = ((unsigned)y < (unsigned)a.H) & (xx < a.W);
        moff[k] = in ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
      }
#pragma unroll
      for (int i = 0; i < GROUP_REGS; ++i) soff[i] = group_piece_off(i, tg, rh, n, sy, a.H, a.W);
      if (BWD) {
#pragma unroll
        for (int k = 0; k < 6; ++k) MB[BWD ? k : 0] = blk.mbits[(moff[k] != 0xffffffffu ? moff[k] : 0u) >> 3];
      }
    };
    if (!BWD) geometry();
    // hand-off of block b - 1 -> this block's halo rows (conv_chain.hip): acknowledge the OUT stores, publish, poll the neighbour, fetch its two rows
    auto hand_off = [&]() {
      // halo pieces of this row half: 2 rows x 48 columns x 8 chunks = 768 = 3 per thread; rows 6sy-2, 6sy-1 (half 0) or 6sy+6, 6sy+7 (half 1)
      unsigned hoff[3], hlds[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int p = tg + 256 * i, pix = p >> 3, r = pix / BSW, col = pix - r * BSW;
        const int y = (rh == 0) ? sy * BSH - 2 + r : sy * BSH + BSH + r;
        hoff[i] = (has_nb && (unsigned)y < (unsigned)a.H && col < a.W) ? (unsigned)(((n * a.H + y) * a.W + col) * 64 + (p & 7) * 8) : 0xffffffffu;
        hlds[i] = swz(((rh == 0) ? r : BSH + 2 + r) * BCOLS + col + 1, p & 7);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      gate_arrive(&gate[6 + rh], lane);
      if (q == 0) {
        gate_wait(&gate[6 + rh], done);
        if (lane == 0) { if (local) ch_store_flag_sc0(flags + (2 * strip + rh) * CH_FLAG_STRIDE, (epoch << 8) + (unsigned)b); else __hip_atomic_store(flags + (2 * strip + rh) * CH_FLAG_STRIDE, (epoch << 8) + (unsigned)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
      if (has_nb) {
        const unsigned want = (unsigned)b;
        unsigned spins = 0;
        unsigned long long t0 = 0;
        for (;;) {
          const unsigned f = __hip_atomic_load(flags + (2 * nb_strip + (1 - rh)) * CH_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((f >> 8) == epoch && (f & 0xffu) >= want) break;
          const int tr = ch_poll_round(spins, t0, a.status);
          if (tr == 1 && lane == 0) atomicExch(a.status, 0x500u + (unsigned)b);
          if (tr) break;
        }
      }
      uint4 Hr[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) Hr[i] = ch_load16_sc1(blk.x + (hoff[i] != 0xffffffffu ? hoff[i] : 0u));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (hoff[i] != 0xffffffffu) *reinterpret_cast<uint4*>(ldx + hlds[i]) = Hr[i];
      gate_arrive(&gate[4 + rh], lane);
      gate_wait(&gate[4 + rh], done);
    };
    auto mlp_to_lds = [&]() {
      const int mtot = a.cr * 64;
      if (tidb < mtot) {
        sw1[tidb] = mw1;                                       // [r][c] as stored
        sw2t[(tidb % a.cr) * 64 + tidb / a.cr] = mw2;           // [c][r] -> [r][c]
      }
      const int which = tidb >> 6, c = tidb & 63;
      if (which == 0) { if (c < RCC_R) svec[c] = mv; }
      else if (which < 4) svec[which * 64 + c] = mv;
      else if (which == 4 && c < RCC_R) svec[32 + c] = mv;
    };

    f32x4 acc[4][3];
    unsigned off[8][2];
    if (!BWD) {
      // ================= forward: first conv as in conv_chain.hip (halo-free half, hand-off, halo-dependent half) =================
      {
        const float4 t = *reinterpret_cast<const float4*>(blk.b1 + c0);
        const f32x4 b4 = (f32x4){t.x, t.y, t.z, t.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[r][c] = b4;
      }
      if (b == 0) {
        sweep_bases(off, 0u, 4 * rh, px, g);
        block_sweep<4>(acc, F, lds, off);
      } else {
        gate_wait(&gate[2], done);
        gate_wait(&gate[3], done);
        sweep_bases(off, 0u, (rh == 0) ? 2 : 4, px, g);
        block_sweep<2>(*reinterpret_cast<f32x4(*)[2][3]>(&acc[(rh == 0) ? 2 : 0]), F, lds, off);
        hand_off();
        RCC_STAMP(2);
        sweep_bases(off, 0u, (rh == 0) ? 0 : 6, px, g);
        block_sweep<2>(*reinterpret_cast<f32x4(*)[2][3]>(&acc[(rh == 0) ? 0 : 2]), F, lds, off);
      }
      mlp_to_lds();
    } else {
      // ================= backward: ds = sum(dy * t2) over the strip -> all strips -> MLP backward -> d_t2 in place; hand-off under the exchange =================
      if (b > 0) {
        gate_wait(&gate[2], done);
        gate_wait(&gate[3], done);                       // dy's own rows (= dx of block b - 1) are in LDS; nobody reads the old T image
      }
      // (the forward pass's t2 rows of this block are in LDS: requested in the prologue (block 0) or under the previous block, and every load this wave
      // has waited for since is younger)
      float part8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < STRIP_REGS; ++i) {
        const int p = tidb + BTHREADS * i;
        const int pix = p >> 3, r = pix / BSW, col = pix - r * BSW;
        if (p < STRIP_PIECES && sy * BSH + r < a.H && col < a.W) {
          float d[8], t[8];
          unpack8(*reinterpret_cast<const uint4*>(ldx + swz((r + 2) * BCOLS + col + 1, tidb & 7)), d);
          unpack8(*reinterpret_cast<const uint4*>(ldu + p * 16), t);
#pragma unroll
          for (int j = 0; j < 8; ++j) part8[j] = fmaf(d[j], t[j], part8[j]);
        }
      }
      float* red = reinterpret_cast<float*>(ldt);          // (the T image is dead here)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(tidb >> 3) * 64 + (tidb & 7) * 8 + j] = part8[j];
      __syncthreads();
      if (tidb < 256) {
        const int c = tidb & 63, part = tidb >> 6;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[(part * 16 + k) * 64 + c];
        red[64 * 64 + part * 64 + c] = s;
      }
      __syncthreads();
      float mine = 0.f;
      if (tidb < 64) mine = (red[64 * 64 + tidb] + red[64 * 64 + 64 + tidb]) + (red[64 * 64 + 128 + tidb] + red[64 * 64 + 192 + tidb]);
      mlp_to_lds();
      __syncthreads();                                     // red is dead: the border columns of the T image are rewritten below
      if (tidb < BTROWS * 2 * 8) {
        const int row = tidb >> 4, side = (tidb >> 3) & 1, chunk = tidb & 7;
        *reinterpret_cast<uint4*>(ldt + swz(row * BCOLS + side * (BCOLS - 1), chunk)) = make_uint4(0, 0, 0, 0);
      }
      // fetch the neighbours' rows, then - the t2 rows of this block have been consumed by every wave: two barriers ago - request the NEXT block's by
      // LDS-DMA: they have the rest of this block to land
      RCC_STAMP(1);
      rcc_post(a, mine, n, sy, tidb, tag);               // this strip's sums are on their way while ...
      if (b > 0) hand_off();             // ... the neighbours' rows are fetched
      RCC_STAMP(2);
      if (b + 1 < a.nblk) t2_dma(a.blk[b + 1].t2_in, tidb);
      {                                                  // the first sweep's filter: lands under the exchange, the MLP and the tile's transform
        const uint4* wp = blk.w1 + (size_t)q * 18 * 64 + lane;
#pragma unroll
        for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
      }
      const float ds = rcc_gather(a, n, tidb, tag, sx);
      RCC_STAMP(3);
      if (tidb < 64) {
        const int c = tidb;
        const float s = svec[3 * 64 + c];
        const float gq = svec[2 * 64 + c];
        const float dz = (ds * gq) * s * (1.f - s);
        float dp = 0.f;
        {
          float dhs[4], w1[4], hid[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = (i < a.cr) ? i : 0;
            dhs[i] = sw2t[r * 64 + c] * dz; w1[i] = sw1[r * 64 + c]; hid[i] = svec[32 + r];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) dhs[i] = wave_sum(dhs[i]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < a.cr) {
              const float dh = (hid[i] > 0.f) ? dhs[i] : 0.f;
              dp = fmaf(w1[i], dh, dp);
            }
          }
        }
        sgate[c] = s * gq;
        sdp[c] = dp * a.inv_hw;
        if (sy == 0) {
          blk.dz[n * 64 + c] = dz;
          if (blk.dzq) blk.dzq[n * 64 + c] = (ds * s) * gq * (1.f - gq);
        }
      }
      __syncthreads();
      {
        const int y0 = sy * BSH - 2;
        const float4 ga = *reinterpret_cast<const float4*>(sgate + (tidb & 7) * 8), gb = *reinterpret_cast<const float4*>(sgate + (tidb & 7) * 8 + 4);
        const float4 pa = *reinterpret_cast<const float4*>(sdp + (tidb & 7) * 8), pb = *reinterpret_cast<const float4*>(sdp + (tidb & 7) * 8 + 4);
#pragma unroll
        for (int i = 0; i < BREGS; ++i) {
          const int p = tidb + BTHREADS * i;
          const int pix = p >> 3, part = p & 7;
          const int lr = pix / BCOLS, lc = pix - lr * BCOLS;
          const int y = y0 + lr, x = lc - 1;
          const bool ok = (p < BPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
          if (ok) {
            uint4* cell = reinterpret_cast<uint4*>(ldx + swz(pix, part));
            float d[8];
            unpack8(*cell, d);
            const uint2 lo = pack4_bf16(fmaf(d[0], ga.x, pa.x), fmaf(d[1], ga.y, pa.y), fmaf(d[2], ga.z, pa.z), fmaf(d[3], ga.w, pa.w));
            const uint2 hi = pack4_bf16(fmaf(d[4], gb.x, pb.x), fmaf(d[5], gb.y, pb.y), fmaf(d[6], gb.z, pb.z), fmaf(d[7], gb.w, pb.w));
            const uint4 o = make_uint4(lo.x, lo.y, hi.x, hi.y);
            *cell = o;
            if (lr >= 2 && lr < 2 + BSH) st16_nt(blk.t2 + (unsigned)(((n * a.H + y) * a.W + x) * 64 + part * 8), o);
          }
        }
      }
      __syncthreads();
      geometry();
      RCC_STAMP(4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      sweep_bases(off, 0u, 4 * rh, px, g);
      block_sweep<4>(acc, F, lds, off);
    }
    RCC_STAMP(5);
    // second filter: L2 hits that land under the epilogue
    {
      const uint4* wp = blk.w2 + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
    }
    // ---- epilogue 1: pairs k < 4: (row k, col tile 0 | 1); k = 4: rows 0 | 1 of col tile 2; k = 5: rows 2 | 3 ----
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const f32x4 tx = (k < 4) ? acc[k < 4 ? k : 0][0] : acc[2 * (k < 4 ? 0 : k - 4)][2];
      const f32x4 ty = (k < 4) ? acc[k < 4 ? k : 0][1] : acc[2 * (k < 4 ? 0 : k - 4) + 1][2];
      float v[8];
      pair_up(tx, ty, g, v);
      if (!BWD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = relu_f32(v[j]);
      }
      const int jr = (k < 4) ? k : (2 * (k - 4) + (g & 1)), c = (k < 4) ? (g & 1) : 2;
      uint4 o = make_uint4(0, 0, 0, 0);
      if (moff[k] != 0xffffffffu) {
        const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
        o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (BWD) o = relu_mask_bits(o, MB[BWD ? k : 0]);
      }
      *reinterpret_cast<uint4*>(ldt + swz((4 * rh + jr) * BCOLS + 16 * c + px + 1, chunk8)) = o;
    }
    RCC_STAMP(6);
    gate_arrive(&gate[rh], lane);
    gate_wait(&gate[rh], done + 4u);
    if (rh == 1) gate_wait(&gate[0], done + 4u);
    uint4 S[GROUP_REGS];
    const bool t_out = blk.t != nullptr;
    if (t_out) group_stage<1>(S, ldt, tg, rh);
    // the row half's own T rows (+ mask bytes) -> HBM from the LDS image, whole lines, non-temporal - all of them in front of the second sweep (conv_block.hip
    // spreads them under it, one piece per third MFMA group: here the 20 staged registers beside the sweep's spilled, and a spill reload between MFMAs is
    // a vmcnt wait that also waits for these very stores)
    if (t_out) {
#pragma unroll
      for (int i = 0; i < GROUP_REGS; ++i) {
        if (soff[i] != 0xffffffffu) {
          st16_nt(blk.t + soff[i], S[i]);
          if (!BWD && blk.mbits) blk.mbits[soff[i] >> 3] = (unsigned char)relu_bits(S[i]);
        }
      }
    }
    // ---- phase 2: rows 3rh .. 3rh+2 of the strip from T rows r .. r+2 ----
    f32x4 acc2[3][3];
    {
      f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (!BWD) { const float4 t = *reinterpret_cast<const float4*>(blk.b2 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc2[r][c] = b4;
    }
    // backward: the residual operand dy (its tile in LDS now holds d_t2) comes back from the tensor the previous block stored - plain loads, tracked by the
    // compiler's wait counters (an asm load would hand over registers the hardware fills LATER): this CU has never read these lines before (no stale L1
    // copy), they are in this XCD's L2 (sc0 stores) or written through (sc1); requested before the sweep, they land under it
    unsigned ooff[4], osoff = 0xffffffffu;
    uint4 P1p[BWD ? 4 : 1];
    uint2 P1s = make_uint2(0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = (k < 3) ? k : (g & 1), c = (k < 3) ? (g & 1) : 2;
      const int y = sy * BSH + 3 * rh + r, xx = 16 * c + px;
      ooff[k] = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
      if (BWD) P1p[BWD ? k : 0] = *reinterpret_cast<const uint4*>(blk.x + (ooff[k] != 0xffffffffu ? ooff[k] : 0u));
    }
    {
      const int y = sy * BSH + 3 * rh + 2, xx = 32 + px;
      osoff = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0) : 0xffffffffu;
      if (BWD) P1s = *reinterpret_cast<const uint2*>(blk.x + (osoff != 0xffffffffu ? osoff : 0u));
    }
    if (rh == 0) {
      sweep_bases(off, (unsigned)BXBYTES, 0, px, g);
      block_sweep<2>(*reinterpret_cast<f32x4(*)[2][3]>(&acc2[0]), F, lds, off);
      gate_wait(&gate[1], done + 4u);
      sweep_bases(off, (unsigned)BXBYTES, 2, px, g);
      block_sweep<1>(*reinterpret_cast<f32x4(*)[1][3]>(&acc2[2]), F, lds, off);
    } else {
      sweep_bases(off, (unsigned)BXBYTES, 3, px, g);
      block_sweep<3>(acc2, F, lds, off);
    }
    RCC_STAMP(7);
    // pairs k < 3: X = (row k, col 0), Y = (row k, col 1); k = 3: X = (0, 2), Y = (1, 2); single: (2, 2)
    float V[4][8], vs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 tx = (k < 3) ? acc2[k < 3 ? k : 0][0] : acc2[0][2];
      const f32x4 ty = (k < 3) ? acc2[k < 3 ? k : 0][1] : acc2[1][2];
      pair_up(tx, ty, g, V[k]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) vs[j] = acc2[2][2][j];

    if (BWD) {
      // dx = dy + conv1^T(gt1) [+ res2] -> LDS, in place of the d_t2 tile's centre rows (dead in phase 2)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (ooff[k] != 0xffffffffu) {
          float m[8];
          unpack8(P1p[BWD ? k : 0], m);
          if (blk.res2) {
            float e[8];
            unpack8(*reinterpret_cast<const uint4*>(blk.res2 + ooff[k]), e);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] += e[j];
          }
          const uint2 lo = pack4_bf16(V[k][0] + m[0], V[k][1] + m[1], V[k][2] + m[2], V[k][3] + m[3]);
          const uint2 hi = pack4_bf16(V[k][4] + m[4], V[k][5] + m[5], V[k][6] + m[6], V[k][7] + m[7]);
          const int r = (k < 3) ? k : (g & 1), c = (k < 3) ? (g & 1) : 2;
          *reinterpret_cast<uint4*>(ldx + swz((3 * rh + r + 2) * BCOLS + 16 * c + px + 1, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      }
      if (osoff != 0xffffffffu) {
        float m[4];
        unpack4_bf16(P1s, m);
        if (blk.res2) {
          float e[4];
          unpack4_bf16(*reinterpret_cast<const uint2*>(blk.res2 + osoff), e);
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] += e[j];
        }
        *reinterpret_cast<uint2*>(ldx + swz((3 * rh + 2 + 2) * BCOLS + 32 + px + 1, 2 * q + (g >> 1)) + (g & 1) * 8) =
            pack4_bf16(vs[0] + m[0], vs[1] + m[1], vs[2] + m[2], vs[3] + m[3]);
      }
    } else {
      // t2 = conv2(t1) + b2: channel sums of the strip for the attention pool, t2 itself to HBM (training)
      float ps8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
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
      if (blk.t2) {                        // t2 goes to HBM through the T image's rows 1 .. 6 (whole lines, below)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (ooff[k] != 0xffffffffu) {
            const int r = (k < 3) ? k : (g & 1), c = (k < 3) ? (g & 1) : 2;
            const uint2 lo = pack4_bf16(V[k][0], V[k][1], V[k][2], V[k][3]), hi = pack4_bf16(V[k][4], V[k][5], V[k][6], V[k][7]);
            *reinterpret_cast<uint4*>(ldt + swz((3 * rh + r + 1) * BCOLS + 16 * c + px + 1, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
        }
        if (osoff != 0xffffffffu)
          *reinterpret_cast<uint2*>(ldt + swz((3 * rh + 2 + 1) * BCOLS + 32 + px + 1, 2 * q + (g >> 1)) + (g & 1) * 8) = pack4_bf16(vs[0], vs[1], vs[2], vs[3]);
      }
      const float mine = (tidb < 64) ? spool[tidb] + spool[64 + tidb] : 0.f;
      rcc_post(a, mine, n, sy, tidb, tag);
      const float tot = rcc_gather(a, n, tidb, tag, sx);      // (its barriers also complete the t2 image)
      RCC_STAMP(8);
      if (blk.t2) {                        // t2 leaves right here (it does not wait for the gate): whole lines, non-temporal
        uint4 S5[STRIP_REGS];
        strip_stage<1>(S5, ldt, tidb);
#pragma unroll
        for (int i = 0; i < STRIP_REGS; ++i) {
          const unsigned so = strip_piece_off(i, tidb, n, sy, a.H, a.W);
          if (so != 0xffffffffu) st16_nt(blk.t2 + so, S5[i]);
        }
      }
      if (tidb < 64) {
        const int c = tidb;
        const float mean = tot * a.inv_hw;
        float z = svec[64 + c];
        {
          float hs[4], w2[4], b1[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = (i < a.cr) ? i : 0;
            hs[i] = sw1[r * 64 + c] * mean; w2[i] = sw2t[r * 64 + c]; b1[i] = svec[r];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) hs[i] = wave_sum(hs[i]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i < a.cr) {
              const float h = fmaxf(hs[i] + b1[i], 0.f);
              z = fmaf(w2[i], h, z);
              if (sy == 0 && c == 0) blk.hidden[n * a.cr + i] = h;
            }
          }
        }
        const float gt = 1.f / (1.f + expf(-z));
        sgate[c] = gt * svec[2 * 64 + c];
        if (sy == 0) { blk.mean[n * 64 + c] = mean; blk.gate[n * 64 + c] = gt; }
      }
      RCC_STAMP(9);
      __syncthreads();
      // out = x + gate * t2, the residual operand from the input tile in LDS; the result replaces it there
      const float4 ga = *reinterpret_cast<const float4*>(sgate + 16 * q + gpair), gb = *reinterpret_cast<const float4*>(sgate + 16 * q + gpair + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (ooff[k] != 0xffffffffu) {
          const int r = (k < 3) ? k : (g & 1), c = (k < 3) ? (g & 1) : 2;
          const int srow = 3 * rh + r, xx = 16 * c + px;
          float m[8];
          unpack8(*reinterpret_cast<const uint4*>(ldx + swz((srow + 2) * BCOLS + xx + 1, chunk8)), m);
          const uint2 lo = pack4_bf16(fmaf(V[k][0], ga.x, m[0]), fmaf(V[k][1], ga.y, m[1]), fmaf(V[k][2], ga.z, m[2]), fmaf(V[k][3], ga.w, m[3]));
          const uint2 hi = pack4_bf16(fmaf(V[k][4], gb.x, m[4]), fmaf(V[k][5], gb.y, m[5]), fmaf(V[k][6], gb.z, m[6]), fmaf(V[k][7], gb.w, m[7]));
          *reinterpret_cast<uint4*>(ldx + swz((srow + 2) * BCOLS + xx + 1, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      }
      if (osoff != 0xffffffffu) {
        const int srow = 3 * rh + 2, xx = 32 + px;
        float m[4];
        unpack4_bf16(*reinterpret_cast<const uint2*>(ldx + swz((srow + 2) * BCOLS + xx + 1, 2 * q + (g >> 1)) + (g & 1) * 8), m);
        const float4 gs = *reinterpret_cast<const float4*>(sgate + c0);
        *reinterpret_cast<uint2*>(ldx + swz((srow + 2) * BCOLS + xx + 1, 2 * q + (g >> 1)) + (g & 1) * 8) =
            pack4_bf16(fmaf(vs[0], gs.x, m[0]), fmaf(vs[1], gs.y, m[1]), fmaf(vs[2], gs.z, m[2]), fmaf(vs[3], gs.w, m[3]));
      }
    }
    // the next block's first filter: requested here - not behind the second sweep as in conv_chain.hip: the 72 registers beside the accumulator pairs of the
    // tail spilled (240 / 476 bytes of scratch per lane) - and lands under the stores and the next block's waits
    if (!BWD && b + 1 < a.nblk) {          // (backward: requested behind the pool exchange of the block that uses it - 72 registers less through its prologue)
      const uint4* wp = a.blk[b + 1].w1 + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
    }
    // ---- OUT (forward: x + gate * t2; backward: dx): per row half, whole lines, handed to the neighbours (conv_chain.hip) ----
    RCC_STAMP(10);
    gate_arrive(&gate[2 + rh], lane);
    gate_wait(&gate[2 + rh], done + 4u);
    group_stage<2>(S, ldx, tg, rh);
#pragma unroll
    for (int i = 0; i < GROUP_REGS; ++i)
      if (soff[i] != 0xffffffffu) { if (local) ch_store16_sc0(blk.out + soff[i], S[i]); else ch_store16_sc1(blk.out + soff[i], S[i]); }
    RCC_STAMP(11);
  }
}

extern "C" int64_t rumpy_rcab_chain_work_bytes(int32_t N, int32_t H) { return chain_work_bytes((int64_t)N * ((H + BSH - 1) / BSH)); }

extern "C" int rumpy_rcab_chain(const rumpy_rcab_chain_args* p, void* stream) {
  if (!p || !p->blocks || !p->work || !p->status || !p->xchg || p->nblocks <= 0 || p->nblocks > 255) { rumpy_set_error("rumpy_rcab_chain: bad argument"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->W > BSW) { rumpy_set_error("rumpy_rcab_chain: needs 0 < W <= 48 (got %d)", p->W); return RUMPY_E_ARG; }
  if (p->cr <= 0 || p->cr > RCC_R) { rumpy_set_error("rumpy_rcab_chain: needs 0 < Cr <= %d (got %d)", RCC_R, p->cr); return RUMPY_E_ARG; }
  const int sy_n = (p->H + BSH - 1) / BSH;
  if (p->N * sy_n > rumpy_device_cus()) { rumpy_set_error("rumpy_rcab_chain: %d strips do not fit %d CUs (all must be co-resident)", p->N * sy_n, rumpy_device_cus()); return RUMPY_E_ARG; }
  if (p->work_bytes < rumpy_rcab_chain_work_bytes(p->N, p->H)) { rumpy_set_error("rumpy_rcab_chain: work buffer too small"); return RUMPY_E_ARG; }
  const int64_t need = (int64_t)p->N * sy_n * 64 * 8;
  if (p->xchg_bytes < need) { rumpy_set_error("rumpy_rcab_chain: exchange buffer too small (%lld < %lld)", (long long)p->xchg_bytes, (long long)need); return RUMPY_E_ARG; }
  if (p->fake_xcc < 0 || (p->fake_xcc > 0 && !p->force_sc1)) { rumpy_set_error("rumpy_rcab_chain: fake_xcc (a test hook) goes with force_sc1"); return RUMPY_E_ARG; }
  RcChainDev d;
  d.blk = reinterpret_cast<const RcBlk*>(p->blocks); d.nblk = p->nblocks; d.N = p->N; d.H = p->H; d.W = p->W; d.sy_n = sy_n; d.cr = p->cr;
  d.inv_hw = 1.0f / ((float)p->H * (float)p->W);
  d.work = (unsigned*)p->work; d.xchg = (unsigned long long*)p->xchg; d.xchg_bytes = (unsigned)need; d.status = (unsigned*)p->status;
  d.nxcd = rumpy_device_xcds(); d.fake_xcc = p->fake_xcc; d.force_sc1 = p->force_sc1;
  if (d.fake_xcc > 0) d.nxcd = d.fake_xcc < CH_MAX_XCD ? d.fake_xcc : CH_MAX_XCD;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(p->N * sy_n);
  if (p->backward) RUMPY_LAUNCH_PROBED(5, (rcab_chain_kernel<true>), grid, dim3(BTHREADS), s, d);
  else RUMPY_LAUNCH_PROBED(5, (rcab_chain_kernel<false>), grid, dim3(BTHREADS), s, d);
  return rumpy_check_launch("rumpy_rcab_chain");
}

// 3x3 same-padding convolution, Cin = 64 (x chunks), on bf16 MFMA: strip kernel (the dominant kernel of the hot path).
//
// Work unit = a strip of 6 output rows x 48 output columns of one image (288 pixels: exactly one strip per CU
// for the headline shape 32 x 48 x 48).  One 512-thread workgroup per CU; wave (q, rh) owns output channels
// 16q..16q+15 of output rows 3rh..3rh+2 and keeps its filter slice (16 co x 576 k = 18 MFMA A-fragments = 72 VGPRs)
// in registers for the whole kernel; waves never exchange data, so each runs at its own pace and stores as soon as
// its 162 MFMAs are done (one barrier per strip, only to recycle the LDS buffers).
//   D[co 16][px 16] += A[co][k = 32 input channels of one tap] * B[k][px]          (v_mfma_f32_16x16x32_bf16)
// The 8 x 50-pixel input halo of a strip is staged HBM -> registers -> LDS as two 32-channel halves (64-byte pixel
// halves at a 96-byte stride: conflict-free ds_read_b128 B-fragments).  One B-fragment (input row, column tile, tap
// column, channel half) feeds up to 3 output rows: 90 LDS reads per 162 MFMAs per wave.
// Epilogue (per lane 4 consecutive channels of a pixel, 8-byte vectors): bias enters as the accumulator's initial
// value, then ReLU, scale, ReLU mask, per-strip channel sums, two residual adds, PixelShuffle scatter, bf16 store;
// mask / residual vectors are prefetched under the MFMAs.  Persistent over strips with the next strip's halo
// prefetched into the other LDS buffer.
#include "common.hpp"

constexpr int SH = 6, SW = 48;
constexpr int SROWS = SH + 2, SCOLS = SW + 2, SPIX = SROWS * SCOLS;   // 8 x 50 = 400 halo pixels
constexpr int HSTRIDE = 96;                                           // bytes per 32-channel pixel half in LDS
constexpr int HHALF = SPIX * HSTRIDE;                                 // 38400: one K half of a stage
constexpr int SSTAGE = 2 * HHALF;                                     // 76800
constexpr int SPIECES = SPIX * 8;                                     // 16-byte pieces per stage
constexpr int STHREADS = 512;
constexpr int SREGS = (SPIECES + STHREADS - 1) / STHREADS;            // 7 per thread

struct StripDev {
  const uint16_t* x; const uint4* w; const float* bias; uint16_t* out;
  const uint16_t* mask; const uint16_t* res1; const uint16_t* res2; float* pool;
  int N, H, W, cout_tiles, in_mode, out_mode, relu; float scale; int sx_n, sy_n, skew;
};

struct StripCoord { int n, sy, sx; };
__device__ __forceinline__ StripCoord decode_strip(int s, int sx_n, int sy_n) {
  StripCoord c;
  c.sx = s % sx_n;
  const int r = s / sx_n;
  c.sy = r % sy_n;
  c.n = r / sy_n;
  return c;
}

// Staging: piece p (16 B) = channels 8*part..8*part+7 of halo pixel pix = p/8 (row pix/50, column pix%50); it lands in K
// half part/4 of the stage.  The geometry is recomputed at each use (a few VALU ops) instead of living in registers
// across the MFMA loop.  Branch-free: every lane loads from a clamped in-bounds address and the result is zeroed
// afterwards, so the loads of a stage issue back to back (no exec-mask regions; 32-bit element offsets).
// mode 0: src is [N,H,W,64*chunks], input chunk `ch` = channels 64ch..64ch+63.
// mode 1: src is [N,2H,2W,64] (a pixel-shuffled gradient); chunk ch = sub-pixel (ch/2, ch%2): PixelShuffle^T gather.
template <int CHUNKS>
__device__ __forceinline__ void strip_issue(uint4 (&R)[SREGS], const uint16_t* __restrict__ src, int mode, int ch, StripCoord c,
                                            int H, int W, int tid) {
  const int y0 = c.sy * SH - 1, x0 = c.sx * SW - 1;
  int base, rstep, cstep;
  if (CHUNKS == 1 || mode == 0) {
    base = ((c.n * H + y0) * W + x0) * (64 * CHUNKS) + ch * 64;
    rstep = W * 64 * CHUNKS; cstep = 64 * CHUNKS;
  } else {
    base = ((c.n * 2 * H + 2 * y0 + (ch >> 1)) * (2 * W) + 2 * x0 + (ch & 1)) * 64;
    rstep = 4 * W * 64; cstep = 128;
  }
#pragma unroll
  for (int i = 0; i < SREGS; ++i) {
    const int p = tid + STHREADS * i;
    const int pix = p >> 3, part = p & 7;
    const int lr = pix / SCOLS, lc = pix - lr * SCOLS;
    const int y = y0 + lr, x = x0 + lc;
    const bool ok = (p < SPIECES) & ((unsigned)y < (unsigned)H) & ((unsigned)x < (unsigned)W);
    const int e = ok ? base + lr * rstep + lc * cstep + part * 8 : 0;
    uint4 v = *reinterpret_cast<const uint4*>(src + (unsigned)e);
    if (!ok) v = make_uint4(0, 0, 0, 0);
    R[i] = v;
  }
}
__device__ __forceinline__ void strip_write(const uint4 (&R)[SREGS], unsigned char* lds, int tid) {
#pragma unroll
  for (int i = 0; i < SREGS; ++i) {
    const int p = tid + STHREADS * i;
    const int pix = p >> 3, part = p & 7;
    if (p < SPIECES) *reinterpret_cast<uint4*>(lds + (part >> 2) * HHALF + pix * HSTRIDE + (part & 3) * 16) = R[i];
  }
}

// STAMP = true is a diagnostic build: lane 0 of every wave writes s_memrealtime (100 MHz) stamps at the phase
// boundaries of its FIRST strip into a.pool (reinterpreted as u64 [workgroup][wave][8]); never used by the product.
// DBG (diagnostic builds only): 1 = MFMAs without the LDS fragment reads, 2 = LDS fragment reads without the MFMAs
// FMT: element format of x, w, out and the residual operands (RUMPY_FMT_F16: the evaluation plans)
template <int CHUNKS, bool STAMP, int DBG = 0, int FMT = RUMPY_FMT_BF16>
__global__ void __launch_bounds__(STHREADS, 2) conv3x3_strip_kernel(StripDev a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * SSTAGE];
  unsigned long long stamps[8];
  int nst = 0;
#define STAMP_HERE() do { if (STAMP && nst < 8) { stamps[nst++] = __builtin_amdgcn_s_memrealtime(); } } while (0)
  STAMP_HERE();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = wave >> 2;
  const int ct = blockIdx.y;
  const int nstrips = a.N * a.sy_n * a.sx_n;

  int strip = xcd_strip(blockIdx.x, gridDim.x);        // contiguous runs of strips per XCD: neighbours share halo rows in L2
  if (strip >= nstrips) return;
  // the input loads go out first (longest latency); the filter comes from L2 behind them
  uint4 R[SREGS];
  strip_issue<CHUNKS>(R, a.x, a.in_mode, 0, decode_strip(strip, a.sx_n, a.sy_n), a.H, a.W, tid);

  // stationary filter slice: packed [ct][chunk][q][s = tap*2 + ci half][lane] x 16 B
  //   (CHUNKS > 1: the 18 fragments are re-fetched from L2 for every input chunk)
  bf16x8 F[18];
  const uint4* wbase = a.w + ((size_t)(ct * CHUNKS * 4 + q) * 18) * 64 + lane;
#pragma unroll
  for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wbase[t * 64]);
  const int c0 = 16 * q + 4 * g;   // this lane's 4 consecutive channels inside the 64-channel tile
  f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (a.bias) {                     // the bias is the accumulator's initial value
    const float4 b4 = *reinterpret_cast<const float4*>(a.bias + ct * 64 + c0);
    bias4 = (f32x4){b4.x, b4.y, b4.z, b4.w};
  }
  STAMP_HERE();                         // 1: all prologue loads issued
  strip_write(R, lds, tid);
  __syncthreads();
  STAMP_HERE();                         // 2: first stage in LDS

  // epilogue operands prefetched under the MFMAs: P0 = mask (or res1), P1 = res1 (or res2)
  const uint16_t* p0 = a.mask ? a.mask : a.res1;
  const uint16_t* p1 = a.mask ? a.res1 : a.res2;
  const uint16_t* p2 = (a.mask && a.res1) ? a.res2 : nullptr;   // only when all three are given (not prefetched)
  int buf = 0;
  // Phase skew (persistent launches): the two waves that share a SIMD would run MFMAs together and then their epilogues together - the
  // matrix pipe idles during every epilogue.  Waves selected by a.skew DEFER the epilogue of strip i to the start of iteration i+1
  // (the accumulators stay in registers over the barrier), so that on each SIMD one wave's epilogue runs under the other's MFMAs.
  const bool skewed = !STAMP && a.skew != 0 &&
                      (((a.skew == 1) ? (wave >> 2) : (a.skew == 2) ? wave : (wave >> 1)) & 1);
  bool pending = false;                 // a deferred epilogue is outstanding
  unsigned poff[4], soff;
  uint4 P0p[4], P1p[4];
  uint2 P0s, P1s;
  f32x4 acc[3][3];
  float* pool_ptr = nullptr;

  auto epilogue = [&]() {
    // ---- epilogue ----
    float ps[4] = {0.f, 0.f, 0.f, 0.f};              // single tile: channels 4g .. 4g+3 of this wave's 16
    float ps8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // paired tiles: channels 4(g&~1) .. +7
    // per-value math on this lane's OWN 4 channels (bias is already in the accumulator)
    auto own = [&](f32x4 t) -> f32x4 {
      if (a.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = relu_f32(t[j]);
      }
      if (a.scale != 1.0f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] *= a.scale;
      }
      return t;
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 tx = own((k < 3) ? acc[k][0] : acc[0][2]);
      const f32x4 ty = own((k < 3) ? acc[k][1] : acc[1][2]);
      float v[8], m[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // v_permlane16_swap_b32 (gfx950): even-g lanes get (own X, X of lane g+1), odd-g lanes (Y of lane g-1, own Y) - see block_common.hpp
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(tx[j]), __float_as_uint(ty[j]), false, false);
        v[j] = __uint_as_float(r[0]);                       // channels gpair .. gpair+3
        v[4 + j] = __uint_as_float(r[1]);                   // channels gpair+4 .. gpair+7
      }
      const bool in = poff[k] != 0xffffffffu;
      if (a.mask) {
        unpack4<FMT>(make_uint2(P0p[k].x, P0p[k].y), *reinterpret_cast<float(*)[4]>(&m[0]));
        unpack4<FMT>(make_uint2(P0p[k].z, P0p[k].w), *reinterpret_cast<float(*)[4]>(&m[4]));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (m[j] > 0.f) ? v[j] : 0.f;
      }
      if (a.pool && in) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ps8[j] += v[j];
      }
      if (!a.mask && p0) {
        unpack4<FMT>(make_uint2(P0p[k].x, P0p[k].y), *reinterpret_cast<float(*)[4]>(&m[0]));
        unpack4<FMT>(make_uint2(P0p[k].z, P0p[k].w), *reinterpret_cast<float(*)[4]>(&m[4]));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += m[j];
      }
      if (p1) {
        unpack4<FMT>(make_uint2(P1p[k].x, P1p[k].y), *reinterpret_cast<float(*)[4]>(&m[0]));
        unpack4<FMT>(make_uint2(P1p[k].z, P1p[k].w), *reinterpret_cast<float(*)[4]>(&m[4]));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += m[j];
      }
      if (in) {
        if (p2) {
          const uint4 t = *reinterpret_cast<const uint4*>(p2 + poff[k]);
          unpack4<FMT>(make_uint2(t.x, t.y), *reinterpret_cast<float(*)[4]>(&m[0]));
          unpack4<FMT>(make_uint2(t.z, t.w), *reinterpret_cast<float(*)[4]>(&m[4]));
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += m[j];
        }
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<uint4*>(a.out + poff[k]) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    {   // the single unpaired tile (row 2, column tile 2): 8-byte path
      const f32x4 t = own(acc[2][2]);
      float v[4] = {t[0], t[1], t[2], t[3]};
      float m[4];
      if (a.mask) {
        unpack4<FMT>(P0s, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (m[j] > 0.f) ? v[j] : 0.f;
      }
      if (a.pool && soff != 0xffffffffu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ps[j] += v[j];
      }
      if (!a.mask && p0) {
        unpack4<FMT>(P0s, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += m[j];
      }
      if (p1) {
        unpack4<FMT>(P1s, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += m[j];
      }
      if (soff != 0xffffffffu) {
        if (p2) {
          unpack4<FMT>(*reinterpret_cast<const uint2*>(p2 + soff), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        *reinterpret_cast<uint2*>(a.out + soff) = pack4<FMT>(v[0], v[1], v[2], v[3]);
      }
    }
    if (a.pool && !STAMP) {
      // per-(strip, row half) channel sums: pool[n][(2*sy + rh) * sx_n + sx][channel]; reduce over the 16 pixel lanes,
      // fold the odd-g lanes (other pixels, same 8 channels) into the even ones, add the single tile's 4+4 channels
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = row16_sum(ps8[j]);                    // (common.hpp: the butterflies on the VALU, same bits as the shuffles)
        t += lane_xor16(t, g);
        ps8[j] = t;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = row16_sum(ps[j]);
        const float up = lane_xor16(t, g);              // the same sums of lane group g ^ 1
        ps8[j] += (g & 1) ? up : t;                      // channels 4(g&~1) + j     belong to the even group
        ps8[4 + j] += (g & 1) ? t : up;                  // channels 4(g&~1) + 4 + j belong to the odd group
      }
      if (px == 0 && !(g & 1)) {
        float* pp = pool_ptr;
        *reinterpret_cast<float4*>(pp) = make_float4(ps8[0], ps8[1], ps8[2], ps8[3]);
        *reinterpret_cast<float4*>(pp + 4) = make_float4(ps8[4], ps8[5], ps8[6], ps8[7]);
      }
    }
  };

  for (; strip < nstrips; strip += gridDim.x) {
    const StripCoord sc = decode_strip(strip, a.sx_n, a.sy_n);
    // Epilogue geometry.  The MFMA leaves 4 channels x 1 pixel per lane (8 B); stores that narrow are issue-bound, so tiles
    // are processed in PAIRS (X, Y): lanes with even g trade their Y values for the neighbour lane's (g+1) X values
    // (one __shfl_xor(.,16) per dword) and end up with 8 consecutive channels of X's pixel, odd-g lanes with 8 channels
    // of Y's pixel -> every global access of the epilogue is a 16-byte vector.  4 pairs + 1 single tile per wave:
    //   pair k<3: X = (row k, col tile 0), Y = (row k, col tile 1); pair 3: X = (0, 2), Y = (1, 2); single: (2, 2).
    if (skewed && pending) epilogue();       // strip i-1, under the other wave's MFMAs of strip i
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = bias4;
    if (a.pool) pool_ptr = a.pool + ((size_t)(sc.n * a.sy_n * 2 + 2 * sc.sy + rh) * a.sx_n + sc.sx) * (64 * a.cout_tiles) + ct * 64 + 16 * q + 4 * g;

    unsigned char* stage = lds;
    bool has_next = false;
#pragma unroll 1
    for (int ch = 0; ch < CHUNKS; ++ch) {
      // prefetch the next stage: next input chunk of this strip, or chunk 0 of the next strip
      const int nstrip = (ch + 1 < CHUNKS) ? strip : strip + (int)gridDim.x;
      const int nch = (ch + 1 < CHUNKS) ? ch + 1 : 0;
      has_next = nstrip < nstrips;
      if (has_next) strip_issue<CHUNKS>(R, a.x, a.in_mode, nch, decode_strip(nstrip, a.sx_n, a.sy_n), a.H, a.W, tid);
      if (ch == CHUNKS - 1) {
        // element offsets (0xffffffff = outside the image) + prefetch of the mask / residual vectors
        auto pix_off = [&](int r, int c) -> unsigned {
          const int y = sc.sy * SH + 3 * rh + r, xx = sc.sx * SW + 16 * c + px;
          if (y >= a.H || xx >= a.W) return 0xffffffffu;
          if (a.out_mode == 0) return (unsigned)(((sc.n * a.H + y) * a.W + xx) * (64 * a.cout_tiles) + ct * 64 + 16 * q);
          return (unsigned)(((sc.n * 2 * a.H + 2 * y + (ct >> 1)) * (2 * a.W) + 2 * xx + (ct & 1)) * 64 + 16 * q);
        };
        const int gpair = 4 * (g & ~1);                    // first of this lane's 8 channels inside the wave's 16
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned ox = (k < 3) ? pix_off(k, 0) : pix_off(0, 2);
          const unsigned oy = (k < 3) ? pix_off(k, 1) : pix_off(1, 2);
          const unsigned o = (g & 1) ? oy : ox;
          poff[k] = (o != 0xffffffffu) ? o + gpair : 0xffffffffu;
          const unsigned oc = (o != 0xffffffffu) ? o + gpair : 0u;   // clamped: the prefetch loads are unconditional
          P0p[k] = make_uint4(0, 0, 0, 0); P1p[k] = make_uint4(0, 0, 0, 0);
          if (p0) P0p[k] = *reinterpret_cast<const uint4*>(p0 + oc);
          if (p1) P1p[k] = *reinterpret_cast<const uint4*>(p1 + oc);
        }
        {
          const unsigned o = pix_off(2, 2);
          soff = (o != 0xffffffffu) ? o + 4 * g : 0xffffffffu;
          const unsigned oc = (o != 0xffffffffu) ? o + 4 * g : 0u;
          P0s = make_uint2(0, 0); P1s = make_uint2(0, 0);
          if (p0) P0s = *reinterpret_cast<const uint2*>(p0 + oc);
          if (p1) P1s = *reinterpret_cast<const uint2*>(p1 + oc);
        }
      }
      stage = lds + buf * SSTAGE;
      {
        // 18 groups (channel half, tap column, column tile) of 5 B-fragment reads + 9 MFMAs, software pipelined: the
        // reads of group i+1 are issued before the MFMAs of group i (two fragment sets), so LDS and the matrix pipe
        // overlap inside one wave instead of alternating (measured: 2.1 us of reads + 2.3 us of MFMAs otherwise add up)
        const unsigned char* wbase = stage + (3 * rh * SCOLS + px) * HSTRIDE + g * 16;
        bf16x8 I[2][5];
        auto load_group = [&](int grp, bf16x8 (&dst)[5]) {
          const int half = grp / 9, kx = (grp % 9) / 3, c = grp % 3;
          const unsigned char* cur = wbase + half * HHALF + (16 * c + kx) * HSTRIDE;
#pragma unroll
          for (int r = 0; r < 5; ++r) {
            if (DBG == 1) dst[r] = F[r];
            else dst[r] = *reinterpret_cast<const bf16x8*>(cur + r * SCOLS * HSTRIDE);
          }
        };
        load_group(0, I[0]);
#pragma unroll
        for (int grp = 0; grp < 18; ++grp) {
          if (grp + 1 < 18) load_group(grp + 1, I[(grp + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);   // keep the next group's reads AHEAD of this group's MFMAs (hipcc sinks them otherwise)
          const int half = grp / 9, kx = (grp % 9) / 3, c = grp % 3;
          if (DBG == 2) {
#pragma unroll
            for (int r = 0; r < 5; ++r) asm volatile("" :: "v"(I[grp & 1][r]));
          } else {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
              for (int r = 0; r < 3; ++r)
                acc[r][c] = mfma16<FMT>(F[(ky * 3 + kx) * 2 + half], I[grp & 1][r + ky], acc[r][c]);
          }
        }
      }
      if (CHUNKS > 1 && has_next) {     // filter fragments of the next stage (L2 hits) land under the hand-over below
        __builtin_amdgcn_sched_barrier(0);   // not above the MFMAs that still read the current fragments (a second set would spill)
#pragma unroll
        for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wbase[(size_t)nch * (4 * 18 * 64) + t * 64]);
      }
      if (ch + 1 < CHUNKS) {            // not the last chunk: hand the LDS buffers over and continue accumulating
        if (has_next) strip_write(R, lds + (buf ^ 1) * SSTAGE, tid);
        __syncthreads();
        buf ^= 1;
      }
    }
    STAMP_HERE();                       // 3: MFMAs issued
    // The next strip's registers go to the other LDS buffer HERE, before the epilogue: a wait for them issued behind the
    // epilogue's stores would have to drain those stores first (the persistent upsampler launches run 16 strips per
    // workgroup; the single-strip 64 -> 64 launches have no next strip).
    if (has_next) strip_write(R, lds + (buf ^ 1) * SSTAGE, tid);
    // (issuing the loads of the strip after the next one here, one iteration earlier, measured 8 % SLOWER on the 96x96 launches)
    STAMP_HERE();                       // 4: next strip staged
    if (skewed) pending = true; else epilogue();
    __syncthreads();
    STAMP_HERE();                       // 5: epilogue done
    buf ^= 1;
    if (STAMP) break;
  }
  if (skewed && pending) epilogue();
  if (STAMP && lane == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.pool) + ((size_t)blockIdx.x * 8 + wave) * 8;
    for (int i = 0; i < 8; ++i) dbg[i] = (i < nst) ? stamps[i] : 0ull;
  }
#undef STAMP_HERE
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

extern "C" int rumpy_conv_pool_tiles(int32_t H, int32_t W, int32_t cin_chunks) {
  if (cin_chunks == 1) return 2 * cdiv(H, SH) * cdiv(W, SW);
  return cdiv(H, TH) * cdiv(W, TW);
}

int rumpy_conv3x3_strip_launch(const rumpy_conv_args* p, hipStream_t s) {
  StripDev d;
  d.x = (const uint16_t*)p->x; d.w = (const uint4*)p->w; d.bias = p->bias; d.out = (uint16_t*)p->out;
  d.mask = (const uint16_t*)p->mask; d.res1 = (const uint16_t*)p->res1; d.res2 = (const uint16_t*)p->res2; d.pool = p->pool;
  d.N = p->N; d.H = p->H; d.W = p->W; d.cout_tiles = p->cout_tiles; d.in_mode = p->in_mode; d.out_mode = p->out_mode;
  d.relu = p->relu; d.scale = p->scale; d.sx_n = cdiv(p->W, SW); d.sy_n = cdiv(p->H, SH);
  const int nstrips = d.N * d.sx_n * d.sy_n;
  static const int skew_env = getenv("RUMPY_STRIP_SKEW") ? atoi(getenv("RUMPY_STRIP_SKEW")) : 1;   // 0 = off (A/B runs)
  int gx = p->grid_x;
  if (gx <= 0) {
    const int slots = rumpy_device_cus() / p->cout_tiles > 0 ? rumpy_device_cus() / p->cout_tiles : 1;
    const int rounds = cdiv(nstrips, slots);
    gx = cdiv(nstrips, rounds);
  }
  if (gx > nstrips) gx = nstrips;
  d.skew = (gx < nstrips) ? skew_env : 0;         // only where a workgroup runs several strips

  if (p->relu >= 0x5754 && p->relu <= 0x5756) {   // diagnostic stamp builds (rumpy_debug_conv_stamps)
    const int dbg = p->relu - 0x5754;
    d.relu = 0;
    if (dbg == 0) hipLaunchKernelGGL((conv3x3_strip_kernel<1, true, 0>), dim3(gx, 1), dim3(STHREADS), 0, s, d);
    else if (dbg == 1) hipLaunchKernelGGL((conv3x3_strip_kernel<1, true, 1>), dim3(gx, 1), dim3(STHREADS), 0, s, d);
    else hipLaunchKernelGGL((conv3x3_strip_kernel<1, true, 2>), dim3(gx, 1), dim3(STHREADS), 0, s, d);
  } else if (p->fmt == RUMPY_FMT_F16) {   // evaluation plans: forward launches only (rumpy_conv3x3 checks cin_chunks == 1, no mask)
    hipLaunchKernelGGL((conv3x3_strip_kernel<1, false, 0, RUMPY_FMT_F16>), dim3(gx, p->cout_tiles), dim3(STHREADS), 0, s, d);
  } else if (p->cin_chunks == 1) {
    hipLaunchKernelGGL((conv3x3_strip_kernel<1, false>), dim3(gx, p->cout_tiles), dim3(STHREADS), 0, s, d);
  } else {
    hipLaunchKernelGGL((conv3x3_strip_kernel<4, false>), dim3(gx, p->cout_tiles), dim3(STHREADS), 0, s, d);
  }
  return 0;
}

// diagnostic: one launch of the stamped build; a->pool must hold grid_x * 8 * 8 u64 (a->grid_x > 0, no pool partials)
extern "C" int rumpy_debug_conv_stamps(const rumpy_conv_args* p, void* stream) {
  if (!p || !p->pool || p->cin_chunks != 1 || p->cout_tiles != 1 || p->grid_x <= 0) { rumpy_set_error("rumpy_debug_conv_stamps: bad argument"); return RUMPY_E_ARG; }
  rumpy_conv_args q = *p;
  q.relu = 0x5754 + ((p->relu >= 1 && p->relu <= 2) ? p->relu : 0);   // relu field selects the ablation (0, 1, 2)
  rumpy_conv3x3_strip_launch(&q, (hipStream_t)stream);
  return rumpy_check_launch("rumpy_debug_conv_stamps");
}

// Layer-resident chain of 3x3 convolutions 64 -> 64 (the EDSR body, forward or data-gradient direction) in ONE launch.
//
// A 64->64 layer on the headline batch is 288 pixels per CU: run as one kernel per layer it pays, every layer, the kernel
// boundary (dispatch + write-back of the 9.4 MB output), the first HBM loads and the store drain around 2.5 us of MFMAs.
// Here every workgroup keeps ITS strip (6 rows x <=48 columns of one image; geometry, wave roles, MFMA sweep and the
// paired 16-byte epilogue are those of conv_strip.hip) for the whole chain: layer l's epilogue writes its bf16 output to
// HBM (the backward pass and later residual adds need it) AND straight into the other LDS buffer in B-fragment layout,
// where it is layer l+1's input.  Only the two halo rows come from the vertical neighbours (strips above / below in the
// same image), through a small exchange buffer written in the style of RCCL's LL protocol: every 8-byte record is
// {2 bf16 channels, 32-bit epoch}, stored with one write-through (sc1) 64-bit store, so the data IS the signal:
//   producer: the two edge rows of the strip are the first tiles of the epilogue; besides the normal output store they
//             go out as records tagged epoch = launch base + layer + 1 (no fence, no flag, no drain);
//   consumer: at the start of the next layer each row group loads its 1536 neighbour records (sc1 loads), runs the 8 of
//             every 9 MFMAs that do not need the halo row, then checks the tags - a stale tag only means the neighbour is
//             late: reload (bounded spin with s_sleep).  Slots are double-buffered by layer parity; a neighbour can be at
//             most one layer ahead, because it needs this strip's edge row to go further.
// MI355X_MICROARCH.md visibility table, row 1 (sc1 store -> sc1 load through L2) is all this relies on; 64-bit stores are
// single-copy atomic.  The epoch base lives in the buffer header and is bumped by a one-thread kernel after every launch,
// so a hipGraph replay never matches the previous replay's records.
// Needs all strips co-resident: grid = N * ceil(H/6) <= number of CUs, one 512-thread workgroup per CU (153.6 KB LDS),
// W <= 48, nothing else on the GPU.  A timed-out spin sets *status; the kernel always terminates.
#include "common.hpp"
#include "rumpy_experimental.h"
#include <type_traits>

constexpr int CSH = 6, CSW = 48;
constexpr int CROWS = CSH + 2, CCOLS = CSW + 2, CPIX = CROWS * CCOLS;   // 8 x 50 halo pixels
constexpr int CSTRIDE = 96;                                             // bytes per 32-channel pixel half in LDS
constexpr int CHALF = CPIX * CSTRIDE;                                   // 38400
constexpr int CSTAGE = 2 * CHALF;                                       // 76800
constexpr int CTHREADS = 512;
constexpr int CPIECES = CPIX * 8, CREGS = (CPIECES + CTHREADS - 1) / CTHREADS;
constexpr unsigned CSPIN_LIMIT = 1u << 20;
constexpr int CREC = CSW * 32;          // 8-byte records of one halo row: 48 pixels x 32 channel pairs
constexpr int CXHDR = 8;                // header records (word 0 = epoch base)

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct ChainDev {
  const uint16_t* x; const rumpy_chain_layer* layers; int nlayers, N, H, W, sy_n;
  unsigned long long* xchg; unsigned* status; unsigned long long* stamps;
};

// Pointers read from the device-side layer table are generic to the compiler (flat_* instructions, which also count on
// lgkmcnt and serialise with the LDS pipeline): every access through them is cast to the global address space.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4 g_cu4;
typedef __attribute__((address_space(1))) const u32x2 g_cu2;
typedef __attribute__((address_space(1))) const f32x4v g_cf4;
typedef __attribute__((address_space(1))) u32x4 g_u4;
typedef __attribute__((address_space(1))) u32x2 g_u2;
// The layer table is read with vector loads (the compiler cannot prove the kernel's own stores leave it alone), so its
// fields look divergent: 64-bit per-lane pointer arithmetic and dozens of VGPRs.  They are uniform: move them to SGPRs.
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long v = (unsigned long long)(uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
// uniform base (SGPR pair) + 32-bit element offset: the saddr + voffset form, no 64-bit per-lane address arithmetic
__device__ __forceinline__ uint4 gload16(const uint16_t* b, unsigned e) {
  const u32x4 t = *(g_cu4*)((uintptr_t)b + (unsigned)(e * 2u));
  return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint2 gload8(const uint16_t* b, unsigned e) {
  const u32x2 t = *(g_cu2*)((uintptr_t)b + (unsigned)(e * 2u));
  return make_uint2(t.x, t.y);
}
__device__ __forceinline__ void gstore16(uint16_t* b, unsigned e, uint4 v) {
  *(g_u4*)((uintptr_t)b + (unsigned)(e * 2u)) = (u32x4){v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void gstore8(uint16_t* b, unsigned e, uint2 v) {
  *(g_u2*)((uintptr_t)b + (unsigned)(e * 2u)) = (u32x2){v.x, v.y};
}
__device__ __forceinline__ void sc1_store64(unsigned long long* b, unsigned rec, unsigned long long v) {
  __hip_atomic_store((gu64*)((uintptr_t)b + (unsigned)(rec * 8u)), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long sc1_load64(const unsigned long long* b, unsigned rec) {
  return __hip_atomic_load((gu64*)((uintptr_t)b + (unsigned)(rec * 8u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MFMA sweep over groups [G0, G1) of the 18 (channel half, tap column, column tile) groups, software pipelined (the reads
// of group i+1 are issued before the MFMAs of group i).  EDGE selects the part of each group:
//   0 = all 5 fragment reads / 9 MFMAs (layer 0: the strip was loaded with its halos)
//   1 = the 4 reads / 8 MFMAs that do not touch the halo row     2 = the 1 read / 1 MFMA that does
// RH = 0: the halo row is window row 0 (used by out row 0, ky 0); RH = 1: window row 4 (out row 2, ky 2).
// Accumulator rows are counted from the edge row: acc[0] is the row that needs the halo (out row 0 / out row 2).
template <int RH, int EDGE>
__device__ __forceinline__ bool frag_wanted(int r) {
  const bool is_halo = (RH == 0) ? (r == 0) : (r == 4);
  return (EDGE == 0) || (EDGE == 1 && !is_halo) || (EDGE == 2 && is_halo);
}
template <int RH, int EDGE, int G0, int G1>
__device__ __forceinline__ void chain_sweep(f32x4 (&acc)[3][3], const bf16x8 (&F)[18], const unsigned char* wbase) {
  bf16x8 I[2][5];
  auto load_group = [&](int grp, bf16x8 (&dst)[5]) {
    const int half = grp / 9, kx = (grp % 9) / 3, c = grp % 3;
    const unsigned char* cur = wbase + half * CHALF + (16 * c + kx) * CSTRIDE;
#pragma unroll
    for (int r = 0; r < 5; ++r)
      if (frag_wanted<RH, EDGE>(r)) dst[r] = *reinterpret_cast<const bf16x8*>(cur + r * CCOLS * CSTRIDE);
  };
  load_group(G0, I[G0 & 1]);
#pragma unroll
  for (int grp = G0; grp < G1; ++grp) {
    if (grp + 1 < G1) load_group(grp + 1, I[(grp + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);   // keep the next group's reads ahead of this group's MFMAs
    const int half = grp / 9, kx = (grp % 9) / 3, c = grp % 3;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const bool uses_halo = (RH == 0) ? (r + ky == 0) : (r + ky == 4);
        if ((EDGE == 0) || (EDGE == 1 && !uses_halo) || (EDGE == 2 && uses_halo))
          acc[RH ? 2 - r : r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[(ky * 3 + kx) * 2 + half], I[grp & 1][r + ky], acc[RH ? 2 - r : r][c], 0, 0, 0);
      }
  }
}

__global__ void __launch_bounds__(CTHREADS, 2) conv_chain_kernel(ChainDev a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * CSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = wave >> 2;
  const int strip = blockIdx.x;
  const int n = strip / a.sy_n, sy = strip - n * a.sy_n;
  const int lane256 = tid & 255;                 // index inside the 4-wave row group

  // ---- prologue: layer 0's input strip (with its halos) straight from HBM, second buffer cleared ----
  {
    uint4 R[CREGS];
    const int y0 = sy * CSH - 1, x0 = -1;
    const int base = ((n * a.H + y0) * a.W + x0) * 64;
#pragma unroll
    for (int i = 0; i < CREGS; ++i) {
      const int p = tid + CTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      const int lr = pix / CCOLS, lc = pix - lr * CCOLS;
      const int y = y0 + lr, x = x0 + lc;
      const bool ok = (p < CPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
      const int e = ok ? base + (lr * a.W + lc) * 64 + part * 8 : 0;
      uint4 v = *reinterpret_cast<const uint4*>(a.x + (unsigned)e);
      if (!ok) v = make_uint4(0, 0, 0, 0);
      R[i] = v;
    }
    for (int i = tid; i < CSTAGE / 16; i += CTHREADS) *reinterpret_cast<uint4*>(lds + CSTAGE + i * 16) = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < CREGS; ++i) {
      const int p = tid + CTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      if (p < CPIECES) *reinterpret_cast<uint4*>(lds + (part >> 2) * CHALF + pix * CSTRIDE + (part & 3) * 16) = R[i];
    }
  }
  bf16x8 F[18];
  {
    const uint16_t* wq = (const uint16_t*)uniform_ptr(a.layers[0].w) + q * 18 * 512;
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(gload16(wq, (unsigned)(lane * 8 + t * 512)));
  }
  __syncthreads();

  // ---- loop-invariant epilogue geometry (every tensor of the chain is [N,H,W,64]) ----
  // Accumulator rows are counted from the strip's EDGE row: acc[0] is strip row 0 for the upper row group (rh = 0) and
  // strip row 5 for the lower one (rh = 1) - the only row whose sums need the neighbour's halo row; acc[1], acc[2] are
  // final after the big sweep.  Tiles are paired for 16-byte accesses as in conv_strip.hip (even-g lanes finish tile X's
  // pixel, odd-g lanes tile Y's, 8 consecutive channels 16q + 4(g&~1) ..):
  //   pair 0: X = (0, col 0), Y = (0, col 1)  [edge row]     pair 1, 2: X = (k, 0), Y = (k, 1)
  //   pair 3: X = (1, col 2), Y = (2, col 2)                  single: (0, col 2)  [edge row]
  // halo hand-off: this row group needs ONE neighbour row of the previous layer's output and owes that neighbour its own edge row
  const bool has_nb = (rh == 0) ? (sy > 0) : (sy + 1 < a.sy_n);
  const unsigned epoch0 = *reinterpret_cast<const unsigned*>(a.xchg);
  unsigned long long* const xrec = a.xchg + CXHDR;
  // slot (strip, edge 0 = top row / 1 = bottom row, layer parity): CREC records
  const unsigned my_slot = (unsigned)((strip * 2 + rh) * 2) * CREC;
  const unsigned nb_slot = (unsigned)((((rh == 0) ? strip - 1 : strip + 1) * 2 + (1 - rh)) * 2) * CREC;

  int buf = 0;
#define CSTAMP(k) do { if (a.stamps && lane == 0 && l < 8) a.stamps[(((size_t)strip * 8 + wave) * 8 + l) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  for (int l = 0; l < a.nlayers; ++l) {
    CSTAMP(0);
    // an opaque zero per layer: address arithmetic that uses it is recomputed here instead of being hoisted out of the layer
    // loop into dozens of long-lived registers (which spill; a scratch reload waits behind the previous layer's stores)
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    const int lane256z = lane256 + zero, pxz = px + zero;
    rumpy_chain_layer ly = a.layers[l];
    ly.bias = uniform_ptr(ly.bias); ly.out = uniform_ptr(ly.out);
    ly.mask = uniform_ptr(ly.mask); ly.res1 = uniform_ptr(ly.res1); ly.res2 = uniform_ptr(ly.res2);
    ly.relu = __builtin_amdgcn_readfirstlane(ly.relu);
    ly.scale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ly.scale)));
    const uint16_t* p0 = (const uint16_t*)(ly.mask ? ly.mask : ly.res1);
    const uint16_t* p1 = (const uint16_t*)(ly.mask ? ly.res1 : ly.res2);
    const uint16_t* p2 = (const uint16_t*)((ly.mask && ly.res1) ? ly.res2 : nullptr);
    unsigned char* cur = lds + buf * CSTAGE;
    unsigned char* nxt = lds + (buf ^ 1) * CSTAGE;
    const unsigned char* wbase = cur + (3 * rh * CCOLS + px) * CSTRIDE + g * 16;
    const bool more = l + 1 < a.nlayers;
    const uint16_t* wnext = more ? (const uint16_t*)uniform_ptr(a.layers[l + 1].w) + q * 18 * 512 : nullptr;
    const unsigned want = epoch0 + (unsigned)l;
    const unsigned long long* slot = xrec + nb_slot + (unsigned)((l - 1) & 1) * CREC;
    const bool recv = (l > 0) && has_nb;

    uint4 P0p[4];
    uint2 P0s;
    unsigned long long hreg[6];
    f32x4 acc[3][3];
    {
      f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ly.bias) { const f32x4v t = *(g_cf4*)(uintptr_t)(ly.bias + 16 * q + 4 * g); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    }
    auto fetch_halo = [&]() {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int e = lane256z + 256 * k;              // record e: pixel e/32, channel pair e%32
        hreg[k] = ((e >> 5) < a.W) ? sc1_load64(slot, (unsigned)e) : ((unsigned long long)want << 32);
      }
    };
    auto halo_ok = [&]() -> bool {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < 6; ++k) ok = ok && ((unsigned)(hreg[k] >> 32) == want);
      return ok;
    };
    // next layer's filter registers, in two parts: the taps phase B still needs (tap row 0 for rh = 0, tap row 2 for
    // rh = 1: fragments 0..5 / 12..17) are replaced after phase B, all others right after the big sweep
    auto load_filters = [&](bool b_part) {
#pragma unroll
      for (int t = 0; t < 18; ++t) {
        const bool in_b = rh ? (t >= 12) : (t < 6);
        if (in_b == b_part) F[t] = as_bf16x8(gload16(wnext, (unsigned)(lane * 8 + t * 512)));
      }
    };

    // (1) the big sweep: layer 0 has its halos already; later layers run everything that does not touch the halo row
    if (l == 0) {
      if (rh == 0) chain_sweep<0, 0, 0, 18>(acc, F, wbase); else chain_sweep<1, 0, 0, 18>(acc, F, wbase);
    } else {
      if (rh == 0) chain_sweep<0, 1, 0, 18>(acc, F, wbase); else chain_sweep<1, 1, 0, 18>(acc, F, wbase);
    }
    CSTAMP(1);
    // epilogue geometry, recomputed per layer so that it does not occupy registers during the sweep
    const int gz = g + zero, gpair = 4 * (gz & ~1);
    auto strip_row = [&](int r) { return rh ? 5 - r : r; };
    unsigned poff[4], soff;        // element offsets in an [N,H,W,64] tensor, 0xffffffff = outside the image
    unsigned plds[4], slds;        // byte offsets of the same vectors inside an LDS input buffer (interior pixel)
  #pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = (k < 3) ? k : ((gz & 1) ? 2 : 1), c = (k < 3) ? (gz & 1) : 2;
      const int sr = strip_row(r), y = sy * CSH + sr, xx = 16 * c + pxz;
      poff[k] = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
      plds[k] = (unsigned)((q >> 1) * CHALF + ((sr + 1) * CCOLS + xx + 1) * CSTRIDE + ((q & 1) * 16 + gpair) * 2);
    }
    {
      const int sr = strip_row(0), y = sy * CSH + sr, xx = 32 + pxz;
      soff = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + 4 * gz) : 0xffffffffu;
      slds = (unsigned)((q >> 1) * CHALF + ((sr + 1) * CCOLS + xx + 1) * CSTRIDE + ((q & 1) * 16 + 4 * gz) * 2);
    }
    // (2) requests whose latency the early epilogue hides: neighbour records, epilogue operand, most of the next filter
    if (recv) fetch_halo();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      P0p[k] = make_uint4(0, 0, 0, 0);
      if (p0) P0p[k] = gload16(p0, (poff[k] != 0xffffffffu) ? poff[k] : 0u);
    }
    P0s = make_uint2(0, 0);
    if (p0) P0s = gload8(p0, (soff != 0xffffffffu) ? soff : 0u);
    if (more) load_filters(false);

    // (3) epilogue in the paired layout: HBM output + next layer's LDS input (+ the edge row as tagged records)
    uint16_t* outp = (uint16_t*)ly.out;
    unsigned long long* const myrec = xrec + my_slot + (unsigned)(l & 1) * CREC;
    const unsigned long long tag = (unsigned long long)(epoch0 + (unsigned)l + 1u) << 32;
    const bool send = more && has_nb;
    auto own = [&](f32x4 t) -> f32x4 {
      if (ly.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = fmaxf(t[j], 0.f);
      }
      if (ly.scale != 1.0f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] *= ly.scale;
      }
      return t;
    };
    auto do_pair = [&](auto KC) {
      constexpr int k = decltype(KC)::value;
      const f32x4 tx = own((k < 3) ? acc[k < 3 ? k : 0][0] : acc[1][2]);
      const f32x4 ty = own((k < 3) ? acc[k < 3 ? k : 0][1] : acc[2][2]);
      float v[8], m[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float snd = (g & 1) ? tx[j] : ty[j];
        const float rcv = __shfl_xor(snd, 16);
        v[j] = (g & 1) ? rcv : tx[j];
        v[4 + j] = (g & 1) ? ty[j] : rcv;
      }
      const bool in = poff[k] != 0xffffffffu;
      if (ly.mask) {
        unpack4_bf16(make_uint2(P0p[k].x, P0p[k].y), *reinterpret_cast<float(*)[4]>(&m[0]));
        unpack4_bf16(make_uint2(P0p[k].z, P0p[k].w), *reinterpret_cast<float(*)[4]>(&m[4]));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (m[j] > 0.f) ? v[j] : 0.f;
      }
      if (!ly.mask && p0) {
        unpack4_bf16(make_uint2(P0p[k].x, P0p[k].y), *reinterpret_cast<float(*)[4]>(&m[0]));
        unpack4_bf16(make_uint2(P0p[k].z, P0p[k].w), *reinterpret_cast<float(*)[4]>(&m[4]));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += m[j];
      }
      if (p1 && in) {     // second and third operands are rare (long skips): loaded on demand
        const uint4 t = gload16(p1, poff[k]);
        unpack4_bf16(make_uint2(t.x, t.y), *reinterpret_cast<float(*)[4]>(&m[0]));
        unpack4_bf16(make_uint2(t.z, t.w), *reinterpret_cast<float(*)[4]>(&m[4]));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += m[j];
      }
      if (p2 && in) {
        const uint4 t = gload16(p2, poff[k]);
        unpack4_bf16(make_uint2(t.x, t.y), *reinterpret_cast<float(*)[4]>(&m[0]));
        unpack4_bf16(make_uint2(t.z, t.w), *reinterpret_cast<float(*)[4]>(&m[4]));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += m[j];
      }
      uint4 o = make_uint4(0, 0, 0, 0);            // pixels outside the image are the next layer's zero padding
      if (in) {
        const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
        o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (k == 0 && send) {
          const unsigned r = (unsigned)((((g & 1) ? 16 : 0) + pxz) * 32 + 8 * q + (gpair >> 1));
          sc1_store64(myrec, r + 0, tag | o.x); sc1_store64(myrec, r + 1, tag | o.y);
          sc1_store64(myrec, r + 2, tag | o.z); sc1_store64(myrec, r + 3, tag | o.w);
        }
        gstore16(outp, poff[k], o);
      }
      if (more) *reinterpret_cast<uint4*>(nxt + plds[k]) = o;
    };
    auto do_single = [&]() {   // the unpaired tile (edge row, column tile 2): 8-byte path
      const f32x4 t = own(acc[0][2]);
      float v[4] = {t[0], t[1], t[2], t[3]};
      float m[4];
      if (ly.mask) {
        unpack4_bf16(P0s, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (m[j] > 0.f) ? v[j] : 0.f;
      }
      if (!ly.mask && p0) {
        unpack4_bf16(P0s, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += m[j];
      }
      if (p1 && soff != 0xffffffffu) {
        unpack4_bf16(gload8(p1, soff), m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += m[j];
      }
      uint2 o = make_uint2(0, 0);
      if (soff != 0xffffffffu) {
        if (p2) {
          unpack4_bf16(gload8(p2, soff), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        o = pack4_bf16(v[0], v[1], v[2], v[3]);
        if (send) {
          const unsigned r = (unsigned)((32 + pxz) * 32 + 8 * q + 2 * g);
          sc1_store64(myrec, r + 0, tag | o.x); sc1_store64(myrec, r + 1, tag | o.y);
        }
        gstore8(outp, soff, o);
      }
      if (more) *reinterpret_cast<uint2*>(nxt + slds) = o;
    };
    using std::integral_constant;
    // (3a) early: the two rows that are already final
    do_pair(integral_constant<int, 1>{});
    do_pair(integral_constant<int, 2>{});
    do_pair(integral_constant<int, 3>{});
    CSTAMP(2);
    // (4) the halo row: check the tags (requested in (2)), LDS rows 0 / 7 of the current buffer, the 18 MFMAs that use it
    if (l > 0) {
      if (recv) {
        unsigned spins = 0;
        while (!__all(halo_ok())) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > CSPIN_LIMIT) { if (lane == 0) atomicExch(a.status, 0x100u + (unsigned)l); break; }
          fetch_halo();
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int e = lane256z + 256 * k, hp = e >> 5, cp = e & 31;
          *reinterpret_cast<unsigned*>(cur + (cp >> 4) * CHALF + (((rh == 0) ? 0 : 7) * CCOLS + hp + 1) * CSTRIDE + (cp & 15) * 4) = (unsigned)hreg[k];
        }
      }
      CSTAMP(3);
      __syncthreads();
      CSTAMP(4);
      if (rh == 0) chain_sweep<0, 2, 0, 18>(acc, F, wbase); else chain_sweep<1, 2, 0, 18>(acc, F, wbase);
    }
    CSTAMP(5);
    if (more) load_filters(true);
    // (3b) late: the edge row, which is also what the neighbour waits for
    do_pair(integral_constant<int, 0>{});
    do_single();
    CSTAMP(6);
    if (more) __syncthreads();      // the next layer's input is complete in LDS
    CSTAMP(7);
    buf ^= 1;
  }
#undef CSTAMP
}

__global__ void chain_epoch_bump_kernel(unsigned* hdr, unsigned inc) {
  unsigned v = hdr[0] + inc;
  if (v > 0xfff00000u) v = 1u;     // wrap long before 2^32 (records that old have been overwritten thousands of times)
  hdr[0] = v;
}

extern "C" int64_t rumpy_conv_chain_xchg_bytes(int32_t nstrips) {
  return nstrips <= 0 ? 0 : 8ll * (CXHDR + (int64_t)nstrips * 4 * CREC);
}

extern "C" int rumpy_conv_chain(const rumpy_chain_args* p, void* stream) {
  if (!p || !p->x || !p->layers || !p->xchg || !p->status || p->nlayers <= 0) { rumpy_set_error("rumpy_conv_chain: bad argument"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->W > CSW) { rumpy_set_error("rumpy_conv_chain: needs 0 < W <= 48 (got %d)", p->W); return RUMPY_E_ARG; }
  const int sy_n = (p->H + CSH - 1) / CSH;
  const int nstrips = p->N * sy_n;
  if (nstrips > rumpy_device_cus()) { rumpy_set_error("rumpy_conv_chain: %d strips do not fit %d CUs (all must be co-resident)", nstrips, rumpy_device_cus()); return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(p->status, 0, sizeof(unsigned), s) != hipSuccess) { rumpy_set_error("rumpy_conv_chain: hipMemsetAsync failed"); return RUMPY_E_LAUNCH; }
  ChainDev d;
  d.x = (const uint16_t*)p->x; d.layers = p->layers; d.nlayers = p->nlayers; d.N = p->N; d.H = p->H; d.W = p->W; d.sy_n = sy_n;
  d.xchg = (unsigned long long*)p->xchg; d.status = p->status; d.stamps = (unsigned long long*)p->stamps;
  rumpy_probe_pre(4, s);
  hipLaunchKernelGGL(conv_chain_kernel, dim3(nstrips), dim3(CTHREADS), 0, s, d);
  rumpy_probe_post(4, s);
  // records of this launch carry epochs base+1 .. base+nlayers-1: move the base past them for the next launch / replay
  hipLaunchKernelGGL(chain_epoch_bump_kernel, dim3(1), dim3(1), 0, s, (unsigned*)p->xchg, (unsigned)p->nlayers + 1u);
  return rumpy_check_launch("rumpy_conv_chain");
}

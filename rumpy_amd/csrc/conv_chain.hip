// Layer-resident chain of 3x3 convolutions 64 -> 64 (the EDSR body, forward or data-gradient direction) in ONE launch.
//
// A 64->64 layer on the headline batch is 288 pixels per CU: run as one kernel per layer it is bound by the kernel
// boundary (launch + write-back of the 9.4 MB output), the first HBM loads and the epilogue drain, not by its 2.6 us
// of MFMAs.  Here every workgroup keeps ITS strip (6 rows x <=48 columns of one image, strip kernel geometry, see
// conv_strip.hip) for the whole chain: layer l's epilogue writes its bf16 output both to HBM (needed by the backward
// pass / as residual source) and straight into the other LDS buffer in B-fragment layout, where it is layer l+1's
// input.  Only the two halo rows come from the vertical neighbours (the strips above / below in the same image):
//   producer: the 2 edge rows -> exchange buffer with write-through (sc1) 8-byte stores; every wave drains them with a
//             counted s_waitcnt, workgroup barrier, ONE lane publishes flag[strip] = layers done (sc1 store);
//   consumer: each wave polls the flag of the neighbour it depends on (relaxed sc1 load, s_sleep, BOUNDED spin),
//             then fetches the halo row with sc1 loads (MI355X_MICROARCH.md, visibility table row 1: no fences).
// The halo wait hides behind the MFMAs that do not touch the halo row (8 of every 9); the filter of the next layer is
// prefetched under the epilogue.  Needs all strips co-resident: grid = N * ceil(H/6) <= number of CUs, one 512-thread
// workgroup per CU (153.6 KB LDS), W <= 48.  A timed-out spin sets *status and the kernel still terminates.
#include "common.hpp"

constexpr int CSH = 6, CSW = 48;
constexpr int CROWS = CSH + 2, CCOLS = CSW + 2, CPIX = CROWS * CCOLS;   // 8 x 50 halo pixels
constexpr int CSTRIDE = 96;                                             // bytes per 32-channel pixel half in LDS
constexpr int CHALF = CPIX * CSTRIDE;                                   // 38400
constexpr int CSTAGE = 2 * CHALF;                                       // 76800
constexpr int CTHREADS = 512;
constexpr int CPIECES = CPIX * 8, CREGS = (CPIECES + CTHREADS - 1) / CTHREADS;
constexpr unsigned CSPIN_LIMIT = 1u << 22;

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct ChainDev {
  const uint16_t* x; const rumpy_chain_layer* layers; int nlayers, N, H, W, sy_n;
  uint16_t* xchg; unsigned* flags; unsigned* status; unsigned long long* stamps;
};

__device__ __forceinline__ void sc1_store64(void* p, unsigned long long v) {
  __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long sc1_load64(const void* p) {
  return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one (channel half, tap column, column tile) group of the MFMA sweep.  EDGE selects which MFMAs run:
//   0 = all 9 (rows r = 0..2, taps ky = 0..2)            (layer 0: halos come with the strip load)
//   1 = the 8 that do not read the halo row              (phase A)
//   2 = the single one that reads the halo row            (phase B)
// RH = 0: the halo row is input row 0 of this wave's window (out row 0, ky 0); RH = 1: window row 4 (out row 2, ky 2).
template <int RH, int EDGE>
__device__ __forceinline__ void chain_group(f32x4 (&acc)[3][3], const bf16x8 (&F)[18], const unsigned char* cur, int half, int kx, int c) {
  // cur points at (window row 0, column px) of channel half `half`, lane's 16-byte k-group included
  bf16x8 I[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const bool is_halo = (RH == 0) ? (r == 0) : (r == 4);
    if ((EDGE == 0) || (EDGE == 1 && !is_halo) || (EDGE == 2 && is_halo))
      I[r] = *reinterpret_cast<const bf16x8*>(cur + (r * CCOLS + 16 * c + kx) * CSTRIDE);
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const bool uses_halo = (RH == 0) ? (r + ky == 0) : (r + ky == 4);
      if ((EDGE == 0) || (EDGE == 1 && !uses_halo) || (EDGE == 2 && uses_halo))
        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[(ky * 3 + kx) * 2 + half], I[r + ky], acc[r][c], 0, 0, 0);
    }
}
// HALVES: bit 0 = channel half 0, bit 1 = channel half 1
template <int RH, int EDGE, int HALVES>
__device__ __forceinline__ void chain_sweep(f32x4 (&acc)[3][3], const bf16x8 (&F)[18], const unsigned char* base) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (!((HALVES >> half) & 1)) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int c = 0; c < 3; ++c) chain_group<RH, EDGE>(acc, F, base + half * CHALF, half, kx, c);
  }
}

__global__ void __launch_bounds__(CTHREADS, 2) conv_chain_kernel(ChainDev a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * CSTAGE];
  __shared__ unsigned drained;                   // waves that have drained their stores, summed over layers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) drained = 0;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = wave >> 2;
  const int strip = blockIdx.x;
  const int n = strip / a.sy_n, sy = strip - n * a.sy_n;
  const int c0 = 16 * q + 4 * g;                 // this lane's 4 consecutive output channels
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
    const uint4* wp = reinterpret_cast<const uint4*>(a.layers[0].w) + (size_t)q * 18 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
  }
  __syncthreads();

  // element offsets of this lane's 9 output vectors in an [N,H,W,64] tensor (0xffffffff = outside the image)
  unsigned off[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int y = sy * CSH + 3 * rh + r;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int xx = 16 * c + px;
      off[r][c] = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0) : 0xffffffffu;
    }
  }
  // halo hand-off geometry: rows of the exchange buffer are [48 px][64 ch]; this wave group needs ONE neighbour row
  const bool has_nb = (rh == 0) ? (sy > 0) : (sy + 1 < a.sy_n && (sy + 1) * CSH < a.H);
  const int nb_strip = (rh == 0) ? strip - 1 : strip + 1;
  const size_t xrow = (size_t)CSW * 64;                        // elements per exchanged row

  int buf = 0;
#define CSTAMP(k) do { if (a.stamps && lane == 0 && l < 8) a.stamps[(((size_t)strip * 8 + wave) * 8 + l) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  for (int l = 0; l < a.nlayers; ++l) {
    CSTAMP(0);
    const rumpy_chain_layer ly = a.layers[l];
    const uint16_t* p0 = (const uint16_t*)(ly.mask ? ly.mask : ly.res1);
    const uint16_t* p1 = (const uint16_t*)(ly.mask ? ly.res1 : ly.res2);
    const uint16_t* p2 = (const uint16_t*)((ly.mask && ly.res1) ? ly.res2 : nullptr);
    unsigned char* cur = lds + buf * CSTAGE;
    unsigned char* nxt = lds + (buf ^ 1) * CSTAGE;
    const unsigned char* wbase = cur + (3 * rh * CCOLS + px) * CSTRIDE + g * 16;

    // (2) epilogue operands (mask / residuals) for this layer
    uint2 P0[3][3], P1[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const unsigned oc = (off[r][c] != 0xffffffffu) ? off[r][c] : 0u;
        P0[r][c] = make_uint2(0, 0); P1[r][c] = make_uint2(0, 0);
        if (p0) P0[r][c] = *reinterpret_cast<const uint2*>(p0 + oc);
        if (p1) P1[r][c] = *reinterpret_cast<const uint2*>(p1 + oc);
      }
    // (3) MFMAs.  Layer 0 has its halos already; later layers first run everything that does not touch the halo row.
    f32x4 acc[3][3];
    {
      f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ly.bias) { const float4 t = *reinterpret_cast<const float4*>(ly.bias + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    }
    if (l == 0) {
      if (rh == 0) chain_sweep<0, 0, 3>(acc, F, wbase); else chain_sweep<1, 0, 3>(acc, F, wbase);
    } else {
      // halo row of layer l's input = the neighbour's edge row of layer l-1.  The neighbour publishes it a hand-off latency
      // after ITS epilogue, i.e. while we run phase A: try once after the first channel half (the fetch then hides under
      // the second half), otherwise wait for it after phase A.
      const unsigned* nbflag = a.flags + nb_strip;
      const uint16_t* row = a.xchg + ((size_t)(((l - 1) & 1) * gridDim.x + nb_strip) * 2 + (rh == 0 ? 1 : 0)) * xrow;
      unsigned long long hreg[3] = {0ull, 0ull, 0ull};
      if (rh == 0) chain_sweep<0, 1, 1>(acc, F, wbase); else chain_sweep<1, 1, 1>(acc, F, wbase);
      // publish layer l-1 of THIS strip without a barrier: every wave drains its own edge stores (issued before this
      // layer's first half-sweep, so long done), counts itself in LDS, and the wave whose count completes the layer signals
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const unsigned old = __hip_atomic_fetch_add(&drained, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old == 8u * (unsigned)l - 1u)
          __hip_atomic_store((gu32*)(a.flags + strip), (unsigned)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      CSTAMP(1);
      bool fetched = !has_nb;
      if (!fetched && __hip_atomic_load((gu32*)nbflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)l) {
#pragma unroll
        for (int k = 0; k < 3; ++k) hreg[k] = sc1_load64(row + (size_t)(lane256 + 256 * k) * 4);
        fetched = true;
      }
      if (rh == 0) chain_sweep<0, 1, 2>(acc, F, wbase); else chain_sweep<1, 1, 2>(acc, F, wbase);
      CSTAMP(2);
      if (!fetched) {
        unsigned spins = 0;
        while (__hip_atomic_load((gu32*)nbflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)l) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > CSPIN_LIMIT) { if (lane == 0) atomicExch(a.status, 0x100u + (unsigned)l); break; }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) hreg[k] = sc1_load64(row + (size_t)(lane256 + 256 * k) * 4);
      }
      // halo row -> LDS (row 0 or 7 of the current buffer; 8-byte pieces: pixel e/16, channels 4*(e%16)..)
      if (has_nb) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int e = lane256 + 256 * k, hp = e >> 4, sub = e & 15;
          *reinterpret_cast<unsigned long long*>(cur + (sub >> 3) * CHALF + (((rh == 0) ? 0 : 7) * CCOLS + hp + 1) * CSTRIDE + (sub & 7) * 8) = hreg[k];
        }
      }
      CSTAMP(3);
      __syncthreads();
      CSTAMP(4);
      if (rh == 0) chain_sweep<0, 2, 3>(acc, F, wbase); else chain_sweep<1, 2, 3>(acc, F, wbase);
    }
    CSTAMP(5);
    // (4) next layer's filter slice: L2 hits that land under the epilogue
    if (l + 1 < a.nlayers) {
      const uint4* wp = reinterpret_cast<const uint4*>(a.layers[l + 1].w) + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
    }
    // (5) epilogue: values, then [edge row -> exchange buffer (sc1)] first, [HBM output + next LDS input] after
    uint2 o[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v[4] = {acc[r][c][0], acc[r][c][1], acc[r][c][2], acc[r][c][3]};
        if (ly.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (ly.scale != 1.0f) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= ly.scale;
        }
        float m[4];
        if (ly.mask) {
          unpack4_bf16(P0[r][c], m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (m[j] > 0.f) ? v[j] : 0.f;
        }
        if (!ly.mask && p0) {
          unpack4_bf16(P0[r][c], m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        if (p1) {
          unpack4_bf16(P1[r][c], m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        if (p2 && off[r][c] != 0xffffffffu) {
          unpack4_bf16(*reinterpret_cast<const uint2*>(p2 + off[r][c]), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        o[r][c] = (off[r][c] != 0xffffffffu) ? pack4_bf16(v[0], v[1], v[2], v[3]) : make_uint2(0, 0);   // outside = padding
      }
    const bool more = l + 1 < a.nlayers;
    if (more) {
      uint16_t* xr = a.xchg + ((size_t)((l & 1) * gridDim.x + strip) * 2 + rh) * xrow;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const uint2 e = (rh == 0) ? o[0][c] : o[2][c];       // my edge row of the strip (static register selection)
        sc1_store64(xr + (size_t)(16 * c + px) * 64 + c0, ((unsigned long long)e.y << 32) | e.x);
      }
    }
    uint16_t* outp = (uint16_t*)ly.out;
    uint16_t* sink = a.xchg + (size_t)2 * gridDim.x * 2 * xrow;   // 64 spare elements: target of out-of-image lanes
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        *reinterpret_cast<uint2*>((off[r][c] != 0xffffffffu) ? outp + off[r][c] : sink + 4 * (lane & 15)) = o[r][c];
        if (more)   // next layer's input: interior pixel (row 3rh+r+1, column 16c+px+1) of the other buffer
          *reinterpret_cast<uint2*>(nxt + (c0 >> 5) * CHALF + ((3 * rh + r + 1) * CCOLS + 16 * c + px + 1) * CSTRIDE + (c0 & 31) * 2) = o[r][c];
      }
    CSTAMP(6);
    if (more) __syncthreads();      // the next layer's input is complete in LDS; its flag is published from inside layer l+1
    CSTAMP(7);
    buf ^= 1;
  }
#undef CSTAMP
}

extern "C" int64_t rumpy_conv_chain_xchg_elems(int32_t nstrips) { return (int64_t)2 * nstrips * 2 * CSW * 64 + 64; }

extern "C" int rumpy_conv_chain(const rumpy_chain_args* p, void* stream) {
  if (!p || !p->x || !p->layers || !p->xchg || !p->flags || !p->status || p->nlayers <= 0) { rumpy_set_error("rumpy_conv_chain: bad argument"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->W > CSW) { rumpy_set_error("rumpy_conv_chain: needs 0 < W <= 48 (got %d)", p->W); return RUMPY_E_ARG; }
  const int sy_n = (p->H + CSH - 1) / CSH;
  const int nstrips = p->N * sy_n;
  if (nstrips > rumpy_device_cus()) { rumpy_set_error("rumpy_conv_chain: %d strips do not fit %d CUs (all must be co-resident)", nstrips, rumpy_device_cus()); return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  // polled words are re-initialised on the stream before every launch (a memset node when captured)
  if (hipMemsetAsync(p->flags, 0, sizeof(unsigned) * nstrips, s) != hipSuccess || hipMemsetAsync(p->status, 0, sizeof(unsigned), s) != hipSuccess) {
    rumpy_set_error("rumpy_conv_chain: hipMemsetAsync failed"); return RUMPY_E_LAUNCH; }
  ChainDev d;
  d.x = (const uint16_t*)p->x; d.layers = p->layers; d.nlayers = p->nlayers; d.N = p->N; d.H = p->H; d.W = p->W; d.sy_n = sy_n;
  d.xchg = (uint16_t*)p->xchg; d.flags = p->flags; d.status = p->status; d.stamps = (unsigned long long*)p->stamps;
  rumpy_probe_pre(4, s);
  hipLaunchKernelGGL(conv_chain_kernel, dim3(nstrips), dim3(CTHREADS), 0, s, d);
  rumpy_probe_post(4, s);
  return rumpy_check_launch("rumpy_conv_chain");
}

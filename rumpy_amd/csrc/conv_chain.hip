// Layer-resident chain of 3x3 convolutions 64 -> 64 (the EDSR body, forward or data-gradient direction) in ONE launch.
//
// A 64->64 layer on the headline batch is 288 pixels per CU: run as one kernel per layer it pays, every layer, the kernel
// boundary (dispatch + write-back of the 9.4 MB output), the first HBM loads and the store drain around 2.5 us of MFMAs.
// Here every workgroup keeps ITS strip (6 rows x <=48 columns of one image; geometry, wave roles, MFMA sweep and the
// paired 16-byte epilogue are those of conv_strip.hip) for the whole chain: layer l's epilogue writes its bf16 output to
// HBM (the backward pass and later residual adds need it) AND straight into the other LDS buffer in B-fragment layout,
// where it is layer l+1's input.  Only the two halo rows come from the vertical neighbours (strips above / below in the
// same image), through the output tensor itself:
//   producer: the strip's two edge rows are stored write-through (sc1); after the first half of the next layer's sweep
//             every wave drains its stores (s_waitcnt vmcnt(0)), counts itself in LDS, and the wave whose count completes
//             the layer publishes flag[strip] = layers done with one sc1 store (no workgroup barrier);
//   consumer: each wave reads the flag of the neighbour it depends on (relaxed sc1 load; bounded spin with s_sleep only if
//             it is late) and fetches the halo row with sc1 loads - MI355X_MICROARCH.md visibility table, row 1: no fences.
// The hand-off latency hides behind the 8 of every 9 MFMAs that do not read the halo row; the filter of the next layer is
// prefetched under the epilogue.  Needs all strips co-resident: grid = N * ceil(H/6) <= number of CUs, one 512-thread
// workgroup per CU (153.6 KB LDS), W <= 48, nothing else on the GPU.  A timed-out spin sets *status; the kernel terminates.
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
  unsigned* flags; unsigned* status; unsigned long long* stamps;
};

__device__ __forceinline__ void sc1_store64(void* p, unsigned long long v) {
  __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long sc1_load64(const void* p) {
  return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MFMA sweep over groups [G0, G1) of the 18 (channel half, tap column, column tile) groups, software pipelined (the reads
// of group i+1 are issued before the MFMAs of group i).  EDGE selects the part of each group:
//   0 = all 5 fragment reads / 9 MFMAs (layer 0: the strip was loaded with its halos)
//   1 = the 4 reads / 8 MFMAs that do not touch the halo row     2 = the 1 read / 1 MFMA that does
// RH = 0: the halo row is window row 0 (used by out row 0, ky 0); RH = 1: window row 4 (out row 2, ky 2).
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
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[(ky * 3 + kx) * 2 + half], I[grp & 1][r + ky], acc[r][c], 0, 0, 0);
      }
  }
}

__global__ void __launch_bounds__(CTHREADS, 2) conv_chain_kernel(ChainDev a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * CSTAGE];
  __shared__ unsigned drained;                   // waves that have drained their stores, summed over layers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = wave >> 2;
  const int strip = blockIdx.x;
  const int n = strip / a.sy_n, sy = strip - n * a.sy_n;
  const int lane256 = tid & 255;                 // index inside the 4-wave row group
  if (tid == 0) drained = 0;

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

  // ---- loop-invariant epilogue geometry (every tensor of the chain is [N,H,W,64]); pairing as in conv_strip.hip ----
  //   pair k<3: X = (row k, col tile 0), Y = (row k, col tile 1); pair 3: X = (0, 2), Y = (1, 2); single: (2, 2)
  //   even-g lanes finish X's pixel, odd-g lanes Y's pixel, 8 consecutive channels 16q + 4(g&~1) ..
  const int gpair = 4 * (g & ~1);
  auto pix_rc = [&](int k, int& r, int& c) {
    if (k < 3) { r = k; c = (g & 1) ? 1 : 0; } else { r = (g & 1) ? 1 : 0; c = 2; }
  };
  unsigned poff[4], soff;        // element offsets in an [N,H,W,64] tensor, 0xffffffff = outside the image
  unsigned plds[4], slds;        // byte offsets of the same vectors inside an LDS input buffer (interior pixel)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int r, c;
    pix_rc(k, r, c);
    const int y = sy * CSH + 3 * rh + r, xx = 16 * c + px;
    poff[k] = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
    plds[k] = (unsigned)((q >> 1) * CHALF + ((3 * rh + r + 1) * CCOLS + xx + 1) * CSTRIDE + ((q & 1) * 16 + gpair) * 2);
  }
  {
    const int y = sy * CSH + 3 * rh + 2, xx = 32 + px;
    soff = (y < a.H && xx < a.W) ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + 4 * g) : 0xffffffffu;
    slds = (unsigned)((q >> 1) * CHALF + ((3 * rh + 3) * CCOLS + xx + 1) * CSTRIDE + ((q & 1) * 16 + 4 * g) * 2);
  }
  // halo hand-off: this wave group needs ONE neighbour row of the previous layer's output
  const bool has_nb = (rh == 0) ? (sy > 0) : (sy + 1 < a.sy_n);
  const int nb_strip = (rh == 0) ? strip - 1 : strip + 1;
  const int halo_y = (rh == 0) ? sy * CSH - 1 : sy * CSH + CSH;
  const unsigned halo_base = (unsigned)((n * a.H + halo_y) * a.W) * 64;

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
    const bool more = l + 1 < a.nlayers;

    // (1) epilogue operands (mask / residuals), 16-byte vectors in the paired layout
    uint4 P0p[4], P1p[4];
    uint2 P0s, P1s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned oc = (poff[k] != 0xffffffffu) ? poff[k] : 0u;
      P0p[k] = make_uint4(0, 0, 0, 0); P1p[k] = make_uint4(0, 0, 0, 0);
      if (p0) P0p[k] = *reinterpret_cast<const uint4*>(p0 + oc);
      if (p1) P1p[k] = *reinterpret_cast<const uint4*>(p1 + oc);
    }
    {
      const unsigned oc = (soff != 0xffffffffu) ? soff : 0u;
      P0s = make_uint2(0, 0); P1s = make_uint2(0, 0);
      if (p0) P0s = *reinterpret_cast<const uint2*>(p0 + oc);
      if (p1) P1s = *reinterpret_cast<const uint2*>(p1 + oc);
    }
    // (2) MFMAs.  Layer 0 has its halos already; later layers first run everything that does not touch the halo row.
    f32x4 acc[3][3];
    {
      f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ly.bias) { const float4 t = *reinterpret_cast<const float4*>(ly.bias + 16 * q + 4 * g); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    }
    if (l == 0) {
      if (rh == 0) chain_sweep<0, 0, 0, 18>(acc, F, wbase); else chain_sweep<1, 0, 0, 18>(acc, F, wbase);
    } else {
      const unsigned* nbflag = a.flags + nb_strip;
      const uint16_t* row = (const uint16_t*)a.layers[l - 1].out + halo_base;
      unsigned long long hreg[3] = {0ull, 0ull, 0ull};
      if (rh == 0) chain_sweep<0, 1, 0, 9>(acc, F, wbase); else chain_sweep<1, 1, 0, 9>(acc, F, wbase);
      // publish layer l-1 of THIS strip without a barrier: every wave drains its own stores (issued before this sweep, so
      // long done), counts itself in LDS, and the wave whose count completes the layer signals
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const unsigned old = __hip_atomic_fetch_add(&drained, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old == 8u * (unsigned)l - 1u)
          __hip_atomic_store((gu32*)(a.flags + strip), (unsigned)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      CSTAMP(1);
      auto fetch_halo = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int e = lane256 + 256 * k;               // 8-byte piece: pixel e/16, channels 4*(e%16) ..
          if ((e >> 4) < a.W) hreg[k] = sc1_load64(row + (size_t)e * 4);
        }
      };
      bool fetched = !has_nb;
      if (!fetched && __hip_atomic_load((gu32*)nbflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)l) {
        fetch_halo();
        fetched = true;
      }
      if (rh == 0) chain_sweep<0, 1, 9, 18>(acc, F, wbase); else chain_sweep<1, 1, 9, 18>(acc, F, wbase);
      CSTAMP(2);
      if (!fetched) {
        unsigned spins = 0;
        while (__hip_atomic_load((gu32*)nbflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)l) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > CSPIN_LIMIT) { if (lane == 0) atomicExch(a.status, 0x100u + (unsigned)l); break; }
        }
        fetch_halo();
      }
      if (has_nb) {   // halo row -> LDS row 0 / 7 of the current buffer
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int e = lane256 + 256 * k, hp = e >> 4, sub = e & 15;
          *reinterpret_cast<unsigned long long*>(cur + (sub >> 3) * CHALF + (((rh == 0) ? 0 : 7) * CCOLS + hp + 1) * CSTRIDE + (sub & 7) * 8) = hreg[k];
        }
      }
      CSTAMP(3);
      __syncthreads();
      CSTAMP(4);
      if (rh == 0) chain_sweep<0, 2, 0, 18>(acc, F, wbase); else chain_sweep<1, 2, 0, 18>(acc, F, wbase);
    }
    CSTAMP(5);
    // (3) next layer's filter slice: L2 hits that land under the epilogue
    if (more) {
      const uint4* wp = reinterpret_cast<const uint4*>(a.layers[l + 1].w) + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(wp[t * 64]);
    }
    // (4) epilogue in the paired layout: HBM output (edge rows write-through) + next layer's LDS input
    uint16_t* outp = (uint16_t*)ly.out;
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
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 tx = own((k < 3) ? acc[k][0] : acc[0][2]);
      const f32x4 ty = own((k < 3) ? acc[k][1] : acc[1][2]);
      float v[8], m[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float send = (g & 1) ? tx[j] : ty[j];
        const float recv = __shfl_xor(send, 16);
        v[j] = (g & 1) ? recv : tx[j];
        v[4 + j] = (g & 1) ? ty[j] : recv;
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
      if (p1) {
        unpack4_bf16(make_uint2(P1p[k].x, P1p[k].y), *reinterpret_cast<float(*)[4]>(&m[0]));
        unpack4_bf16(make_uint2(P1p[k].z, P1p[k].w), *reinterpret_cast<float(*)[4]>(&m[4]));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += m[j];
      }
      if (p2 && in) {
        const uint4 t = *reinterpret_cast<const uint4*>(p2 + poff[k]);
        unpack4_bf16(make_uint2(t.x, t.y), *reinterpret_cast<float(*)[4]>(&m[0]));
        unpack4_bf16(make_uint2(t.z, t.w), *reinterpret_cast<float(*)[4]>(&m[4]));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += m[j];
      }
      uint4 o = make_uint4(0, 0, 0, 0);            // pixels outside the image are the next layer's zero padding
      if (in) {
        const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
        o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        // edge rows of the strip (row 0 for rh = 0: pair 0 and X of pair 3; row 5 for rh = 1: pair 2) go out write-through
        const bool edge = (rh == 0) ? (k == 0 || (k == 3 && !(g & 1))) : (k == 2);
        if (more && edge) {
          sc1_store64(outp + poff[k], ((unsigned long long)o.y << 32) | o.x);
          sc1_store64(outp + poff[k] + 4, ((unsigned long long)o.w << 32) | o.z);
        } else {
          *reinterpret_cast<uint4*>(outp + poff[k]) = o;
        }
      }
      if (more) *reinterpret_cast<uint4*>(nxt + plds[k]) = o;
    }
    {   // the single unpaired tile (row 2, column tile 2): 8-byte path; an edge row for rh = 1
      const f32x4 t = own(acc[2][2]);
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
      if (p1) {
        unpack4_bf16(P1s, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += m[j];
      }
      uint2 o = make_uint2(0, 0);
      if (soff != 0xffffffffu) {
        if (p2) {
          unpack4_bf16(*reinterpret_cast<const uint2*>(p2 + soff), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        o = pack4_bf16(v[0], v[1], v[2], v[3]);
        if (more && rh == 1) sc1_store64(outp + soff, ((unsigned long long)o.y << 32) | o.x);
        else *reinterpret_cast<uint2*>(outp + soff) = o;
      }
      if (more) *reinterpret_cast<uint2*>(nxt + slds) = o;
    }
    CSTAMP(6);
    if (more) __syncthreads();      // the next layer's input is complete in LDS; its flag is published from inside layer l+1
    CSTAMP(7);
    buf ^= 1;
  }
#undef CSTAMP
}

extern "C" int rumpy_conv_chain(const rumpy_chain_args* p, void* stream) {
  if (!p || !p->x || !p->layers || !p->flags || !p->status || p->nlayers <= 0) { rumpy_set_error("rumpy_conv_chain: bad argument"); return RUMPY_E_ARG; }
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
  d.flags = p->flags; d.status = p->status; d.stamps = (unsigned long long*)p->stamps;
  rumpy_probe_pre(4, s);
  hipLaunchKernelGGL(conv_chain_kernel, dim3(nstrips), dim3(CTHREADS), 0, s, d);
  rumpy_probe_post(4, s);
  return rumpy_check_launch("rumpy_conv_chain");
}

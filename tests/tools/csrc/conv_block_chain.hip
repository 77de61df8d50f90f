// A chain of residual blocks (the EDSR body, forward or data-gradient direction) in ONE launch: conv_block.hip's block with
// the strip resident in LDS from block to block.
//
// conv_block.hip pays per block: the kernel boundary (write-back + dispatch, ~4 us), the first loads of the 10-row input
// tile (~2.6 us) and 1.67x input traffic.  Here every workgroup keeps its strip (6 rows x <= 48 columns of one image) for the
// whole chain: a block's output goes to HBM (later passes read it) AND in place into the LDS input image; only the two
// rows above and the two rows below come from the vertical neighbours, as epoch-tagged 8-byte records {2 bf16 channels,
// 32-bit tag} written with write-through (sc1) 64-bit stores and polled with sc1 loads - the data is the signal: no flags, no
// fences, no drains (the protocol of conv_chain.hip, MI355X_MICROARCH.md visibility table row 1).  The wait hides behind the
// half of the first convolution that does not touch the halo rows:
//   requests for the neighbour's records -> first conv on the 2 rows per wave that need no halo (108 MFMAs) -> tags checked,
//   halo rows -> LDS, barrier -> first conv on the 2 halo-dependent rows -> epilogue 1 (T -> HBM + LDS), barrier -> second
//   conv -> epilogue 2: output -> HBM + LDS (in place) + edge rows as records, barrier.
// Slots are double-buffered by block parity (a neighbour is at most one block ahead: it needs this strip's edge rows to go
// further); the tag base lives in the exchange buffer's header and is advanced by a one-thread kernel after every launch, so
// repeated launches and hipGraph replays never match stale records.
// Needs every strip co-resident: N * ceil(H/6) <= CUs (one 512-thread workgroup per CU, 115 KB LDS), W <= 48, and no other
// kernel occupying CUs.  A timed-out spin sets *status (the host raises on it); the kernel always terminates.
#include "block_common.hpp"
#include "rumpy_experimental.h"
#include <type_traits>

#ifndef BCHAIN_ABL
#define BCHAIN_ABL 0     // 9: phase stamps of block 8 into a.status + 16 words... (diagnostic builds only, tests/tools/build_abl.sh)
#endif
constexpr int BCREC = 2 * BSW * 32;     // 8-byte records of one edge: 2 rows x 48 pixels x 32 channel pairs
constexpr int BCHDR = 8;                // header records (word 0 = tag base)
constexpr unsigned BCSPIN = 1u << 20;

typedef unsigned int bc_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int bc_u32x2 __attribute__((ext_vector_type(2)));
typedef float bc_f32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t bc_rsrc;

struct BChainDev {
  const rumpy_block_args* blocks; int nblocks, N, H, W, sy_n;
  unsigned long long* xchg; unsigned xchg_bytes; unsigned* status;
};

// Every global access is a raw BUFFER instruction: a uniform resource (base + size in SGPRs), a 32-bit per-lane byte
// offset, and the hardware's range check - a lane whose offset is BC_OOB loads zeros / stores nothing.  That makes every load
// and store unconditional (one instruction on every path, no exec-mask branches): the compiler can then wait with counted
// vmcnt values.  A conditional memory operation makes it drain the whole queue instead - including write-through stores
// that take microseconds - and a per-lane pointer select would cost 64-bit address arithmetic and long-lived register pairs.
constexpr unsigned BC_OOB = 0x80000000u;        // beyond every buffer of the path (the largest is 151 MB)
constexpr int BC_SC1 = 16;                      // cache-policy bit of the buffer intrinsics: system-coherent (write-through / L2) access
template <typename T>
__device__ __forceinline__ T* bc_uniform(T* p) {   // fields of the device-side block table are uniform: move them to SGPRs
  const unsigned long long v = (unsigned long long)(uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ bc_rsrc bc_buffer(const void* p, unsigned bytes) {   // p == NULL: an empty buffer (all lanes out of range)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? bytes : 0u, 0x00020000);
}
__device__ __forceinline__ uint4 bc_load16(bc_rsrc r, unsigned byte) {
  const bc_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, byte, 0, 0);
  return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint2 bc_load8(bc_rsrc r, unsigned byte) {
  const bc_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, byte, 0, 0);
  return make_uint2(t.x, t.y);
}
__device__ __forceinline__ void bc_store16(bc_rsrc r, unsigned byte, uint4 v) { __builtin_amdgcn_raw_buffer_store_b128((bc_u32x4){v.x, v.y, v.z, v.w}, r, byte, 0, 0); }
__device__ __forceinline__ void bc_store8(bc_rsrc r, unsigned byte, uint2 v) { __builtin_amdgcn_raw_buffer_store_b64((bc_u32x2){v.x, v.y}, r, byte, 0, 0); }
// one tagged record {2 bf16 channels, tag}: a single 8-byte write-through store / L2 load
__device__ __forceinline__ void bc_send(bc_rsrc r, unsigned byte, unsigned data, unsigned tag) {
  __builtin_amdgcn_raw_buffer_store_b64((bc_u32x2){data, tag}, r, byte, 0, BC_SC1);
}
__device__ __forceinline__ uint2 bc_recv(bc_rsrc r, unsigned byte) {
  const bc_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, byte, 0, BC_SC1);
  return make_uint2(t.x, t.y);
}

// The block loop, specialised on the row group (RH = wave / 4): both copies run in the same workgroup and meet at the same
// s_barrier instructions (a wave-uniform split; the barrier counts waves, not program counters).
template <int RH, bool MASK>
__device__ __forceinline__ void block_chain_body(const BChainDev& a, unsigned char* lds, int tid, int lane, int q, int px_, int g_,
                                                 int strip, int n, int sy) {
  unsigned char* const ldx = lds;
  unsigned char* const ldt = lds + BXBYTES;
  const bool has_nb = (RH == 0) ? (sy > 0) : (sy + 1 < a.sy_n);
  const unsigned epoch0 = *reinterpret_cast<const unsigned*>(a.xchg);
  const bc_rsrc rrec = bc_buffer(a.xchg, a.xchg_bytes);
  const unsigned tensor_bytes = (unsigned)(a.N * a.H * a.W) * 128u;
  // slot (strip, edge 0 = top rows 0,1 / 1 = bottom rows 4,5, block parity): BCREC records after the header
  const unsigned my_slot = (unsigned)(BCHDR + ((strip * 2 + RH) * 2) * BCREC) * 8u;
  const unsigned nb_slot = (unsigned)(BCHDR + ((((RH == 0) ? strip - 1 : strip + 1) * 2 + (1 - RH)) * 2) * BCREC) * 8u;
  // image row of the first halo row this group receives, its row in the LDS input image
  const int halo_y0 = (RH == 0) ? sy * BSH - 2 : sy * BSH + BSH;
  const int halo_xrow0 = (RH == 0) ? 0 : BSH + 2;
  const int px0 = px_, g0 = g_, lane256_0 = tid & 255;

  bf16x8 F[18];
  bc_f32x4 bias1;
  {
    const bc_rsrc rw = bc_buffer(bc_uniform(a.blocks[0].w1), 64 * 576 * 2);
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(bc_load16(rw, (unsigned)(q * 18 * 1024 + lane * 16 + t * 1024)));
    const bc_rsrc rb = bc_buffer(bc_uniform(a.blocks[0].b1), 256);
    const uint4 t = bc_load16(rb, (unsigned)(16 * q + 4 * g0) * 4u);
    bias1 = (bc_f32x4){__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};   // zeros without a bias
  }

  unsigned long long stamps[10];
  int nst = 0;
#define BC_STAMP() do { if (BCHAIN_ABL == 9 && b == 8 && nst < 10) stamps[nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)
  for (int b = 0; b < a.nblocks; ++b) {
    // an opaque zero per block: the lane geometry derived from it is recomputed here instead of being hoisted out of the
    // block loop into dozens of long-lived registers (which spill: conv_chain.hip)
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    const int px = px0 + zero, g = g0 + zero, lane256 = lane256_0 + zero;
    const int c0 = 16 * q + 4 * g;
    const int gpair = 4 * (g & ~1);
    const int chunk8 = 2 * q + (gpair >> 3);
    rumpy_block_args d = a.blocks[b];
    const bool more = b + 1 < a.nblocks;
    const bc_rsrc rmask = bc_buffer(bc_uniform(d.mask), tensor_bytes);
    const bc_rsrc rt = bc_buffer(bc_uniform(d.t), tensor_bytes);
    const bc_rsrc rout = bc_buffer(bc_uniform(d.out), tensor_bytes);
    const bc_rsrc rw2 = bc_buffer(bc_uniform(d.w2), 64 * 576 * 2);
    const bc_rsrc rb2 = bc_buffer(bc_uniform(d.b2), 256);
    const bc_rsrc rw1n = bc_buffer(bc_uniform(a.blocks[more ? b + 1 : b].w1), 64 * 576 * 2);
    const bc_rsrc rb1n = bc_buffer(bc_uniform(a.blocks[more ? b + 1 : b].b1), 256);
    const int relu1 = __builtin_amdgcn_readfirstlane(d.relu1);
    const float scale1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(d.scale1)));
    const float scale2 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(d.scale2)));
    const bool recv = (b > 0) && has_nb;
    const unsigned want = epoch0 + (unsigned)b;
    const unsigned slot = (recv ? nb_slot : my_slot) + (unsigned)((b + 1) & 1) * (BCREC * 8u);

    // (1) requests that land under the first half of convA: the neighbour's records (an unused record - beyond the image - is
    //     requested out of range and reads as zero; validity is applied when the tags are checked)
    uint2 hreg[12];
    auto rec_used = [&](int k) -> bool {
      const int e = lane256 + 256 * k;                 // record e: halo row e / 1536, pixel (e % 1536) / 32, channel pair e % 32
      const int hr = e / (BSW * 32), pixel = (e - hr * (BSW * 32)) >> 5;
      return (pixel < a.W) & (halo_y0 + hr < a.H);
    };
    auto fetch_halo = [&]() {
#pragma unroll
      for (int k = 0; k < 12; ++k) hreg[k] = bc_recv(rrec, rec_used(k) ? slot + (unsigned)(lane256 + 256 * k) * 8u : BC_OOB);
    };
    auto halo_ok = [&]() -> bool {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < 12; ++k) ok = ok && (!rec_used(k) || hreg[k].y == want);
      return ok;
    };
    BC_STAMP();                                         // 0: block start
    fetch_halo();                                       // on every path: without a neighbour the own slot is read and ignored

    // (2) convA on the two T rows of this wave that need no halo row (RH 0: T rows 2,3 from input rows 2..5; RH 1: T rows 4,5
    //     from input rows 4..7), then on the two that do (RH 0: T rows 0,1 from input rows 0..3; RH 1: T rows 6,7 from 6..9)
    f32x4 accL[2][3], accD[2][3];
    const f32x4 b14 = (f32x4){bias1.x, bias1.y, bias1.z, bias1.w};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) accL[r][c] = b14;
    {
      unsigned off[8][2];
      sweep_bases(off, 0u, (RH == 0) ? 2 : 4, px, g);
      block_sweep<2>(accL, F, lds, off);
    }
    BC_STAMP();                                         // 1: first half of sweep 1 done
    // ReLU-mask vectors of epilogue 1 (data-gradient chains): requested now, they land under the halo step and the second half
    auto t_offset = [&](int k) -> unsigned {          // byte offset of pair k's vector in a [N,H,W,64] tensor, BC_OOB outside the image
      const int jr = (k < 4) ? k : (2 * (k - 4) + (g & 1)), c = (k < 4) ? (g & 1) : 2;
      const int y = sy * BSH - 1 + 4 * RH + jr, xx = 16 * c + px;
      const bool in = ((unsigned)y < (unsigned)a.H) & (xx < a.W);
      return in ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) * 2u : BC_OOB;
    };
    uint4 M[MASK ? 6 : 1];
    if (MASK) {
#pragma unroll
      for (int k = 0; k < 6; ++k) M[k] = bc_load16(rmask, t_offset(k));
    }
    if (b > 0) {
      if (recv) {
        unsigned spins = 0;
        while (!__all(halo_ok())) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > BCSPIN) { if (lane == 0) atomicExch(a.status, 0x200u + (unsigned)b); break; }
          fetch_halo();
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const int e = lane256 + 256 * k;
          const int hr = e / (BSW * 32), rem = e - hr * (BSW * 32), pixel = rem >> 5, cp = rem & 31;
          *reinterpret_cast<unsigned*>(ldx + swz((halo_xrow0 + hr) * BCOLS + pixel + 1, cp >> 2) + (cp & 3) * 4) = rec_used(k) ? hreg[k].x : 0u;
        }
      }
      __syncthreads();
    }
    BC_STAMP();                                         // 2: halo rows in LDS + barrier
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) accD[r][c] = b14;
    {
      unsigned off[8][2];
      sweep_bases(off, 0u, (RH == 0) ? 0 : 6, px, g);
      block_sweep<2>(accD, F, lds, off);
    }
    BC_STAMP();                                         // 3: second half of sweep 1 done
    // second filter + its bias: L2 hits that land under the epilogue
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(bc_load16(rw2, (unsigned)(q * 18 * 1024 + lane * 16 + t * 1024)));
    const uint4 bias2u = bc_load16(rb2, (unsigned)c0 * 4u);

    // (3) epilogue 1: T rows j = 4RH + jr -> HBM (the strip's own rows) + LDS.  Pairs: k < 4: X = (row k, col 0), Y = (row k, col 1);
    //     k = 4: X = (0, 2), Y = (1, 2); k = 5: X = (2, 2), Y = (3, 2); wave row jr is accD[jr] / accL[jr-2] for RH 0, accL[jr] / accD[jr-2] for RH 1
    {
      auto post1 = [&](f32x4 t) -> f32x4 {
        if (relu1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = fmaxf(t[j], 0.f);
        }
        if (scale1 != 1.0f) {
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] *= scale1;
        }
        return t;
      };
      auto tile = [&](auto JR, auto CC) -> f32x4 {
        constexpr int jr = decltype(JR)::value, c = decltype(CC)::value;
        if (RH == 0) return (jr < 2) ? accD[jr < 2 ? jr : 0][c] : accL[jr < 2 ? 0 : jr - 2][c];
        return (jr < 2) ? accL[jr < 2 ? jr : 0][c] : accD[jr < 2 ? 0 : jr - 2][c];
      };
      auto do_pair = [&](auto KC) {
        constexpr int k = decltype(KC)::value;
        constexpr int xr = (k < 4) ? k : 2 * (k - 4), yr = (k < 4) ? k : 2 * (k - 4) + 1;
        constexpr int xc = (k < 4) ? 0 : 2, yc = (k < 4) ? 1 : 2;
        const f32x4 tx = post1(tile(std::integral_constant<int, xr>{}, std::integral_constant<int, xc>{}));
        const f32x4 ty = post1(tile(std::integral_constant<int, yr>{}, std::integral_constant<int, yc>{}));
        float v[8];
        pair_up(tx, ty, g, v);
        if (MASK) {
          float m[8];
          unpack8(M[MASK ? k : 0], m);
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) v[jj] = (m[jj] > 0.f) ? v[jj] : 0.f;
        }
        const int jr = (k < 4) ? k : (2 * (k - 4) + (g & 1)), c = (k < 4) ? (g & 1) : 2;
        const int j = 4 * RH + jr, xx = 16 * c + px;
        const unsigned toff = t_offset(k);
        const bool in = toff != BC_OOB;
        const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
        const uint4 o = in ? make_uint4(lo.x, lo.y, hi.x, hi.y) : make_uint4(0, 0, 0, 0);     // outside the image: convB's zero padding
        bc_store16(rt, (in && j >= 1 && j <= BSH) ? toff : BC_OOB, o);                           // the strip's own rows only
        *reinterpret_cast<uint4*>(ldt + swz(j * BCOLS + xx + 1, chunk8)) = o;
      };
      do_pair(std::integral_constant<int, 0>{}); do_pair(std::integral_constant<int, 1>{}); do_pair(std::integral_constant<int, 2>{});
      do_pair(std::integral_constant<int, 3>{}); do_pair(std::integral_constant<int, 4>{}); do_pair(std::integral_constant<int, 5>{});
    }
    BC_STAMP();                                         // 4: epilogue 1 done
    __syncthreads();
    BC_STAMP();                                         // 5: barrier

    // (4) convB: output rows 3RH .. 3RH+2 of the strip from T rows r .. r+2
    f32x4 acc[3][3];
    {
      const f32x4 b4 = (f32x4){__uint_as_float(bias2u.x), __uint_as_float(bias2u.y), __uint_as_float(bias2u.z), __uint_as_float(bias2u.w)};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] = b4;
      unsigned off[8][2];
      sweep_bases(off, (unsigned)BXBYTES, 3 * RH, px, g);
      block_sweep<3>(acc, F, lds, off);
    }
    BC_STAMP();                                         // 6: sweep 2 done
    // the next block's first filter + bias: they land under the epilogue and the halo wait
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(bc_load16(rw1n, (unsigned)(q * 18 * 1024 + lane * 16 + t * 1024)));
    {
      const uint4 t = bc_load16(rb1n, (unsigned)c0 * 4u);
      bias1 = (bc_f32x4){__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
    }
    // (5) epilogue 2: OUT = X + scale2 * (convB(T) + b2) -> HBM, in place into the LDS input image, edge rows as tagged records.
    //     pairs k < 3: X = (row k, col 0), Y = (row k, col 1); k = 3: X = (0, 2), Y = (1, 2); single: (2, 2).
    //     edge rows of the group: RH 0 -> strip rows 0, 1 (wave rows 0, 1); RH 1 -> strip rows 4, 5 (wave rows 1, 2)
    {
      const unsigned myrec = my_slot + (unsigned)(b & 1) * (BCREC * 8u);
      const unsigned tag = epoch0 + (unsigned)b + 1u;
      const bool send = more && has_nb;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 tx = (k < 3) ? acc[k < 3 ? k : 0][0] : acc[0][2];
        const f32x4 ty = (k < 3) ? acc[k < 3 ? k : 0][1] : acc[1][2];
        float v[8], m[8];
        pair_up(tx, ty, g, v);
        const int r = (k < 3) ? k : (g & 1), c = (k < 3) ? (g & 1) : 2;
        const int srow = 3 * RH + r, y = sy * BSH + srow, xx = 16 * c + px;
        const bool in = (y < a.H) & (xx < a.W);
        unsigned char* xp = ldx + swz((srow + 2) * BCOLS + xx + 1, chunk8);
        unpack8(*reinterpret_cast<const uint4*>(xp), m);          // residual = the input image
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], scale2, m[j]);
        const unsigned e = (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) * 2u;
        const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
        const uint4 o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        const int er = (RH == 0) ? r : r - 1;                     // row inside the edge (valid when 0 or 1)
        {
          const bool sd = in & send & (er >= 0) & (er < 2);
          const unsigned rec = sd ? myrec + (unsigned)(er * (BSW * 32) + xx * 32 + 8 * q + (gpair >> 1)) * 8u : BC_OOB;
          bc_send(rrec, rec, o.x, tag); bc_send(rrec, rec + 8u, o.y, tag);
          bc_send(rrec, rec + 16u, o.z, tag); bc_send(rrec, rec + 24u, o.w, tag);
        }
        bc_store16(rout, in ? e : BC_OOB, o);
        if (more && in) *reinterpret_cast<uint4*>(xp) = o;       // the next block's input, in place
      }
      {
        const int srow = 3 * RH + 2, y = sy * BSH + srow, xx = 32 + px;
        const bool in = (y < a.H) & (xx < a.W);
        float v[4] = {acc[2][2][0], acc[2][2][1], acc[2][2][2], acc[2][2][3]};
        float m[4];
        unsigned char* xp = ldx + swz((srow + 2) * BCOLS + xx + 1, 2 * q + (g >> 1)) + (g & 1) * 8;
        unpack4_bf16(*reinterpret_cast<const uint2*>(xp), m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], scale2, m[j]);
        const unsigned e = (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0) * 2u;
        const uint2 o = pack4_bf16(v[0], v[1], v[2], v[3]);
        {
          const bool sd = in & send & (RH == 1);                   // wave row 2 of the lower group = strip row 5 = edge row 1
          const unsigned rec = sd ? myrec + (unsigned)(1 * (BSW * 32) + xx * 32 + 8 * q + 2 * g) * 8u : BC_OOB;
          bc_send(rrec, rec, o.x, tag); bc_send(rrec, rec + 8u, o.y, tag);
        }
        bc_store8(rout, in ? e : BC_OOB, o);
        if (more && in) *reinterpret_cast<uint2*>(xp) = o;
      }
    }
    BC_STAMP();                                         // 7: epilogue 2 done
    if (more) __syncthreads();      // the next block's input is complete in LDS
    BC_STAMP();                                         // 8: barrier
  }
  if (BCHAIN_ABL == 9 && lane == 0) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.xchg) + (a.xchg_bytes >> 3) + ((size_t)strip * 8 + q + 4 * RH) * 16;
    for (int i = 0; i < 10; ++i) dbg[i] = i < nst ? stamps[i] : 0ull;
  }
#undef BC_STAMP
}

template <bool MASK>
__global__ void __launch_bounds__(BTHREADS, 2) conv_block_chain_kernel(BChainDev a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[BXBYTES + BTBYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = wave >> 2;
  const int strip = blockIdx.x;
  const int n = strip / a.sy_n, sy = strip - n * a.sy_n;
  // ---- prologue: block 0's input rows 6sy-2 .. 6sy+7, columns -1 .. 48 (zero outside the image), T border columns ----
  {
    const bc_rsrc rx = bc_buffer(bc_uniform(a.blocks[0].x), (unsigned)(a.N * a.H * a.W) * 128u);
    uint4 R[BREGS];
    const int y0 = sy * BSH - 2;
#pragma unroll
    for (int i = 0; i < BREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      const int lr = pix / BCOLS, lc = pix - lr * BCOLS;
      const int y = y0 + lr, xc = lc - 1;
      const bool ok = (p < BPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)xc < (unsigned)a.W);
      R[i] = bc_load16(rx, ok ? (unsigned)(((n * a.H + y) * a.W + xc) * 64 + part * 8) * 2u : BC_OOB);    // out of range reads zeros
    }
    if (tid < BTROWS * 2 * 8) {
      const int row = tid >> 4, side = (tid >> 3) & 1, chunk = tid & 7;
      *reinterpret_cast<uint4*>(lds + BXBYTES + swz(row * BCOLS + side * (BCOLS - 1), chunk)) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      if (p < BPIECES) *reinterpret_cast<uint4*>(lds + swz(pix, part)) = R[i];
    }
  }
  __syncthreads();
  if (rh == 0) block_chain_body<0, MASK>(a, lds, tid, lane, q, px, g, strip, n, sy);
  else block_chain_body<1, MASK>(a, lds, tid, lane, q, px, g, strip, n, sy);
}

__global__ void block_chain_epoch_kernel(unsigned* hdr, unsigned inc) {
  unsigned v = hdr[0] + inc;
  if (v > 0xfff00000u) v = 1u;     // wrap long before 2^32 (records that old have been overwritten thousands of times)
  hdr[0] = v;
}

extern "C" int64_t rumpy_block_chain_xchg_bytes(int32_t nstrips) {
  return nstrips <= 0 ? 0 : 8ll * (BCHDR + (int64_t)nstrips * 4 * BCREC);
}

extern "C" int rumpy_block_chain(const rumpy_block_chain_args* p, void* stream) {
  if (!p || !p->blocks || !p->xchg || !p->status || p->nblocks <= 0) { rumpy_set_error("rumpy_block_chain: bad argument"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->W > BSW) { rumpy_set_error("rumpy_block_chain: needs 0 < W <= 48 (got %d)", p->W); return RUMPY_E_ARG; }
  const int sy_n = (p->H + BSH - 1) / BSH;
  const int nstrips = p->N * sy_n;
  if (nstrips > rumpy_device_cus()) { rumpy_set_error("rumpy_block_chain: %d strips do not fit %d CUs (all must be co-resident)", nstrips, rumpy_device_cus()); return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(p->status, 0, sizeof(unsigned), s) != hipSuccess) { rumpy_set_error("rumpy_block_chain: hipMemsetAsync failed"); return RUMPY_E_LAUNCH; }
  BChainDev d;
  d.blocks = p->blocks; d.nblocks = p->nblocks; d.N = p->N; d.H = p->H; d.W = p->W; d.sy_n = sy_n;
  d.xchg = (unsigned long long*)p->xchg; d.xchg_bytes = (unsigned)rumpy_block_chain_xchg_bytes(nstrips); d.status = p->status;
  rumpy_probe_pre(5, s);
  if (p->masked) hipLaunchKernelGGL(conv_block_chain_kernel<true>, dim3(nstrips), dim3(BTHREADS), 0, s, d);
  else hipLaunchKernelGGL(conv_block_chain_kernel<false>, dim3(nstrips), dim3(BTHREADS), 0, s, d);
  rumpy_probe_post(5, s);
  // records of this launch carry tags base+1 .. base+nblocks-1: move the base past them for the next launch / replay
  hipLaunchKernelGGL(block_chain_epoch_kernel, dim3(1), dim3(1), 0, s, (unsigned*)p->xchg, (unsigned)p->nblocks + 1u);
  return rumpy_check_launch("rumpy_block_chain");
}

// Shared by conv_rcab.hip and conv_rcab_fp8.hip: the device-side argument block of the one-launch RCAB kernels and the all-gather of per-strip
// channel sums among the strips of an image (8-byte records {fp32 value, tag} through HBM: the data is the signal).
#pragma once
#include "block_common.hpp"

typedef unsigned int rc_u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rc_rsrc;
constexpr int RC_SC1 = 16;
constexpr unsigned RC_SPIN = 1u << 20;
constexpr int RC_MAXR = 16;

struct RcabDev {
  const uint16_t* x; const uint4* w1; const float* b1; const uint4* w2; const float* b2;      // (fp8 launches: w1 / w2 = the fp8 filter images)
  uint16_t* t; uint16_t* t2; const uint16_t* t2_in; const uint16_t* mask; const uint16_t* res2; uint16_t* out;
  int N, H, W, sy_n;
  int ct_n, ns;              // column tiles per strip row, strips (workgroups) per image = sy_n * ct_n
  const float* cw1; const float* cb1; const float* cw2; const float* cb2; int cr; float inv_hw;
  float* mean; float* hidden; float* gate; const float* qgate; float* dz; float* dzq;
  unsigned long long* xchg; unsigned xchg_bytes; const unsigned* epoch; unsigned seq; unsigned* status;
  unsigned char* mbits;      // ReLU mask of t1 as one byte per 8 channels (block_common.hpp::relu_bits): forward writes, backward reads
  const unsigned* sw1; const unsigned* sw2; unsigned* site;      // precision fp8 (conv_rcab_fp8.hip): filter scale exponents, the launch's site record
};

__device__ __forceinline__ float wave_sum(float t) { return wave64_sum(t, (int)(threadIdx.x & 63)); }   // common.hpp: on the VALU, same bits as six __shfl_xor stages

// all-gather of one fp32 per (strip, channel) among the strips of image n; returns (threads < 64: channel tid) the sum over strips
// in strip order.  sx: LDS scratch of 8 * 64 floats.  Called by all 512 threads.
__device__ __forceinline__ float strip_allsum(const RcabDev& a, float mine, int n, int si, int tid, unsigned tag, float* sx) {
  const rc_rsrc rr = __builtin_amdgcn_make_buffer_rsrc((void*)a.xchg, 0, a.xchg_bytes, 0x00020000);
  const int c = tid & 63, w = tid >> 6;
  if (tid < 64) __builtin_amdgcn_raw_buffer_store_b64((rc_u32x2){__float_as_uint(mine), tag}, rr, (unsigned)(((n * a.ns + si) * 64 + c) * 8), 0, RC_SC1);
  float total = 0.f;
  for (int s0 = 0; s0 < a.ns; s0 += 8) {
    const int s = s0 + w;
    float val = 0.f;
    if (s < a.ns) {                                         // wave-uniform
      const unsigned byte = (unsigned)(((n * a.ns + s) * 64 + c) * 8);
      rc_u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rr, byte, 0, RC_SC1);
      unsigned spins = 0;
      while (!__all(r.y == tag)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > RC_SPIN) { if (c == 0) atomicExch(a.status, 0x300u + a.seq); break; }
        r = __builtin_amdgcn_raw_buffer_load_b64(rr, byte, 0, RC_SC1);
      }
      val = __uint_as_float(r.x);
    }
    __syncthreads();                                        // the previous round's sx has been consumed
    sx[w * 64 + c] = val;
    __syncthreads();
    if (tid < 64) {
#pragma unroll
      for (int k = 0; k < 8; ++k) total += sx[k * 64 + c];  // strips beyond ns contributed zeros
    }
  }
  return total;
}


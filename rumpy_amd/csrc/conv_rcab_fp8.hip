// precision = 'fp8' for the network BASELINE config 5 names (blind QRCAN / RCAN): the one-launch residual channel-attention block of
// conv_rcab.hip with both 3x3 sweeps on the block-scaled fp8 MFMA (conv_block_fp8.hip / fp8_common.hpp say what is fp8 and what is not, and
// how the scales are managed).  W <= 48 (one 6-row strip across the image per workgroup), bf16 tensors in HBM, ReLU mask as bytes.
//
//   forward :  t1 = relu(conv1(x) + b1) ; t2 = conv2(t1) + b2 ; gate = CA(mean_hw(t2)) [* gate_q] ; out = x + gate * t2
//              fp8 images: x (e4m3, scale exponent site[0]) and t1 (e4m3, site[1]); t2, the pool, the squeeze-excite MLP and the gate are fp32.
//   backward:  ds = sum_hw(dy * t2) -> (dz, dh, dp) through the MLP ; d_t2 = dy * gate + dp / HW ; gt1 = [t1 > 0] . conv2^T(d_t2) ;
//              dx = dy + conv1^T(gt1)
//              fp8 images: d_t2 (e5m2, site[0]) and gt1 (e5m2, site[1]); ds, the MLP backward and d_t2 itself are fp32 / bf16 as before.
//
// LDS (145 KB): a bf16 image of the strip's own 6 x 48 pixels (forward: x = the residual operand, overwritten by OUT; backward: dy = the operand
// of the product sums and of dx, overwritten by dx), the two fp8 images (10 x 50 and 8 x 50 pixels, 64 bytes per pixel), a second bf16 6 x 48
// image (t1 / gt1 on their way to HBM; forward: then t2; backward, first: scratch of the product sums), the exchange / MLP vectors.
// The exchange among the strips of an image, its tags and its watchdog are conv_rcab.hip's (rcab_common.hpp).
#include "fp8_common.hpp"
#include "rcab_common.hpp"

constexpr int R8_PIECES = BSH * BSW * 8;                                // 2304 16-byte pieces of a strip's own pixels
constexpr int R8_REGS = (R8_PIECES + BTHREADS - 1) / BTHREADS;          // 5

template <bool BWD>
__global__ void __launch_bounds__(BTHREADS, 2) rcab_fp8_kernel(RcabDev a) {
  constexpr bool E5M2 = BWD;
  f8_saturating_mode();                   // fp8 conversions clamp what outgrew its scale (fp8_common.hpp)
  __shared__ __attribute__((aligned(16))) unsigned char lds[F8_LDS];
  __shared__ float sx[8 * 64];
  __shared__ float spool[2 * 64];
  __shared__ __attribute__((aligned(16))) float sgate[64];
  __shared__ __attribute__((aligned(16))) float sdp[64];
  __shared__ float sw1[RC_MAXR * 64];      // [r][c] = conv_du.0.weight
  __shared__ float sw2t[RC_MAXR * 64];     // [r][c] = conv_du.2.weight[c][r]
  __shared__ unsigned gate[4];
  __shared__ unsigned amax_s[2];
  __shared__ float svec[4 * 64];           // [0] conv_du.0.bias (cr) | [1] conv_du.2.bias | [2] q gate | [3] bwd: forward gate ; hidden at [0][32..]
  __shared__ __attribute__((aligned(16))) unsigned char ldummy[64 * 16];
  unsigned char* const lc16 = lds;
  unsigned char* const lx8 = lds + F8_OFF_X8;
  unsigned char* const lt8 = lds + F8_OFF_T8;
  unsigned char* const lt16 = lds + F8_OFF_T16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = __builtin_amdgcn_readfirstlane(wave >> 2), tg = tid & 255;
  if (tid < 4) gate[tid] = 0u;
  if (tid < 2) amax_s[tid] = 0u;
  const int nwg = gridDim.x;
  const bool remap = ((nwg & 7) == 0) && (((nwg >> 3) % a.ns) == 0) && (a.ns <= 32);
  const int strip = remap ? xcd_strip(blockIdx.x, nwg) : (int)blockIdx.x;
  const int n = strip / a.ns, si = strip - n * a.ns, sy = si;
  const unsigned tag = (*a.epoch << 12) + a.seq;
  const int sbx = f8_exp(a.site[0]), sbt = f8_exp(a.site[1]);
  const int sa1 = f8_exp(*a.sw1), sa2 = f8_exp(*a.sw2);
  const float x_scale = __uint_as_float((unsigned)sbx << 23), t_scale = __uint_as_float((unsigned)sbt << 23);
  const bool t_keep = a.t != nullptr;
  float am_x = 0.f, am_t = 0.f;

  // ---- phase 0: tile rows 6sy-2 .. 6sy+7, columns -1 .. 48 (branch-free loads, zero outside the image) ----
  uint4 R[BREGS];                          // forward: consumed at once; backward: dy, kept until the exchange has produced d_t2's operands
  uint4 T2[BWD ? R8_REGS : 1];
  float mw1[2], mw2[2], mv = 0.f;
  {
    const int mtot = a.cr * 64;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + BTHREADS * k;
      mw1[k] = a.cw1[i < mtot ? i : 0];
      mw2[k] = a.cw2[i < mtot ? i : 0];
    }
    const int which = tid >> 6, c = tid & 63;
    const int cr_c = c < a.cr ? c : 0;
    const float* src = a.cb2; int idx = c;
    if (which == 0) { src = a.cb1; idx = cr_c; }
    else if (which == 2 && a.qgate) { src = a.qgate; idx = n * 64 + c; }
    else if (which == 3) { src = a.gate; idx = n * 64 + c; }
    else if (which == 4) { src = a.hidden; idx = n * a.cr + cr_c; }
    mv = src[idx];
    if (which == 2 && !a.qgate) mv = 1.f;
  }
  {
    unsigned am_xb = 0u;
    const int y0 = sy * BSH - 2;
#pragma unroll
    for (int i = 0; i < BREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      const int lr = pix / BCOLS, lc = pix - lr * BCOLS;
      const int y = y0 + lr, x = lc - 1;
      const bool ok = (p < BPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
      const int e = ok ? ((n * a.H + y) * a.W + x) * 64 + part * 8 : 0;
      uint4 v = *reinterpret_cast<const uint4*>(a.x + (unsigned)e);
      v = keep_if(v, ok);
      R[i] = v;
    }
    if (BWD) {   // the strip's own rows of the forward conv2 output
#pragma unroll
      for (int i = 0; i < R8_REGS; ++i) {
        const int p = tid + BTHREADS * i;
        const int pix = p >> 3, r = pix / BSW, col = pix - r * BSW;
        const int y = sy * BSH + r;
        const bool ok = (p < R8_PIECES) & (y < a.H) & (col < a.W);
        const int e = ok ? ((n * a.H + y) * a.W + col) * 64 + (p & 7) * 8 : 0;
        uint4 v = *reinterpret_cast<const uint4*>(a.t2_in + (unsigned)e);
        v = keep_if(v, ok);
        T2[BWD ? i : 0] = v;
      }
    }
    if (tid < BTROWS * 2 * 4) {           // border columns of the T image: the second conv's zero padding
      const int row = tid >> 3, side = (tid >> 2) & 1, quarter = tid & 3;
      *reinterpret_cast<uint4*>(lt8 + f8_swz(row * BCOLS + side * (BCOLS - 1), quarter)) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      if (p < BPIECES) {
        const int lr = pix / BCOLS, lc = pix - lr * BCOLS;
        if (lr >= 2 && lr < 2 + BSH && lc >= 1 && lc <= BSW) *reinterpret_cast<uint4*>(lc16 + swz((lr - 2) * BSW + lc - 1, part)) = R[i];
        if (!BWD) {                        // forward: the tile is the first sweep's operand as it is
          am_xb = f8_amax_bf16(am_xb, R[i]);
          *reinterpret_cast<uint2*>(lx8 + f8_swz(pix, part >> 1) + (part & 1) * 8) = f8_pack8_bf16<E5M2>(R[i], x_scale);
        }
      }
    }
    if (!BWD) am_x = f8_amax_bf16_value(am_xb);
    {
      const int mtot = a.cr * 64;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = tid + BTHREADS * k;
        if (i < mtot) {
          sw1[i] = mw1[k];
          sw2t[(i % a.cr) * 64 + i / a.cr] = mw2[k];
        }
      }
      const int which = tid >> 6, c = tid & 63;
      if (which == 0) { if (c < RC_MAXR) svec[c] = mv; }
      else if (which < 4) svec[which * 64 + c] = mv;
      else if (which == 4 && c < RC_MAXR) svec[32 + c] = mv;
    }
  }
  f8_v8i A[5];
  {
    const f8_v8i* wp = reinterpret_cast<const f8_v8i*>(a.w1) + (size_t)q * 5 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 5; ++t) A[t] = wp[t * 64];
  }
  const int c0 = 16 * q + 4 * g;
  const int gpair = 4 * (g & ~1);
  const int chunk8 = 2 * q + (gpair >> 3);
  __syncthreads();

  if (BWD) {
    // ---- phase 0b: ds = sum over the strip of dy * t2 per channel -> all strips of the image -> MLP backward -> d_t2 ----
    float part8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < R8_REGS; ++i) {
      const int p = tid + BTHREADS * i;
      if (p < R8_PIECES) {
        float d[8], t[8];
        unpack8(*reinterpret_cast<const uint4*>(lc16 + swz(p >> 3, tid & 7)), d);
        unpack8(T2[BWD ? i : 0], t);
#pragma unroll
        for (int j = 0; j < 8; ++j) part8[j] = fmaf(d[j], t[j], part8[j]);
      }
    }
    // threads with the same chunk (tid & 7) hold partial sums of the same 8 channels: [k = tid >> 3][chunk][8] in the (still unused) second
    // bf16 image, then 256 threads add 16 k's each, then 64 threads add the 4 parts - fixed order (conv_rcab.hip)
    float* red = reinterpret_cast<float*>(lt16);
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
    const float ds = strip_allsum(a, mine, n, si, tid, tag, sx);
    if (tid < 64) {
      const int c = tid;
      const float s = svec[3 * 64 + c];
      const float gq = svec[2 * 64 + c];
      const float dz = (ds * gq) * s * (1.f - s);
      float dp = 0.f;
      for (int r0 = 0; r0 < a.cr; r0 += 4) {
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
    // d_t2 = dy * gate + dp / HW on every pixel of the tile inside the image (outside: the zero padding) -> the e5m2 image the first sweep
    // reads; the strip's own pixels also go to HBM as bf16 (conv2's weight gradient reads them): 8 lanes per pixel, whole lines
    {
      const int y0 = sy * BSH - 2;
      const float4 ga = *reinterpret_cast<const float4*>(sgate + (tid & 7) * 8), gb = *reinterpret_cast<const float4*>(sgate + (tid & 7) * 8 + 4);
      const float4 pa = *reinterpret_cast<const float4*>(sdp + (tid & 7) * 8), pb = *reinterpret_cast<const float4*>(sdp + (tid & 7) * 8 + 4);
#pragma unroll
      for (int i = 0; i < BREGS; ++i) {
        const int p = tid + BTHREADS * i;
        const int pix = p >> 3, part = p & 7;
        const int lr = pix / BCOLS, lc = pix - lr * BCOLS;
        const int y = y0 + lr, x = lc - 1;
        const bool ok = (p < BPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
        float d[8];
        unpack8(R[i], d);
        float o[8] = {fmaf(d[0], ga.x, pa.x), fmaf(d[1], ga.y, pa.y), fmaf(d[2], ga.z, pa.z), fmaf(d[3], ga.w, pa.w),
                      fmaf(d[4], gb.x, pb.x), fmaf(d[5], gb.y, pb.y), fmaf(d[6], gb.z, pb.z), fmaf(d[7], gb.w, pb.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = ok ? o[j] : 0.f; am_x = fmaxf(am_x, fabsf(o[j])); }
        asm volatile("" : "+v"(am_x));      // taken HERE: left alone, the compiler sinks the whole max chain to its use behind the first sweep and spills the 64 values it reads
        if (p < BPIECES) *reinterpret_cast<uint2*>(lx8 + f8_swz(pix, part >> 1) + (part & 1) * 8) = f8_pack8<E5M2>(o, x_scale);
        if (ok && lr >= 2 && lr < 2 + BSH && lc >= 1 && lc <= BSW) {
          const uint2 lo = pack4_bf16(o[0], o[1], o[2], o[3]), hi = pack4_bf16(o[4], o[5], o[6], o[7]);
          st16_nt(a.t2 + (unsigned)(((n * a.H + y) * a.W + x) * 64 + part * 8), make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
      }
    }
    __syncthreads();
  }

  // ---- phase 1: T rows j = 4rh .. 4rh+3 (image rows 6sy-1+j) ----
  {
    f32x4 acc[4][3];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!BWD) { const float4 t = *reinterpret_cast<const float4*>(a.b1 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    unsigned MB[BWD ? 6 : 1];
    if (BWD) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int jr = (k < 4) ? k : (2 * (k - 4) + (g & 1)), c = (k < 4) ? (g & 1) : 2;
        const int y = sy * BSH - 1 + 4 * rh + jr, xx = 16 * c + px;
        const bool in = ((unsigned)y < (unsigned)a.H) & (xx < a.W);
        MB[BWD ? k : 0] = a.mbits[in ? (unsigned)(((n * a.H + y) * a.W + xx) * 8 + chunk8) : 0u];
      }
    }
    unsigned fb[8];
    f8_bases(fb, (unsigned)F8_OFF_X8, 4 * rh, px, g);
    f8_sweep<4, E5M2>(acc, A, lds, fb, sa1, sbx);
    {
      const f8_v8i* wp = reinterpret_cast<const f8_v8i*>(a.w2) + (size_t)q * 5 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 5; ++t) A[t] = wp[t * 64];
    }
    // (lane constants of the epilogue tied behind the sweep, selects instead of branches: conv_block_fp8.hip says why)
    int pxe = px, ge = g;
    asm volatile("" : "+v"(pxe), "+v"(ge) : "v"(acc[3][2]));
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const f32x4 tx = (k < 4) ? acc[k < 4 ? k : 0][0] : acc[2 * (k < 4 ? 0 : k - 4)][2];
      const f32x4 ty = (k < 4) ? acc[k < 4 ? k : 0][1] : acc[2 * (k < 4 ? 0 : k - 4) + 1][2];
      float v[8];
      pair_up(tx, ty, g, v);
      const int jr = (k < 4) ? k : (2 * (k - 4) + (ge & 1)), c = (k < 4) ? (ge & 1) : 2;
      const int j = 4 * rh + jr, xx = 16 * c + pxe;
      const int y = sy * BSH - 1 + j;
      const bool in = ((unsigned)y < (unsigned)a.H) & (xx < a.W);
      if (!BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = in ? relu_f32(v[e]) : 0.f;
      } else {
        const unsigned mb = in ? MB[BWD ? k : 0] : 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((mb >> e) & 1u) ? v[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) am_t = fmaxf(am_t, fabsf(v[e]));
      const uint2 o8 = f8_pack8<E5M2>(v, t_scale);
      const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
      const int gp = 4 * (ge & ~1);
      *reinterpret_cast<uint2*>(lt8 + f8_swz(j * BCOLS + xx + 1, q) + gp) = o8;
      const bool own = t_keep & (j >= 1) & (j <= BSH);
      *reinterpret_cast<uint4*>(own ? lt16 + swz((j - 1) * BSW + xx, 2 * q + (gp >> 3)) : ldummy + lane * 16) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    am_x = f8_wave_max(am_x, lane);
    am_t = f8_wave_max(am_t, lane);
    if (lane == 0) {
      __hip_atomic_fetch_max(&amax_s[0], __float_as_uint(am_x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_max(&amax_s[1], __float_as_uint(am_t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    gate_arrive(&gate[rh], lane);
  }
  gate_wait(&gate[rh], 4u);
  if (rh == 1) gate_wait(&gate[0], 4u);
  int px2 = px, g2 = g, tg2 = tg, tid2 = tid;
  asm volatile("" : "+v"(px2), "+v"(g2), "+v"(tg2), "+v"(tid2) :: "memory");
  unsigned soffg[GROUP_REGS];
#pragma unroll
  for (int i = 0; i < GROUP_REGS; ++i) soffg[i] = group_piece_off(i, tg2, rh, n, sy, a.H, a.W);
  // the row half's own strip rows of t1 / gt1 (forward: + the ReLU mask bytes) leave for HBM under the second sweep, one 16-byte piece per thread
  // after every other step (conv_block_fp8.hip), staged from the finished bf16 image first
  uint4 St[GROUP_REGS];
  const bool t_out = a.t != nullptr;
  if (t_out) f8_stage48(St, lt16, tg2, rh);
  auto t_store = [&](int step) {          // step is a constant after unrolling
    if (step % 2 == 0 && step / 2 < GROUP_REGS) {
      const int i = step / 2 < GROUP_REGS ? step / 2 : 0;
      if (t_out && soffg[i] != 0xffffffffu) {
        st16_nt(a.t + soffg[i], St[i]);
        if (!BWD && a.mbits) a.mbits[soffg[i] >> 3] = (unsigned char)relu_bits(St[i]);
      }
    }
  };

  // ---- phase 2: rows 3rh .. 3rh+2 of the strip from T rows r .. r+2 ----
  {
    f32x4 acc[3][3];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!BWD) { const float4 t = *reinterpret_cast<const float4*>(a.b2 + 16 * q + 4 * g2); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    unsigned fb[8];
    if (rh == 0) {
      f8_bases(fb, (unsigned)F8_OFF_T8, 0, px2, g2);
      f8_sweep<2, E5M2, decltype(t_store)>(*reinterpret_cast<f32x4(*)[2][3]>(&acc[0]), A, lds, fb, sa2, sbt, t_store);
      gate_wait(&gate[1], 4u);
      f8_bases(fb, (unsigned)F8_OFF_T8, 2, px2, g2);
      f8_sweep<1, E5M2>(*reinterpret_cast<f32x4(*)[1][3]>(&acc[2]), A, lds, fb, sa2, sbt);
    } else {
      f8_bases(fb, (unsigned)F8_OFF_T8, 3, px2, g2);
      f8_sweep<3, E5M2, decltype(t_store)>(acc, A, lds, fb, sa2, sbt, t_store);
    }
    int px3 = px2, g3 = g2;
    asm volatile("" : "+v"(px3), "+v"(g3) : "v"(acc[2][2]));
    const int gp3 = 4 * (g3 & ~1), ch3 = 2 * q + (gp3 >> 3), c03 = 16 * q + 4 * g3;
    // pairs k < 3: X = (row k, col 0), Y = (row k, col 1); k = 3: X = (0, 2), Y = (1, 2); single: (2, 2)
    float V[4][8], vs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 tx = (k < 3) ? acc[k < 3 ? k : 0][0] : acc[0][2];
      const f32x4 ty = (k < 3) ? acc[k < 3 ? k : 0][1] : acc[1][2];
      pair_up(tx, ty, g3, V[k]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) vs[j] = acc[2][2][j];
    bool inp[4];
    unsigned char* cell[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = (k < 3) ? k : (g3 & 1), c = (k < 3) ? (g3 & 1) : 2;
      const int srow = 3 * rh + r, xx = 16 * c + px3;
      inp[k] = (sy * BSH + srow < a.H) & (xx < a.W);
      cell[k] = lc16 + swz(srow * BSW + xx, ch3);
    }
    const int srow_s = 3 * rh + 2, xx_s = 32 + px3;
    const bool in_s = (sy * BSH + srow_s < a.H) & (xx_s < a.W);
    unsigned char* const cell_s = lc16 + swz(srow_s * BSW + xx_s, 2 * q + (g3 >> 1)) + (g3 & 1) * 8;

    if (BWD) {
      // dx = dy + conv1^T(gt1) [+ res2]: dy from the bf16 image of the strip, dx in its place (a pixel outside the image computes on zeros and is never stored)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float m[8];
        unpack8(*reinterpret_cast<const uint4*>(cell[k]), m);
        if (a.res2) {
          const int r = (k < 3) ? k : (g3 & 1), c = (k < 3) ? (g3 & 1) : 2;
          const int y = sy * BSH + 3 * rh + r, xx = 16 * c + px3;
          float e[8];
          unpack8(*reinterpret_cast<const uint4*>(a.res2 + (inp[k] ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gp3) : 0u)), e);
#pragma unroll
          for (int j = 0; j < 8; ++j) m[j] += e[j];
        }
        const uint2 lo = pack4_bf16(V[k][0] + m[0], V[k][1] + m[1], V[k][2] + m[2], V[k][3] + m[3]);
        const uint2 hi = pack4_bf16(V[k][4] + m[4], V[k][5] + m[5], V[k][6] + m[6], V[k][7] + m[7]);
        *reinterpret_cast<uint4*>(cell[k]) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
      {
        float m[4];
        unpack4_bf16(*reinterpret_cast<const uint2*>(cell_s), m);
        if (a.res2) {
          float e[4];
          unpack4_bf16(*reinterpret_cast<const uint2*>(a.res2 + (in_s ? (unsigned)(((n * a.H + sy * BSH + srow_s) * a.W + xx_s) * 64 + c03) : 0u)), e);
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] += e[j];
        }
        *reinterpret_cast<uint2*>(cell_s) = pack4_bf16(vs[0] + m[0], vs[1] + m[1], vs[2] + m[2], vs[3] + m[3]);
      }
    } else {
      // t2 = conv2(t1) + b2: channel sums of the strip for the attention pool, t2 itself to HBM when training
      float ps8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) ps8[j] += inp[k] ? V[k][j] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) ps[j] += in_s ? vs[j] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = row16_sum(ps8[j]);
        t += lane_xor16(t, g3);
        ps8[j] = t;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = row16_sum(ps[j]);
        const float up = lane_xor16(t, g3);
        ps8[j] += (g3 & 1) ? up : t;
        ps8[4 + j] += (g3 & 1) ? t : up;
      }
      if (px3 == 0 && !(g3 & 1)) {
        float* pp = spool + rh * 64 + 16 * q + 4 * g3;
        *reinterpret_cast<float4*>(pp) = make_float4(ps8[0], ps8[1], ps8[2], ps8[3]);
        *reinterpret_cast<float4*>(pp + 4) = make_float4(ps8[4], ps8[5], ps8[6], ps8[7]);
      }
      __syncthreads();                     // every wave has finished its second sweep and staged its t1 pieces: the second bf16 image is free
      unsigned soff[R8_REGS];
#pragma unroll
      for (int i = 0; i < R8_REGS; ++i) soff[i] = strip_piece_off(i, tid2, n, sy, a.H, a.W);
      if (a.t2) {                          // training: t2 goes to HBM through the second bf16 image (whole lines, below)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = (k < 3) ? k : (g3 & 1), c = (k < 3) ? (g3 & 1) : 2;
          const uint2 lo = pack4_bf16(V[k][0], V[k][1], V[k][2], V[k][3]), hi = pack4_bf16(V[k][4], V[k][5], V[k][6], V[k][7]);
          *reinterpret_cast<uint4*>(lt16 + swz((3 * rh + r) * BSW + 16 * c + px3, ch3)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        *reinterpret_cast<uint2*>(lt16 + swz(srow_s * BSW + xx_s, 2 * q + (g3 >> 1)) + (g3 & 1) * 8) = pack4_bf16(vs[0], vs[1], vs[2], vs[3]);
      }
      const float mine = (tid2 < 64) ? spool[tid2 & 63] + spool[64 + (tid2 & 63)] : 0.f;
      const float tot = strip_allsum(a, mine, n, si, tid, tag, sx);      // (its barriers also complete the t2 image)
      uint4 S[R8_REGS];
      if (a.t2) {
#pragma unroll
        for (int i = 0; i < R8_REGS; ++i) {
          const int p = tid2 + BTHREADS * i;
          S[i] = *reinterpret_cast<const uint4*>(lt16 + swz((p < R8_PIECES ? p : 0) >> 3, p & 7));
        }
      }
      if (tid < 64) {
        const int c = tid;
        const float mean = tot * a.inv_hw;
        float z = svec[64 + c];
        for (int r0 = 0; r0 < a.cr; r0 += 4) {
          float hs[4], w2[4], b1[4];
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
        for (int i = 0; i < R8_REGS; ++i)
          if (soff[i] != 0xffffffffu) st16_nt(a.t2 + soff[i], S[i]);
      }
      // out = x + gate * t2, the residual operand from the bf16 image of the strip; the result replaces it there
      const float4 ga = *reinterpret_cast<const float4*>(sgate + 16 * q + gp3), gb = *reinterpret_cast<const float4*>(sgate + 16 * q + gp3 + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float m[8];
        unpack8(*reinterpret_cast<const uint4*>(cell[k]), m);
        const uint2 lo = pack4_bf16(fmaf(V[k][0], ga.x, m[0]), fmaf(V[k][1], ga.y, m[1]), fmaf(V[k][2], ga.z, m[2]), fmaf(V[k][3], ga.w, m[3]));
        const uint2 hi = pack4_bf16(fmaf(V[k][4], gb.x, m[4]), fmaf(V[k][5], gb.y, m[5]), fmaf(V[k][6], gb.z, m[6]), fmaf(V[k][7], gb.w, m[7]));
        *reinterpret_cast<uint4*>(cell[k]) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
      {
        float m[4];
        unpack4_bf16(*reinterpret_cast<const uint2*>(cell_s), m);
        const float4 gs = *reinterpret_cast<const float4*>(sgate + c03);
        *reinterpret_cast<uint2*>(cell_s) = pack4_bf16(fmaf(vs[0], gs.x, m[0]), fmaf(vs[1], gs.y, m[1]), fmaf(vs[2], gs.z, m[2]), fmaf(vs[3], gs.w, m[3]));
      }
    }
  }
  // ---- OUT (forward: x + gate * t2; backward: dx) sits in the bf16 image of the strip -> whole lines, non-temporal ----
  if (BWD) {
    gate_arrive(&gate[2 + rh], lane);
    gate_wait(&gate[2 + rh], 4u);
    uint4 S[GROUP_REGS];
    f8_stage48(S, lc16, tg2, rh);
#pragma unroll
    for (int i = 0; i < GROUP_REGS; ++i)
      if (soffg[i] != 0xffffffffu) st16_nt(a.out + soffg[i], S[i]);
  } else {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R8_REGS; ++i) {
      const int p = tid2 + BTHREADS * i;
      const unsigned so = strip_piece_off(i, tid2, n, sy, a.H, a.W);
      const uint4 v = *reinterpret_cast<const uint4*>(lc16 + swz((p < R8_PIECES ? p : 0) >> 3, p & 7));
      if (so != 0xffffffffu) st16_nt(a.out + so, v);
    }
  }
  // amax of the two image tensors: one entry per (workgroup, row half) of the site record (conv_block_fp8.hip); every wave has added its share
  // before the barrier / gate above
  if (tg2 == 0) {
    unsigned* e = a.site + RUMPY_FP8_SITE_HEAD + 2 * (2 * blockIdx.x + rh);
    e[0] = __hip_atomic_load(&amax_s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    e[1] = __hip_atomic_load(&amax_s[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

int rumpy_rcab_fp8_launch(const rumpy_rcab_args* p, const RcabDev& d0, hipStream_t s, bool bwd, const char* what) {
  if (p->W > BSW || p->fmt != RUMPY_FMT_BF16 || (bwd && !p->maskbits)) {
    rumpy_set_error("%s: the fp8 images go with W <= %d, bf16 tensors and (backward) the ReLU mask as bytes", what, BSW); return RUMPY_E_ARG; }
  if (!p->w2_f8 || !p->f8_sw1 || !p->f8_sw2 || !p->f8_site) { rumpy_set_error("%s: fp8 launch needs w1_f8, w2_f8, f8_sw1, f8_sw2 and f8_site", what); return RUMPY_E_ARG; }
  RcabDev d = d0;
  d.w1 = (const uint4*)p->w1_f8; d.w2 = (const uint4*)p->w2_f8; d.sw1 = p->f8_sw1; d.sw2 = p->f8_sw2; d.site = p->f8_site;
  const dim3 grid(d.N * d.ns);
  if (p->f8_entries < 2 * (int)grid.x) { rumpy_set_error("%s: f8_entries %d < 2 * %u workgroups (rumpy_fp8_site_entries)", what, p->f8_entries, grid.x); return RUMPY_E_ARG; }
  if (bwd) RUMPY_LAUNCH_PROBED(5, (rcab_fp8_kernel<true>), grid, dim3(BTHREADS), s, d);
  else RUMPY_LAUNCH_PROBED(5, (rcab_fp8_kernel<false>), grid, dim3(BTHREADS), s, d);
  return 0;
}

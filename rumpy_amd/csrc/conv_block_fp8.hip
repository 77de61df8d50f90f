// precision = 'fp8' (BASELINE config 5 names "fp8 MFMA conv"): the residual block of conv_block.hip with BOTH 3x3 sweeps on the block-scaled
// fp8 matrix instruction of gfx950, v_mfma_scale_f32_16x16x128_f8f6f4 - an opt-in of its own accuracy class, never the default.
//
//   FORM 1, forward (training or inference):  T = relu(conv1(X) + b1)            OUT = X + s * (conv2(T) + b2)
//   FORM 3, data gradient:                    GT = maskbits . s * conv2^T(G)     GX  = G + conv1^T(GT) [+ res2]
//
// What is fp8 and what is not.  Only the two operands of the matrix pipe: the filter images (OCP e4m3, one power-of-two scale per conv, packed
// by rumpy_fp8_pack after every optimizer step) and the activation / gradient images the sweeps read from LDS (e4m3 forward, e5m2 backward; one
// power-of-two scale per tensor).  Products are exact, accumulation is fp32.  Bias, ReLU, res_scale, the ReLU mask, the residual operand, every
// tensor in HBM (X, T, OUT stay bf16 - the weight gradients and the neighbouring launches read them) and the optimizer are those of the bf16 path.
// The scales travel through the instruction's e8m0 scale operands, so no value is ever multiplied by a scale in software.
//
// Scale management = delayed scaling, device side only.  A launch reads the e8m0 exponents of its two image tensors from its `site` record and
// leaves the amax of both (as the values really were this step, fp32) in the record, one entry per (workgroup, row half), plain stores;
// rumpy_fp8_rotate, one small launch in front of every pass, turns last pass's amax into this pass's exponent such that amax / scale lies in
// [128, 256) - 1.75 x growth from one step to the next still fits e4m3 (448), far more fits e5m2; what outgrows even that is clamped to the
// largest finite value - and clears the entries.  The host never reads a scale.  The first pass of a plan runs twice (once to measure;
// rumpy_amd/engine.py).
//
// Geometry = conv_block.hip's W <= 48 geometry (one 6-row strip across the image per 512-thread workgroup, wave (q, rh) = output channels 16 q ..
// of one row half, row-half gates instead of workgroup barriers, whole-line non-temporal stores from LDS images).  What differs (measured first as
// tests/tools/fp8/conv_block_fp8.hip in round 3: 11.0 against 13.8 us per inference-form launch):
//  * LDS holds fp8 images of X (10 x 50 pixels) and T (8 x 50): 64 bytes per pixel, 16-byte chunk index XOR-ed with 2 * ((pixel >> 2) & 1)
//    (conflict-free ds_read_b128 of chunk g of 16 consecutive pixels); a bf16 image of the strip's own 6 x 48 pixels of X (residual operand; OUT
//    is written over it and leaves from there) and one of T (leaves for HBM from there; training).  131 KB.  X is converted while its tile is
//    staged, T in the first phase's epilogue (v_cvt_scalef32_pk_{fp8,bf8}_f32) - never inside a sweep.
//  * One MFMA has K = 128 = two taps x 64 channels; lane (pixel px, group g) supplies 32 bytes.  The 9 taps of an output tile are 5 MFMAs - three
//    "taps (ky, kx 0 | kx 1)" on row fragments shared by the output rows they feed, "taps (ky 0 | ky 1, kx 2)" and "(nothing | ky 2, kx 2)" on
//    windows of one register run of column-2 pieces (fp8_common.hpp::f8_sweep).  (tests/tools/fp8/fp8_probe.hip: any lane / byte -> k assignment
//    works as long as both operands use the same one.)  5 MFMAs of 32 cycles against 18 of 16: 0.56 of the matrix-pipe time, and half of the
//    LDS fragment bytes.
#include "fp8_common.hpp"

struct BlockF8Dev {
  const uint16_t* x; const f8_v8i* w1; const float* b1; const f8_v8i* w2; const float* b2;
  const uint16_t* res2; uint16_t* t; uint16_t* out; unsigned char* mbits;
  int N, H, W, sy_n; float scale1, scale2;
  const unsigned* sw1; const unsigned* sw2;        // e8m0 exponents of the two filter images (rumpy_fp8_pack)
  unsigned* site;                                  // record of this launch (rumpy_amd.h): [0] / [1] exponents of the X / T images, [2] entries, then the amax pairs
};

template <int FORM>
__global__ void __launch_bounds__(BTHREADS, 2) conv_block_fp8_kernel(BlockF8Dev a) {
  constexpr bool E5M2 = FORM == 3;
  f8_saturating_mode();                   // fp8 conversions clamp what outgrew its scale (fp8_common.hpp)
  __shared__ __attribute__((aligned(16))) unsigned char lds[F8_LDS];
  __shared__ unsigned gate[4];
  __shared__ unsigned amax_s[2];
  __shared__ __attribute__((aligned(16))) unsigned char ldummy[64 * 16];
  unsigned char* const lc16 = lds;
  unsigned char* const lx8 = lds + F8_OFF_X8;
  unsigned char* const lt8 = lds + F8_OFF_T8;
  unsigned char* const lt16 = lds + F8_OFF_T16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int q = wave & 3, rh = __builtin_amdgcn_readfirstlane(wave >> 2), tg = tid & 255;
  const int strip = xcd_strip(blockIdx.x, gridDim.x);
  const int n = strip / a.sy_n, sy = strip - n * a.sy_n;
  const int sbx = f8_exp(a.site[0]), sbt = f8_exp(a.site[1]);
  const int sa1 = f8_exp(*a.sw1), sa2 = f8_exp(*a.sw2);
  const float x_scale = __uint_as_float((unsigned)sbx << 23), t_scale = __uint_as_float((unsigned)sbt << 23);
  float am_x = 0.f, am_t = 0.f;
  const bool t_keep = a.t != nullptr;

  // ---- phase 0: input tile -> fp8 image (matrix operand); its centre 6 x 48 pixels also as they are (residual operand) ----
  {
    uint4 R[BREGS];
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
      if (!ok) v = make_uint4(0, 0, 0, 0);
      R[i] = v;
    }
    if (tid < 4) gate[tid] = 0u;
    if (tid < 2) amax_s[tid] = 0u;
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
        am_xb = f8_amax_bf16(am_xb, R[i]);
        *reinterpret_cast<uint2*>(lx8 + f8_swz(pix, part >> 1) + (part & 1) * 8) = f8_pack8_bf16<E5M2>(R[i], x_scale);
      }
    }
    am_x = f8_amax_bf16_value(am_xb);
  }
  f8_v8i A[5];
  {
    const f8_v8i* wp = a.w1 + (size_t)q * 5 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 5; ++t) A[t] = wp[t * 64];
  }
  const int c0 = 16 * q + 4 * g;
  const int gpair = 4 * (g & ~1);
  const int chunk8 = 2 * q + (gpair >> 3);
  __syncthreads();

  // ---- phase 1: T rows j = 4rh .. 4rh+3 (image rows 6sy-1+j) ----
  {
    f32x4 acc[4][3];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.b1) { const float4 t = *reinterpret_cast<const float4*>(a.b1 + c0); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    // FORM 3: the ReLU mask bytes of this lane's pixels (conv_block.hip: one byte per 8 channels), requested before the sweep
    unsigned MB[FORM == 3 ? 6 : 1];
    if (FORM == 3) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int jr = (k < 4) ? k : (2 * (k - 4) + (g & 1)), c = (k < 4) ? (g & 1) : 2;
        const int y = sy * BSH - 1 + 4 * rh + jr, xx = 16 * c + px;
        const bool in = ((unsigned)y < (unsigned)a.H) & (xx < a.W);
        MB[FORM == 3 ? k : 0] = a.mbits[in ? (unsigned)(((n * a.H + y) * a.W + xx) * 8 + chunk8) : 0u];
      }
    }
    unsigned fb[8];
    f8_bases(fb, (unsigned)F8_OFF_X8, 4 * rh, px, g);
    f8_sweep<4, E5M2>(acc, A, lds, fb, sa1, sbx);
    {
      const f8_v8i* wp = a.w2 + (size_t)q * 5 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 5; ++t) A[t] = wp[t * 64];
    }
    // the epilogue's addresses and predicates are lane constants: left to itself the compiler computes all of them above the sweep and carries
    // them through it (the sweep runs at 230 of 256 registers: 110 spilled registers, most of them inside the MFMA loop).  Tying the lane's
    // pixel index to the last accumulator keeps that arithmetic behind the sweep.
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
      // branch-free (selects only): with real branches here LLVM sinks the last column's MFMAs into the epilogue's blocks and their fragments
      // live - spilled - through all of it (440 bytes of scratch per lane in the first build)
      if (FORM == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = in ? relu_f32(v[e]) : 0.f;
      } else {
        const unsigned mb = in ? MB[FORM == 3 ? k : 0] : 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((mb >> e) & 1u) ? v[e] * a.scale1 : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) am_t = fmaxf(am_t, fabsf(v[e]));
      const uint2 o8 = f8_pack8<E5M2>(v, t_scale);            // outside the image: zeros = the second conv's padding
      const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
      const int gp = 4 * (ge & ~1);
      *reinterpret_cast<uint2*>(lt8 + f8_swz(j * BCOLS + xx + 1, q) + gp) = o8;        // this wave's 16 channels = quarter q; 8 of them per lane
      // the strip's own rows also as bf16 (what leaves for HBM); halo rows - and everything when T is not kept - go to a per-lane dummy slot
      const bool own = t_keep & (j >= 1) & (j <= BSH);
      *reinterpret_cast<uint4*>(own ? lt16 + swz((j - 1) * BSW + xx, 2 * q + (gp >> 3)) : ldummy + lane * 16) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    // the two amax values of this wave -> the workgroup's (read by the row halves' leaders at the end)
    am_x = f8_wave_max(am_x, lane);
    am_t = f8_wave_max(am_t, lane);
    if (lane == 0) {
      __hip_atomic_fetch_max(&amax_s[0], __float_as_uint(am_x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_max(&amax_s[1], __float_as_uint(am_t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    gate_arrive(&gate[rh], lane);          // this wave's 16 channels of T rows 4rh .. 4rh+3 are in LDS
  }
  gate_wait(&gate[rh], 4u);
  if (rh == 1) gate_wait(&gate[0], 4u);
  // (lane constants of the second phase - store offsets, fragment bases, staging addresses - are derived from copies of the lane ids made HERE:
  // computed from the kernel's first instructions on, as the compiler would, they ride through the first sweep and spill its fragments)
  int px2 = px, g2 = g, tg2 = tg;
  asm volatile("" : "+v"(px2), "+v"(g2), "+v"(tg2) :: "memory");
  unsigned soff[GROUP_REGS];
#pragma unroll
  for (int i = 0; i < GROUP_REGS; ++i) soff[i] = group_piece_off(i, tg2, rh, n, sy, a.H, a.W);
  // the row half's own strip rows of T (and, forward, their ReLU mask bytes) leave for HBM from the finished bf16 image: whole lines, non-temporal,
  // one piece after every other step of the second sweep
  // ... one 16-byte piece per thread after every other step of the second sweep (conv_block.hip), staged from the finished bf16 image first.
  // (With the first form of the sweep - 112 fragment registers - the five staged pieces did not fit and went to scratch; the stores then sat in front
  // of the sweep.  Under it they are worth 0.8 % of the step.)
  uint4 S[GROUP_REGS];
  const bool t_out = a.t != nullptr;
  if (t_out) f8_stage48(S, lt16, tg2, rh);
  auto t_store = [&](int step) {          // step is a constant after unrolling
    if (step % 2 == 0 && step / 2 < GROUP_REGS) {
      const int i = step / 2 < GROUP_REGS ? step / 2 : 0;
      if (t_out && soff[i] != 0xffffffffu) {
        st16_nt(a.t + soff[i], S[i]);
        if (FORM == 1 && a.mbits) a.mbits[soff[i] >> 3] = (unsigned char)relu_bits(S[i]);
      }
    }
  };

  // ---- phase 2: output rows 3rh .. 3rh+2 ; OUT = X + scale2 * (convB(T) + b2) [+ res2] ----
  {
    f32x4 acc[3][3];
    f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.b2) { const float4 t = *reinterpret_cast<const float4*>(a.b2 + 16 * q + 4 * g2); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] = b4;
    unsigned fb[8];
    if (rh == 0) {
      f8_bases(fb, (unsigned)F8_OFF_T8, 0, px2, g2);
      f8_sweep<2, E5M2, decltype(t_store)>(*reinterpret_cast<f32x4(*)[2][3]>(&acc[0]), A, lds, fb, sa2, sbt, t_store);   // output rows 0, 1 <- T rows 0 .. 3
      gate_wait(&gate[1], 4u);
      f8_bases(fb, (unsigned)F8_OFF_T8, 2, px2, g2);
      f8_sweep<1, E5M2>(*reinterpret_cast<f32x4(*)[1][3]>(&acc[2]), A, lds, fb, sa2, sbt);                                // output row 2 <- T rows 2 .. 4
    } else {
      f8_bases(fb, (unsigned)F8_OFF_T8, 3, px2, g2);
      f8_sweep<3, E5M2, decltype(t_store)>(acc, A, lds, fb, sa2, sbt, t_store);                                           // output rows 3 .. 5 <- T rows 3 .. 7
    }
    int px3 = px2, g3 = g2;
    asm volatile("" : "+v"(px3), "+v"(g3) : "v"(acc[2][2]));
    const int gp3 = 4 * (g3 & ~1), ch3 = 2 * q + (gp3 >> 3), c03 = 16 * q + 4 * g3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 tx = (k < 3) ? acc[k < 3 ? k : 0][0] : acc[0][2];
      const f32x4 ty = (k < 3) ? acc[k < 3 ? k : 0][1] : acc[1][2];
      float v[8], m[8];
      pair_up(tx, ty, g3, v);
      const int r = (k < 3) ? k : (g3 & 1), c = (k < 3) ? (g3 & 1) : 2;
      const int srow = 3 * rh + r, y = sy * BSH + srow, xx = 16 * c + px3;
      // branch-free as well: a pixel outside the image computes on the zeros of its LDS slot and is never stored (soff)
      const bool in = (y < a.H) & (xx < a.W);
      unsigned char* const pp = lc16 + swz(srow * BSW + xx, ch3);
      unpack8(*reinterpret_cast<const uint4*>(pp), m);                                            // residual = the input tile
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a.scale2, m[j]);
      if (a.res2) {
        unpack8(*reinterpret_cast<const uint4*>(a.res2 + (in ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gp3) : 0u)), m);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += m[j];
      }
      const uint2 lo = pack4_bf16(v[0], v[1], v[2], v[3]), hi = pack4_bf16(v[4], v[5], v[6], v[7]);
      *reinterpret_cast<uint4*>(pp) = make_uint4(lo.x, lo.y, hi.x, hi.y);                         // OUT image, in place of the input pixel
    }
    {
      const int srow = 3 * rh + 2, y = sy * BSH + srow, xx = 32 + px3;
      const bool in = (y < a.H) & (xx < a.W);
      float v[4] = {acc[2][2][0], acc[2][2][1], acc[2][2][2], acc[2][2][3]};
      float m[4];
      unsigned char* const pp = lc16 + swz(srow * BSW + xx, 2 * q + (g3 >> 1)) + (g3 & 1) * 8;
      unpack4_bf16(*reinterpret_cast<const uint2*>(pp), m);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], a.scale2, m[j]);
      if (a.res2) {
        unpack4_bf16(*reinterpret_cast<const uint2*>(a.res2 + (in ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c03) : 0u)), m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += m[j];
      }
      *reinterpret_cast<uint2*>(pp) = pack4_bf16(v[0], v[1], v[2], v[3]);
    }
  }
  gate_arrive(&gate[2 + rh], lane);
  gate_wait(&gate[2 + rh], 4u);
  {
    uint4 S[GROUP_REGS];
    f8_stage48(S, lc16, tg2, rh);
#pragma unroll
    for (int i = 0; i < GROUP_REGS; ++i)
      if (soff[i] != 0xffffffffu) st16_nt(a.out + soff[i], S[i]);
  }
  // this row half's amax (its four waves have added theirs in front of the gate) -> this row half's entry of the site record: plain stores, no
  // global atomics (1024 same-line device-scope atomics per launch cost it 4 us); rumpy_fp8_rotate takes the maximum over the entries
  if (tg2 == 0) {
    unsigned* e = a.site + RUMPY_FP8_SITE_HEAD + 2 * (2 * blockIdx.x + rh);
    e[0] = __hip_atomic_load(&amax_s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    e[1] = __hip_atomic_load(&amax_s[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

int rumpy_conv_block_fp8_launch(const rumpy_block_args* p, hipStream_t s) {
  const bool fwd = p->relu1 && p->scale1 == 1.0f && !p->mask;
  const bool dgrad = !p->relu1 && p->maskbits && !p->b1;
  if (!(fwd || dgrad) || p->res_mode != 0 || p->pool || p->fmt != RUMPY_FMT_BF16 || p->W > BSW || p->col_tile) {
    rumpy_set_error("rumpy_conv_block: the fp8 images go with the ResBlock forward form or its mask-byte data-gradient form, bf16 tensors, W <= %d", BSW);
    return RUMPY_E_ARG;
  }
  if (!p->w2_f8 || !p->f8_sw1 || !p->f8_sw2 || !p->f8_site) { rumpy_set_error("rumpy_conv_block: fp8 launch needs w1_f8, w2_f8, f8_sw1, f8_sw2 and f8_site"); return RUMPY_E_ARG; }
  BlockF8Dev d;
  d.x = (const uint16_t*)p->x; d.w1 = (const f8_v8i*)p->w1_f8; d.b1 = p->b1; d.w2 = (const f8_v8i*)p->w2_f8; d.b2 = p->b2;
  d.res2 = (const uint16_t*)p->res2; d.t = (uint16_t*)p->t; d.out = (uint16_t*)p->out; d.mbits = (unsigned char*)p->maskbits;
  d.N = p->N; d.H = p->H; d.W = p->W; d.sy_n = (p->H + BSH - 1) / BSH; d.scale1 = p->scale1; d.scale2 = p->scale2;
  d.sw1 = p->f8_sw1; d.sw2 = p->f8_sw2; d.site = p->f8_site;
  const dim3 grid(d.N * d.sy_n);
  if (p->f8_entries < 2 * (int)grid.x) { rumpy_set_error("rumpy_conv_block: f8_entries %d < 2 * %u workgroups (rumpy_fp8_site_entries)", p->f8_entries, grid.x); return RUMPY_E_ARG; }
  if (fwd) RUMPY_LAUNCH_PROBED(5, (conv_block_fp8_kernel<1>), grid, dim3(BTHREADS), s, d);
  else RUMPY_LAUNCH_PROBED(5, (conv_block_fp8_kernel<3>), grid, dim3(BTHREADS), s, d);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// rumpy_fp8_pack: fp32 OIHW master filters of 64 -> 64 3x3 convs -> the kernel's filter images [q 4][mfma 5][lane 64][32 bytes], forward image
// (A rows = output channels) and data-gradient image (the transposed, flipped filter: A rows = input channels), e4m3 of w / 2^(e - 127) with ONE
// exponent e per conv chosen so that amax / scale lies in [128, 256); e goes to *exponent.  One workgroup per conv, after every optimizer step.
// ---------------------------------------------------------------------------------------------------------------------------------------
// One workgroup of 1024 threads per conv: the filter is read ONCE, coalesced, into LDS (36864 floats, rows of 64 x 9 padded to 577 so that the
// gathers below spread over the banks), its amax falls out of the same pass, and both images are gathered from LDS.  (Earlier forms of this
// round: 16 parts per conv that each took the amax for themselves - 166 us per RCAN step for 400 convs; an amax launch + 8 converting parts
// per conv gathering from global memory - 102 us.)
constexpr int F8_WROW = 577;
__global__ void __launch_bounds__(1024) fp8_pack_kernel(const rumpy_fp8_pack_item* items) {
  const rumpy_fp8_pack_item it = items[blockIdx.x];
  __shared__ float wl[64 * F8_WROW];
  __shared__ float red[16];
  const int tid = threadIdx.x;
  float am = 0.f;
  {
    const float4* w4 = reinterpret_cast<const float4*>(it.w);
    float4 v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = w4[tid + 1024 * i];              // 64 * 64 * 9 / 4 = 9216 vectors
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      am = fmaxf(fmaxf(am, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
      const int e = 4 * (tid + 1024 * i), row = e / 576, c = e - row * 576;      // 576 % 4 == 0: a vector never straddles two rows
      float* d = wl + row * F8_WROW + c;
      d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
    }
  }
  am = f8_wave_max(am, tid & 63);
  if ((tid & 63) == 0) red[tid >> 6] = am;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) am = fmaxf(am, red[i]);
  int e8 = (int)((__float_as_uint(am) >> 23) & 255u) - 7;       // amax in [2^(E-127), 2^(E-126)) -> amax / 2^(E-134) in [128, 256)
  if (am == 0.f || !(am < 3e38f)) e8 = 127;
  e8 = e8 < 1 ? 1 : (e8 > 254 ? 254 : e8);
  if (tid == 0) *it.exponent = (unsigned)e8;
  const float scale = __uint_as_float((unsigned)e8 << 23);
  // word i of an image = bytes 4 (i & 7) .. + 3 of (lane, mfma, q); fwd: row = output channel co, k = input channel ci, tap (ky, kx);
  // dgrad: row = input channel, k = output channel, tap flipped
  constexpr int WORDS = 4 * 5 * 64 * 8;       // 10240 words per image
  for (int img = 0; img < 2; ++img) {
    unsigned* dst = reinterpret_cast<unsigned*>(img ? it.img_dgrad : it.img_fwd);
    if (!dst) continue;
    for (int i = tid; i < WORDS; i += 1024) {
      const int w4 = i & 7, lane = (i >> 3) & 63, m = (i >> 9) % 5, q = i / (5 * 64 * 8);
      const int r = lane & 15, g = lane >> 4, row = 16 * q + r;
      float f[4];
#pragma unroll
      for (int b4 = 0; b4 < 4; ++b4) {
        const int b = 4 * w4 + b4;
        int k, ky, kx;
        bool zero = false;
        if (m < 3) { ky = m; kx = b >> 4; }                    // P[ky]: bytes 0-15 tap (ky, 0), 16-31 tap (ky, 1)
        else if (m == 3) { ky = b >> 4; kx = 2; }              // Q01: bytes 0-15 tap (0, 2), 16-31 tap (1, 2)
        else { ky = 2; kx = 2; zero = b < 16; }                // Q2: bytes 0-15 zero, 16-31 tap (2, 2)
        k = 16 * g + (b & 15);
        float v = 0.f;
        if (!zero) v = img ? wl[k * F8_WROW + (row * 3 + (2 - ky)) * 3 + (2 - kx)] : wl[row * F8_WROW + (k * 3 + ky) * 3 + kx];
        f[b4] = v;
      }
      f8_v2s o = {0, 0};
      o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, f[0], f[1], scale, false);
      o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, f[2], f[3], scale, true);
      dst[i] = __builtin_bit_cast(unsigned, o);
    }
  }
}

extern "C" int rumpy_fp8_pack(const rumpy_fp8_pack_item* items, int32_t n, void* stream) {
  if (n <= 0) return 0;
  if (!items) { rumpy_set_error("rumpy_fp8_pack: null table"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(fp8_pack_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, items);
  return rumpy_check_launch("rumpy_fp8_pack");
}

// rumpy_fp8_rotate: n site records of `words` 32-bit words each from `sites`: exponent of image tensor k <- from the maximum over the record's
// amax entries (unchanged when nothing was recorded), entries cleared.  In front of every pass that runs fp8 launches; one workgroup per record.
__global__ void __launch_bounds__(256) fp8_rotate_kernel(unsigned* sites, int words) {
  unsigned* rec = sites + (size_t)blockIdx.x * words;
  const int cnt = min((int)rec[2], (words - RUMPY_FP8_SITE_HEAD) / 2);
  __shared__ unsigned red[2][256];
  unsigned m0 = 0u, m1 = 0u;
  for (int i = threadIdx.x; i < cnt; i += 256) {
    unsigned* e = rec + RUMPY_FP8_SITE_HEAD + 2 * i;
    m0 = max(m0, e[0]); m1 = max(m1, e[1]);
    e[0] = 0u; e[1] = 0u;
  }
  red[0][threadIdx.x] = m0; red[1][threadIdx.x] = m1;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      red[0][threadIdx.x] = max(red[0][threadIdx.x], red[0][threadIdx.x + off]);
      red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 2) {
    const unsigned m = red[threadIdx.x][0];
    const int E = (int)((m >> 23) & 255u);
    if (m != 0u && E != 255) {
      const int e = E - 7;
      rec[threadIdx.x] = (unsigned)(e < 1 ? 1 : (e > 254 ? 254 : e));
    } else if ((rec[threadIdx.x] & 255u) == 0u) {
      rec[threadIdx.x] = 127u;
    }
  }
}

extern "C" int rumpy_fp8_rotate(void* sites, int32_t n, int32_t words, void* stream) {
  if (n <= 0) return 0;
  if (!sites || words < RUMPY_FP8_SITE_HEAD) { rumpy_set_error("rumpy_fp8_rotate: null table / record shorter than its head"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(fp8_rotate_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (unsigned*)sites, words);
  return rumpy_check_launch("rumpy_fp8_rotate");
}
// entries a launch of rumpy_conv_block with fp8 images writes: two (one per row half) per workgroup
extern "C" int rumpy_fp8_site_entries(int32_t N, int32_t H, int32_t W) {
  (void)W;
  return 2 * N * ((H + BSH - 1) / BSH);
}

// test hook of the conversions' overflow behaviour (tests/test_fp8_gpu.py): out[2 i], out[2 i + 1] = the e4m3 / e5m2 byte of in[i] / scale.
// ovfl != 0: with MODE.FP16_OVFL set for the wave, as the fp8 kernels run (f8_saturating_mode): out-of-range values then convert to the largest
// finite number instead of NaN / inf.
__global__ void fp8_convert_kernel(const float* in, float scale, unsigned char* out, int n, int ovfl) {
  if (ovfl) f8_saturating_mode();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f8_v2s a = {0, 0}, b = {0, 0};
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(a, in[i], 0.f, scale, false);
  b = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(b, in[i], 0.f, scale, false);
  out[2 * i] = (unsigned char)(__builtin_bit_cast(unsigned, a) & 255u);
  out[2 * i + 1] = (unsigned char)(__builtin_bit_cast(unsigned, b) & 255u);
}
extern "C" int rumpy_fp8_convert(const float* in, float scale, void* out, int32_t n, int32_t ovfl, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(fp8_convert_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, scale, (unsigned char*)out, n, ovfl);
  return rumpy_check_launch("rumpy_fp8_convert");
}

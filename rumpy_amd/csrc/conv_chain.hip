// A chain of residual blocks (the EDSR body, forward or data-gradient direction) in ONE persistent launch - conv_block.hip's block with the strip
// resident in LDS from block to block, the two halo rows above and below handed over by the vertical neighbours THROUGH THE XCD'S L2 (round 5).
//
// History (DESIGN_HISTORY.md 4.2 items 5, 10, 12): two earlier chains lost to per-block launches because a hand-off between workgroups went through
// the memory side - on this 8-XCD part a write that another XCD must see is a write-through to HBM, 0.45 us per hop plus the acknowledgement in the
// chain.  tests/tools/overlap/xcd_probe.hip (round 4) showed the way out: an sc1 LOAD is served by the XCD's own L2 when the line is there, so a
// record stored with sc0 ONLY (stays dirty in the writer's L2) reaches a reader on the SAME XCD in 0.29 us and never leaves the die.  Round 5 measured
// what that buys the chain (profiles/r05_chain_xcd_local.txt: 16 blocks 32 x 48 x 48, training form): 225 / 231 us forward / data gradient against
// 268 / 265 us for one launch per block (-16 % / -13 %; memory-side hand-off: 289 / 285; no hand-off at all - wrong results - 219 / 220).
//
// Made correct by construction, not by dispatch order: a workgroup does not run strip blockIdx.x - it reads the XCD it is on (s_getreg XCC_ID) and
// CLAIMS a strip from that XCD's counter.  XCD x owns the images x, x + 8, ...: all strips of an image are claimed by workgroups behind one L2,
// whatever order or placement the dispatcher chose.  A workgroup whose XCD is oversubscribed (its counter is past its share: another queue holds CUs
// elsewhere) claims a leftover strip of another XCD.  Every strip publishes where it physically runs; a row half hands its rows over with sc0 stores
// only if the neighbour that reads them sits on the same XCD, and with write-through (sc1) stores otherwise.  Loads are sc1 everywhere: they see a
// dirty line of the own L2 and, failing that, the memory side - never a stale copy (the probe's cross-XCD sc1 / sc1 ping-pong relies on the same).
// OUT rows stored sc0 are ordinary dirty L2 lines: written back at the end of the kernel like any store, for the launches that follow.
//
// Needs every strip co-resident (N * ceil(H/6) <= CUs, one 512-thread workgroup per CU): a poll that does not complete within CH_TIMEOUT (0.5 s)
// stores a code in *status; every other poll of the launch then gives up at its next look at that word (chain_common.hpp::ch_poll_round), the
// launch drains in milliseconds with garbage results, the optimizer launch of the step reads the same word and changes nothing, and the host - which
// reads it back with the loss - switches the engine to one launch per block (engine.py::degrade; data-parallel ranks raise).  W <= 48.  Everything inside a block - sweeps, epilogues, row-half gates, whole-line stores -
// is conv_block.hip (forms 1 and 3): the results are bitwise those of one launch per block (tests/test_chain_gpu.py).
#include "chain_common.hpp"
#ifndef CHAIN_AHEAD
#define CHAIN_AHEAD 2      // fragment reads two groups ahead in the one- and two-row sweeps (round 5: 228.3 -> 226.4 us per launch, no spills; three-row sweeps: spills)
#endif

struct ChainBlk {
  const uint16_t* x; const uint4* w1; const float* b1; const uint4* w2; const float* b2;
  const uint16_t* res2; uint16_t* t; uint16_t* out; unsigned char* mbits; float scale1, scale2;
};
static_assert(sizeof(ChainBlk) == sizeof(rumpy_res_chain_block), "rumpy_res_chain_block is the device-side block record");
struct ChainDev {
  const ChainBlk* blk; int nblk, N, H, W, sy_n; unsigned* work; unsigned* status; int nxcd, fake_xcc, force_sc1;
  // the single conv at the chain's outer end (rumpy_res_chain_args.edge_*; edge_w = NULL: none): FORM 1 behind the last block, FORM 3 in front of the first
  const uint4* edge_w; const float* edge_b; const uint16_t* edge_x; const uint16_t* edge_res; uint16_t* edge_out;
};

// -DCHAIN_STAMPS (measurement builds only, tests/tools/r05_chain_stamps.sh): phase time stamps (s_memrealtime, 100 MHz) of every wave in the MIDDLE block
#ifdef CHAIN_STAMPS
__device__ unsigned long long* g_chain_stamps;
#define CH_STAMP(k) do { if (b == a.nblk / 2 && (threadIdx.x & 63) == 0 && g_chain_stamps) g_chain_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int rumpy_debug_chain_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_stamps), &buf, sizeof(buf)); }
#else
#define CH_STAMP(k) do { } while (0)
#endif

// Geometry (round 6): G6 = strips of 6 rows x 48 columns (W <= 48; BlockGeo<3, false, 6>, the kernel of round 5) and G4 = strips of 4 rows x 64 columns
// (48 < W <= 64: the reference's shipped training crops, div2k/edsr.toml:16 - 16 crops of 64 x 64 are 256 strips).  G4 has the registers and the LDS of G6: a
// row half owns 3 T rows x 4 column tiles = the 12 accumulator tiles of G6's 4 x 3, the images are 8 x 66 and 6 x 66 pixels = 118 KB (G6: 115 KB).  Per
// accumulator the MFMA order is conv_block.hip's in both: bitwise the per-block launches (which cut a 64-pixel image into two 32-column tiles).
typedef BlockGeo<3, false, 6> ChainG6;
typedef BlockGeo<4, false, 4> ChainG4;

// FORM 1: forward (ReLU, mask bytes written if given); FORM 3: data gradient (* scale1, mask bytes read)
// EDGE: with the single conv at the chain's outer end (its own instantiations: the plain chain keeps its register budget; G6 only)
template <int FORM, int FMT = RUMPY_FMT_BF16, bool EDGE = false, class G = ChainG6>
__global__ void __launch_bounds__(BTHREADS, 2) block_chain_kernel(ChainDev a) {
  constexpr int NC = G::NC, SH = G::SH, OR = G::OR, TR = G::TR, COLS = G::XC, OW = G::OW;       // (XC = TC = OW + 2 columns in both LDS images)
  constexpr int NP1 = TR * NC / 2;                     // paired tiles of the first epilogue (G6: 4 rows x 3 = 6 pairs; G4: 3 rows x 4 = 6 pairs)
  constexpr int NP2 = OR * NC / 2;                     // ... of the second (G6: 4 pairs + one single tile; G4: 4 pairs)
  constexpr int HREGS = (2 * OW * 8 + 255) / 256;      // a row half's two halo rows as 16-byte pieces per thread
  static_assert(!EDGE || SH == 6, "the edge conv is built into the 6-row geometry");
  static_assert(G::XC == G::TC && (NC == 3 || NC == 4) && (TR * NC) % 2 == 0, "geometries of this kernel");
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::XBYTES + G::TBYTES];
  __shared__ unsigned gate[10];            // per row half: T rows written [0,1], OUT rows written [2,3], halo rows in LDS [4,5], stores acknowledged [6,7], edge conv's rows staged [8,9]
  unsigned char* const ldx = lds;
  unsigned char* const ldt = lds + G::XBYTES;
  const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = wave & 3, rh = wave >> 2;
  // ---- which strip this workgroup runs: CLAIMED, per XCD (top of this file; chain_common.hpp) ----
  __shared__ int claim[4];
  const ChainPlace place = chain_claim(a.work, a.N, a.sy_n, a.nxcd, a.fake_xcc, claim);
  const unsigned epoch = place.epoch;
  const int strip = place.strip;
  const int n = strip / a.sy_n, sy = strip - n * a.sy_n;
  const bool has_nb = (rh == 0) ? (sy > 0) : (sy + 1 < a.sy_n);
  const int nb_strip = (rh == 0) ? strip - 1 : strip + 1;
  unsigned* const flags = chain_flags(a.work, gridDim.x);
  // this row half's hand-off: through the XCD's L2 (sc0 stores) when the neighbour that reads its rows runs on the same XCD, write-through otherwise
  const bool local = has_nb && !a.force_sc1 && chain_same_xcd(a.work, epoch, nb_strip, place.xcc, a.status);
  const ChainBlk b0 = a.blk[0];
  constexpr bool pre = FORM == 3 && EDGE;        // the edge conv in front of block 0: the tile loaded here is ITS input
  constexpr bool post = FORM == 1 && EDGE;

  // ---- block 0: input rows 6sy-2 .. 6sy+7, columns -1 .. 48 -> LDS (conv_block.hip) ----
  {
    const uint16_t* const src0 = pre ? a.edge_x : b0.x;
    uint4 R[G::XREGS];
    const int y0 = sy * SH - 2;
#pragma unroll
    for (int i = 0; i < G::XREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      const int lr = pix / COLS, lc = pix - lr * COLS;
      const int y = y0 + lr, x = lc - 1;
      const bool ok = (p < G::XPIECES) & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
      const int e = ok ? ((n * a.H + y) * a.W + x) * 64 + part * 8 : 0;
      uint4 v = ch_gld16(src0 + (unsigned)e);
      R[i] = keep_if(v, ok);
    }
    if (tid < 10) gate[tid] = 0u;
    if (tid < G::TROWS * 2 * 8) {          // border columns of the T image: convB's zero padding, never written by the epilogues
      const int row = tid >> 4, side = (tid >> 3) & 1, chunk = tid & 7;
      *reinterpret_cast<uint4*>(ldt + swz(row * COLS + side * (COLS - 1), chunk)) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < G::XREGS; ++i) {
      const int p = tid + BTHREADS * i;
      const int pix = p >> 3, part = p & 7;
      if (p < G::XPIECES) *reinterpret_cast<uint4*>(ldx + swz(pix, part)) = R[i];
    }
  }
  bf16x8 F[18];
  {
    const uint4* wp = (pre ? a.edge_w : b0.w1) + (size_t)q * 18 * 64 + lane0;
#pragma unroll
    for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(ch_gld16(wp + t * 64));
  }
  __syncthreads();

  // The edge conv's three rows of this row half, accumulated in acc3 (bias included) -> [+ res] -> rows 3rh+1 .. 3rh+3 of the T image (scratch here: no
  // sweep reads it), from where they leave as whole lines.  Pairs as in the blocks' second epilogue.
  auto edge_rows_to_t = [&](f32x4 (&acc3)[3][3], const uint16_t* res) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int px = lane & 15, g = lane >> 4, c0 = 16 * q + 4 * g, gpair = 4 * (g & ~1), chunk8 = 2 * q + (gpair >> 3);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 tx = (k < 3) ? acc3[k < 3 ? k : 0][0] : acc3[0][2];
      const f32x4 ty = (k < 3) ? acc3[k < 3 ? k : 0][1] : acc3[1][2];
      float v[8];
      pair_up(tx, ty, g, v);
      const int r = (k < 3) ? k : (g & 1), c = (k < 3) ? (g & 1) : 2;
      const int srow = 3 * rh + r, y = sy * BSH + srow, xx = 16 * c + px;
      if (y < a.H && xx < a.W) {
        if (res) {
          float m[8];
          unpack8<FMT>(ch_gld16(res + (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair)), m);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += m[j];
        }
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<uint4*>(ldt + swz((srow + 1) * BCOLS + xx + 1, chunk8)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    {
      const int srow = 3 * rh + 2, y = sy * BSH + srow, xx = 32 + px;
      if (y < a.H && xx < a.W) {
        float v[4] = {acc3[2][2][0], acc3[2][2][1], acc3[2][2][2], acc3[2][2][3]};
        if (res) {
          float m[4];
          unpack4<FMT>(ch_gld8b(res + (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0)), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        *reinterpret_cast<uint2*>(ldt + swz((srow + 1) * BCOLS + xx + 1, 2 * q + (g >> 1)) + (g & 1) * 8) = pack4<FMT>(v[0], v[1], v[2], v[3]);
      }
    }
  };
  if (pre) {
    // ---- the conv in front of the chain (data gradient of the body-end conv): this half's three rows from input rows 3rh .. 3rh+4 of the tile ----
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    f32x4 acc3[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc3[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned off[8][2];
    sweep_bases(off, 0u, 3 * rh + 1, lane & 15, lane >> 4);
    block_sweep<3, FMT>(acc3, F, lds, off);
    {
      const uint4* wp = b0.w1 + (size_t)q * 18 * 64 + lane0;      // block 0's first filter, under the epilogue
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(ch_gld16(wp + t * 64));
    }
    edge_rows_to_t(acc3, nullptr);
    gate_arrive(&gate[8 + rh], lane);
    gate_wait(&gate[8], 4u);
    gate_wait(&gate[9], 4u);                 // both halves are through their sweeps (they read each other's rows of the tile) and have staged their rows
    // the rows become the resident strip (input image rows 3rh+2 .. 3rh+4) and go to HBM for the neighbours and the weight gradient: whole lines
    const int tg = 64 * q + lane;
    uint4 S[GROUP_REGS];
    group_stage<1>(S, ldt, tg, rh);
#pragma unroll
    for (int i = 0; i < GROUP_REGS; ++i) {
      const int p = tg + 256 * i, pix = p >> 3, r = pix / BSW, col = pix - r * BSW;
      const unsigned so = group_piece_off(i, tg, rh, n, sy, a.H, a.W);
      if (p < GROUP_PIECES) *reinterpret_cast<uint4*>(ldx + swz((3 * rh + r + 2) * BCOLS + col + 1, p & 7)) = (so != 0xffffffffu) ? S[i] : make_uint4(0, 0, 0, 0);
      if (so != 0xffffffffu) { if (local) ch_store16_sc0(const_cast<uint16_t*>(b0.x) + so, S[i]); else ch_store16_sc1(const_cast<uint16_t*>(b0.x) + so, S[i]); }
    }
    gate_arrive(&gate[2 + rh], lane);
  }
  constexpr unsigned pre4 = pre ? 4u : 0u;       // the gates of the block boundary (OUT rows written, halo rows in, stores acknowledged) have seen one more round

  for (int b = 0; b < a.nblk; ++b) {
    const ChainBlk blk = a.blk[b];
    CH_STAMP(0);
    // Lane geometry is recomputed per block from an opaque copy of the lane id: hoisted out of the loop it would hold ~100 VGPRs for the
    // whole chain (the sweeps' read bases alone are 5 x 16) and spill; per block it is ~150 VALU instructions.
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int px = lane & 15, g = lane >> 4, tg = 64 * q + lane;
    const int c0 = 16 * q + 4 * g;
    const int gpair = 4 * (g & ~1);
    const int chunk8 = 2 * q + (gpair >> 3);
    // lane geometry that does not change from block to block: offsets of the T pairs and of this thread's store pieces in a [N,H,W,64] tensor
    // pairs of the first epilogue: three column tiles (G6): k < TR: (wave row k, column tile 0 | 1), k >= TR: (rows 2(k-TR) | 2(k-TR)+1, column tile 2);
    // four (G4): (wave row k / 2, column tiles 2(k%2) | 2(k%2)+1)
    auto p1_row = [&](int k) -> int { return NC == 3 ? ((k < TR) ? k : (2 * (k - TR) + (g & 1))) : k / 2; };
    auto p1_col = [&](int k) -> int { return NC == 3 ? ((k < TR) ? (g & 1) : 2) : 2 * (k % 2) + (g & 1); };
    unsigned moff[NP1], soff[G::GREGS];
#pragma unroll
    for (int k = 0; k < NP1; ++k) {
      const int jr = p1_row(k), c = p1_col(k);
      const int y = sy * SH - 1 + TR * rh + jr, xx = 16 * c + px;
      const bool in = ((unsigned)y < (unsigned)a.H) & (xx < a.W);
      moff[k] = in ? (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair) : 0xffffffffu;
    }
#pragma unroll
    for (int i = 0; i < G::GREGS; ++i) soff[i] = group_piece_off<G>(i, tg, rh, n, sy, a.H, a.W);
    // halo pieces of this row half: 2 rows x OW columns x 8 chunks (768 = 3 per thread | 1024 = 4); rows SH sy - 2, - 1 (half 0) or SH sy + SH, + 1 (half 1)
    unsigned hoff[HREGS], hlds[HREGS];
#pragma unroll
    for (int i = 0; i < HREGS; ++i) {
      const int p = tg + 256 * i, pix = p >> 3, r = pix / OW, col = pix - r * OW;
      const int y = (rh == 0) ? sy * SH - 2 + r : sy * SH + SH + r;
      hoff[i] = (p < 2 * OW * 8 && has_nb && (unsigned)y < (unsigned)a.H && col < a.W) ? (unsigned)(((n * a.H + y) * a.W + col) * 64 + (p & 7) * 8) : 0xffffffffu;
      hlds[i] = swz(((rh == 0) ? r : SH + 2 + r) * COLS + col + 1, p & 7);
    }
    const unsigned done = 4u * (unsigned)b + pre4;       // boundary-gate counts at the end of block b - 1
    const unsigned tdone = 4u * (unsigned)b;             // T-gate counts
    unsigned MB[FORM == 3 ? NP1 : 1];
    if (FORM == 3) {
#pragma unroll
      for (int k = 0; k < NP1; ++k) MB[FORM == 3 ? k : 0] = ch_gld8(blk.mbits + ((moff[k] != 0xffffffffu ? moff[k] : 0u) >> 3));
    }
    f32x4 acc[TR][NC];                                   // wave row jr (T row TR rh + jr)
    {
      f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (blk.b1) b4 = ch_gldf4(blk.b1 + c0);
#pragma unroll
      for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[r][c] = b4;
    }
    unsigned off[8][2];
    constexpr int NA = TR - 2;                           // T rows of a row half that need no halo row (G6: 2, G4: 1); the other two do
    if (b == 0 && !pre) {
      sweep_bases<COLS>(off, 0u, TR * rh, px, g);
      block_sweep<TR, FMT, NoHook, NC, COLS>(acc, F, lds, off);
    } else {
      // (a) the two T rows that need no halo row: row half 0 -> T rows 2, 3 (wave rows 2, 3); row half 1 -> T rows 4, 5 (wave rows 0, 1).
      // (Round 5 also tried NOT to bring the halves into step here - a half waits for its own OUT rows only, computes the three T rows that need nothing
      // from the other half, and the fourth behind the other half's gate - so that one half's epilogue could run beside the other's sweep as inside a
      // per-block launch: bitwise equal, 28.5 k against 28.9 k patches/s on one box, profiles/r05_negative_results.txt item 6.  Not kept.)
      gate_wait(&gate[2], done);
      gate_wait(&gate[3], done);                         // both halves' OUT rows of block b - 1 are in LDS (and nobody reads the old T image)
      CH_STAMP(1);
      sweep_bases<COLS>(off, 0u, (rh == 0) ? 2 : TR, px, g);      // (G6: T rows 2, 3 | 4, 5; G4: T row 2 | 3)
      block_sweep<NA, FMT, NoHook, NC, COLS, CHAIN_AHEAD>(*reinterpret_cast<f32x4(*)[NA][NC]>(&acc[(rh == 0) ? 2 : 0]), F, lds, off);
      CH_STAMP(2);
      // (b) publish block b - 1: this wave's OUT stores are acknowledged (under the sweep above) -> count in -> one lane stores the flag.  (Publishing IN
      // FRONT of the sweep - the neighbours see the flag a sweep earlier, this wave stalls 0.4 us for the rest of its acknowledgements - measured the
      // same within noise: 28.67 k against 28.78 k patches/s, profiles/r05_negative_results.txt item 8.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CH_STAMP(3);
      gate_arrive(&gate[6 + rh], lane);
      if (q == 0) {
        gate_wait(&gate[6 + rh], done);
        if (place.stall && b == 1) ch_stall();           // (test hook, chain_common.hpp::CH_W_STALL)
        if (lane == 0) { if (local) ch_store_flag_sc0(flags + (2 * strip + rh) * CH_FLAG_STRIDE, (epoch << 8) + (unsigned)b); else __hip_atomic_store(flags + (2 * strip + rh) * CH_FLAG_STRIDE, (epoch << 8) + (unsigned)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
      // (c) the neighbour's two rows: poll its flag, fetch, write to the halo rows of the input image
      if (has_nb) {
        const unsigned want = (epoch << 8) + (unsigned)b;
        unsigned spins = 0;
        unsigned long long t0 = 0;
        for (;;) {
          const unsigned f = __hip_atomic_load(flags + (2 * nb_strip + (1 - rh)) * CH_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((f >> 8) == epoch && (f & 0xffu) >= (want & 0xffu)) break;
          const int tr = ch_poll_round(spins, t0, a.status);
          if (tr == 1 && lane == 0) atomicExch(a.status, 0x500u + (unsigned)b);
          if (tr) break;
        }
      }
      CH_STAMP(4);
      uint4 Hr[HREGS];
#pragma unroll
      for (int i = 0; i < HREGS; ++i) {
        Hr[i] = ch_load16_sc1(blk.x + (hoff[i] != 0xffffffffu ? hoff[i] : 0u));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < HREGS; ++i)
        if (hoff[i] != 0xffffffffu) *reinterpret_cast<uint4*>(ldx + hlds[i]) = Hr[i];
      gate_arrive(&gate[4 + rh], lane);
      gate_wait(&gate[4 + rh], done);
      CH_STAMP(5);
      // (d) the two T rows that do: row half 0 -> T rows 0, 1 (input rows 0 .. 3); row half 1 -> T rows 6, 7 (input rows 6 .. 9)
      sweep_bases<COLS>(off, 0u, (rh == 0) ? 0 : 2 * TR - 2, px, g);      // (G6: input rows 0 .. 3 | 6 .. 9; G4: 0 .. 3 | 4 .. 7)
      block_sweep<2, FMT, NoHook, NC, COLS, CHAIN_AHEAD>(*reinterpret_cast<f32x4(*)[2][NC]>(&acc[(rh == 0) ? 0 : TR - 2]), F, lds, off);
    }
    CH_STAMP(6);
    // second filter: L2 hits that land under the epilogue
    {
      const uint4* wp = blk.w2 + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(ch_gld16(wp + t * 64));
    }
    // ---- epilogue 1 (conv_block.hip): pairs k < 4: (row k, col tile 0 | 1); k = 4: rows 0 | 1 of col tile 2; k = 5: rows 2 | 3 ----
#pragma unroll
    for (int k = 0; k < NP1; ++k) {
      f32x4 tx, ty;
      if (NC == 3) {
        tx = (k < TR) ? acc[k < TR ? k : 0][0] : acc[2 * (k < TR ? 0 : k - TR)][NC - 1];
        ty = (k < TR) ? acc[k < TR ? k : 0][1] : acc[2 * (k < TR ? 0 : k - TR) + 1][NC - 1];
      } else {
        tx = acc[(k / 2) % TR][(2 * (k % 2)) % NC];
        ty = acc[(k / 2) % TR][(2 * (k % 2) + 1) % NC];
      }
      if (FORM == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { tx[j] = relu_f32(tx[j]); ty[j] = relu_f32(ty[j]); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { tx[j] *= blk.scale1; ty[j] *= blk.scale1; }
      }
      float v[8];
      pair_up(tx, ty, g, v);
      const int jr = p1_row(k), c = p1_col(k);
      uint4 o = make_uint4(0, 0, 0, 0);                  // outside the image: convB's zero padding
      if (moff[k] != 0xffffffffu) {
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (FORM == 3) o = relu_mask_bits(o, MB[FORM == 3 ? k : 0]);
      }
      *reinterpret_cast<uint4*>(ldt + swz((TR * rh + jr) * COLS + 16 * c + px + 1, chunk8)) = o;
    }
    CH_STAMP(7);
    gate_arrive(&gate[rh], lane);
    gate_wait(&gate[rh], tdone + 4u);
    if (rh == 1) gate_wait(&gate[0], tdone + 4u);
    // the row half's own strip rows of T (+ mask bytes) -> HBM from the LDS image: whole lines, non-temporal, under the second sweep
    uint4 S[G::GREGS];
    const bool t_out = blk.t != nullptr;
    if (t_out) group_stage<1, G>(S, ldt, tg, rh);
    auto t_store = [&](int grp) {
      if (grp % 3 == 0 && grp / 3 < G::GREGS) {
        const int i = grp / 3 < G::GREGS ? grp / 3 : 0;
        if (t_out && soff[i] != 0xffffffffu) {
          ch_gst16_nt(blk.t + soff[i], S[i]);
          if (FORM == 1 && blk.mbits) ch_gst8(blk.mbits + (soff[i] >> 3), relu_bits(S[i]));
        }
      }
    };
    // ---- phase 2: OUT = X + scale2 * (convB(T) + b2) [+ res2], in place over the input image's centre rows ----
    f32x4 acc2[OR][NC];
    {
      f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (blk.b2) b4 = ch_gldf4(blk.b2 + c0);
#pragma unroll
      for (int r = 0; r < OR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc2[r][c] = b4;
    }
    if (rh == 0) {      // output rows 0 .. OR-2 from this half's own T rows 0 .. OR; row OR-1 also needs the other half's first T row
      sweep_bases<COLS>(off, (unsigned)G::XBYTES, 0, px, g);
      block_sweep<OR - 1, FMT, decltype(t_store), NC, COLS, CHAIN_AHEAD>(*reinterpret_cast<f32x4(*)[OR - 1][NC]>(&acc2[0]), F, lds, off, t_store);
      gate_wait(&gate[1], tdone + 4u);
      sweep_bases<COLS>(off, (unsigned)G::XBYTES, OR - 1, px, g);
      block_sweep<1, FMT, NoHook, NC, COLS, CHAIN_AHEAD>(*reinterpret_cast<f32x4(*)[1][NC]>(&acc2[OR - 1]), F, lds, off);
    } else {
      sweep_bases<COLS>(off, (unsigned)G::XBYTES, OR, px, g);
      block_sweep<OR, FMT, decltype(t_store), NC, COLS>(acc2, F, lds, off, t_store);
    }
    CH_STAMP(8);
    if (b + 1 < a.nblk || post) {                        // the next block's first filter (or the edge conv's) lands under the epilogue and the halo step
      const uint4* wp = (b + 1 < a.nblk ? a.blk[b + 1].w1 : a.edge_w) + (size_t)q * 18 * 64 + lane;
#pragma unroll
      for (int t = 0; t < 18; ++t) F[t] = as_bf16x8(ch_gld16(wp + t * 64));
    }
    // pairs of the second epilogue: G6: k < 3: (row k, column tile 0 | 1), k = 3: (rows 0 | 1, column tile 2) + the single tile (row 2, column tile 2) below;
    // G4: (row k / 2, column tiles 2(k%2) | 2(k%2)+1)
#pragma unroll
    for (int k = 0; k < NP2; ++k) {
      f32x4 tx, ty;
      if (NC == 3) { tx = (k < OR) ? acc2[k < OR ? k : 0][0] : acc2[0][NC - 1]; ty = (k < OR) ? acc2[k < OR ? k : 0][1] : acc2[1 % OR][NC - 1]; }
      else { tx = acc2[(k / 2) % OR][(2 * (k % 2)) % NC]; ty = acc2[(k / 2) % OR][(2 * (k % 2) + 1) % NC]; }
      float v[8], m[8];
      pair_up(tx, ty, g, v);
      const int r = NC == 3 ? ((k < OR) ? k : (g & 1)) : k / 2, c = NC == 3 ? ((k < OR) ? (g & 1) : 2) : 2 * (k % 2) + (g & 1);
      const int srow = OR * rh + r, y = sy * SH + srow, xx = 16 * c + px;
      if (y < a.H && xx < a.W) {
        unsigned char* cell = ldx + swz((srow + 2) * COLS + xx + 1, chunk8);
        unpack8<FMT>(*reinterpret_cast<const uint4*>(cell), m);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], blk.scale2, m[j]);
        if (blk.res2) {
          unpack8<FMT>(ch_gld16(blk.res2 + (unsigned)(((n * a.H + y) * a.W + xx) * 64 + 16 * q + gpair)), m);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += m[j];
        }
        const uint2 lo = pack4<FMT>(v[0], v[1], v[2], v[3]), hi = pack4<FMT>(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<uint4*>(cell) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    if (NC == 3) {                                       // (three column tiles: the ninth tile of a row half has no partner)
      const int srow = OR * rh + 2, y = sy * SH + srow, xx = 32 + px;
      if (y < a.H && xx < a.W) {
        unsigned char* cell = ldx + swz((srow + 2) * COLS + xx + 1, 2 * q + (g >> 1)) + (g & 1) * 8;
        float v[4] = {acc2[2 % OR][NC - 1][0], acc2[2 % OR][NC - 1][1], acc2[2 % OR][NC - 1][2], acc2[2 % OR][NC - 1][3]};
        float m[4];
        unpack4<FMT>(*reinterpret_cast<const uint2*>(cell), m);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], blk.scale2, m[j]);
        if (blk.res2) {
          unpack4<FMT>(ch_gld8b(blk.res2 + (unsigned)(((n * a.H + y) * a.W + xx) * 64 + c0)), m);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += m[j];
        }
        *reinterpret_cast<uint2*>(cell) = pack4<FMT>(v[0], v[1], v[2], v[3]);
      }
    }
    CH_STAMP(9);
    gate_arrive(&gate[2 + rh], lane);
    gate_wait(&gate[2 + rh], done + 4u);
    {                                                    // this row half's 3 OUT rows -> HBM: whole lines, write-through (the neighbours read them back)
      group_stage<2, G>(S, ldx, tg, rh);
#pragma unroll
      for (int i = 0; i < G::GREGS; ++i)
        if (soff[i] != 0xffffffffu) { if (local) ch_store16_sc0(blk.out + soff[i], S[i]); else ch_store16_sc1(blk.out + soff[i], S[i]); }
    }
    CH_STAMP(10);
  }
  if (post) {
    // ---- the conv behind the chain (EDSR's body-end conv + the global skip): block-start steps once more, then ONE sweep and its rows out through the T image ----
    const unsigned b = (unsigned)a.nblk, done = 4u * b;
    const ChainBlk last = a.blk[a.nblk - 1];
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int px = lane & 15, g = lane >> 4, tg = 64 * q + lane;
    f32x4 acc3[3][3];
    {
      f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (a.edge_b) b4 = ch_gldf4(a.edge_b + 16 * q + 4 * g);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc3[r][c] = b4;
    }
    unsigned hoff[3], hlds[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int p = tg + 256 * i, pix = p >> 3, r = pix / BSW, col = pix - r * BSW;
      const int y = (rh == 0) ? sy * BSH - 2 + r : sy * BSH + BSH + r;
      hoff[i] = (has_nb && (unsigned)y < (unsigned)a.H && col < a.W) ? (unsigned)(((n * a.H + y) * a.W + col) * 64 + (p & 7) * 8) : 0xffffffffu;
      hlds[i] = swz(((rh == 0) ? r : BSH + 2 + r) * BCOLS + col + 1, p & 7);
    }
    gate_wait(&gate[2], done);
    gate_wait(&gate[3], done);
    unsigned off[8][2];
    // the two rows that need no halo row (strip rows 1, 2 | 3, 4: input image rows 2 .. 5 | 4 .. 7)
    sweep_bases(off, 0u, (rh == 0) ? 2 : 4, px, g);
    block_sweep<2, FMT>(*reinterpret_cast<f32x4(*)[2][3]>(&acc3[(rh == 0) ? 1 : 0]), F, lds, off);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gate_arrive(&gate[6 + rh], lane);
    if (q == 0) {
      gate_wait(&gate[6 + rh], done);
      if (lane == 0) { if (local) ch_store_flag_sc0(flags + (2 * strip + rh) * CH_FLAG_STRIDE, (epoch << 8) + b); else __hip_atomic_store(flags + (2 * strip + rh) * CH_FLAG_STRIDE, (epoch << 8) + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    if (has_nb) {
      unsigned spins = 0;
      unsigned long long t0 = 0;
      for (;;) {
        const unsigned f = __hip_atomic_load(flags + (2 * nb_strip + (1 - rh)) * CH_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((f >> 8) == epoch && (f & 0xffu) >= (b & 0xffu)) break;
        const int tr = ch_poll_round(spins, t0, a.status);
        if (tr == 1 && lane == 0) atomicExch(a.status, 0x500u + b);
        if (tr) break;
      }
    }
    {
      uint4 Hr[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) Hr[i] = ch_load16_sc1(last.out + (hoff[i] != 0xffffffffu ? hoff[i] : 0u));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (hoff[i] != 0xffffffffu) *reinterpret_cast<uint4*>(ldx + hlds[i]) = Hr[i];
    }
    gate_arrive(&gate[4 + rh], lane);
    gate_wait(&gate[4 + rh], done);
    // the row that does (strip row 0 | 5)
    sweep_bases(off, 0u, (rh == 0) ? 1 : 6, px, g);
    block_sweep<1, FMT>(*reinterpret_cast<f32x4(*)[1][3]>(&acc3[(rh == 0) ? 0 : 2]), F, lds, off);
    edge_rows_to_t(acc3, a.edge_res);
    gate_arrive(&gate[8 + rh], lane);
    gate_wait(&gate[8 + rh], 4u);
    uint4 S[GROUP_REGS];
    group_stage<1>(S, ldt, tg, rh);
#pragma unroll
    for (int i = 0; i < GROUP_REGS; ++i) {
      const unsigned so = group_piece_off(i, tg, rh, n, sy, a.H, a.W);
      if (so != 0xffffffffu) ch_gst16_nt(a.edge_out + so, S[i]);
    }
  }
}

// (sized for the 4-row geometry, which has the more strips: one work buffer serves a plan's launches at either geometry)
extern "C" int64_t rumpy_res_chain_work_bytes(int32_t N, int32_t H) {
  return chain_work_bytes((int64_t)N * ((H + ChainG4::SH - 1) / ChainG4::SH));
}
// strips of an [N, H, W] launch (all of them must be co-resident): rows per strip 6 (W <= 48) or 4 (48 < W <= 64); 0: W is beyond the kernel
extern "C" int32_t rumpy_res_chain_strips(int32_t N, int32_t H, int32_t W) {
  if (W <= 0 || W > ChainG4::OW) return 0;
  const int sh = W <= ChainG6::OW ? ChainG6::SH : ChainG4::SH;
  return N * ((H + sh - 1) / sh);
}

extern "C" int rumpy_res_chain(const rumpy_res_chain_args* p, void* stream) {
  if (!p || !p->blocks || !p->work || !p->status || p->nblocks <= 0 || p->nblocks > 255) { rumpy_set_error("rumpy_res_chain: bad argument"); return RUMPY_E_ARG; }
  if (p->N <= 0 || p->H <= 0 || p->W <= 0 || p->W > ChainG4::OW) { rumpy_set_error("rumpy_res_chain: needs 0 < W <= 64 (got %d)", p->W); return RUMPY_E_ARG; }
  const bool wide = p->W > ChainG6::OW;        // 48 < W <= 64: strips of 4 rows x 64 columns
  if (wide && p->edge_w) { rumpy_set_error("rumpy_res_chain: the conv at the chain's outer end is built into the 6-row geometry (W <= 48)"); return RUMPY_E_ARG; }
  if (p->fmt != RUMPY_FMT_BF16 && !(p->fmt == RUMPY_FMT_F16 && !p->backward)) { rumpy_set_error("rumpy_res_chain: fmt %d is a forward-only format", p->fmt); return RUMPY_E_ARG; }
  const int sy_n = (p->H + (wide ? ChainG4::SH : ChainG6::SH) - 1) / (wide ? ChainG4::SH : ChainG6::SH);
  if (p->N * sy_n > rumpy_device_cus()) { rumpy_set_error("rumpy_res_chain: %d strips do not fit %d CUs (all must be co-resident)", p->N * sy_n, rumpy_device_cus()); return RUMPY_E_ARG; }
  if (p->work_bytes < rumpy_res_chain_work_bytes(p->N, p->H)) { rumpy_set_error("rumpy_res_chain: work buffer too small"); return RUMPY_E_ARG; }
  if (p->fake_xcc < 0 || (p->fake_xcc > 0 && !p->force_sc1)) { rumpy_set_error("rumpy_res_chain: fake_xcc (a test hook) goes with force_sc1"); return RUMPY_E_ARG; }
  ChainDev d;
  d.blk = reinterpret_cast<const ChainBlk*>(p->blocks); d.nblk = p->nblocks; d.N = p->N; d.H = p->H; d.W = p->W; d.sy_n = sy_n;
  d.work = (unsigned*)p->work; d.status = (unsigned*)p->status;
  d.nxcd = rumpy_device_xcds(); d.fake_xcc = p->fake_xcc; d.force_sc1 = p->force_sc1;
  d.edge_w = (const uint4*)p->edge_w; d.edge_b = p->edge_b; d.edge_x = (const uint16_t*)p->edge_x; d.edge_res = (const uint16_t*)p->edge_res; d.edge_out = (uint16_t*)p->edge_out;
  if (p->edge_w && (p->nblocks > 254 || (p->backward ? !p->edge_x : !p->edge_out))) {
    rumpy_set_error("rumpy_res_chain: the edge conv needs edge_x (backward) / edge_out (forward) and at most 254 blocks"); return RUMPY_E_ARG; }
  if (d.fake_xcc > 0) d.nxcd = d.fake_xcc < CH_MAX_XCD ? d.fake_xcc : CH_MAX_XCD;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(p->N * sy_n);
  if (wide) {
    if (p->backward) RUMPY_LAUNCH_PROBED(5, (block_chain_kernel<3, RUMPY_FMT_BF16, false, ChainG4>), grid, dim3(BTHREADS), s, d);
    else if (p->fmt == RUMPY_FMT_F16) RUMPY_LAUNCH_PROBED(5, (block_chain_kernel<1, RUMPY_FMT_F16, false, ChainG4>), grid, dim3(BTHREADS), s, d);
    else RUMPY_LAUNCH_PROBED(5, (block_chain_kernel<1, RUMPY_FMT_BF16, false, ChainG4>), grid, dim3(BTHREADS), s, d);
  } else if (p->edge_w) {
    if (p->backward) RUMPY_LAUNCH_PROBED(5, (block_chain_kernel<3, RUMPY_FMT_BF16, true>), grid, dim3(BTHREADS), s, d);
    else if (p->fmt == RUMPY_FMT_F16) RUMPY_LAUNCH_PROBED(5, (block_chain_kernel<1, RUMPY_FMT_F16, true>), grid, dim3(BTHREADS), s, d);
    else RUMPY_LAUNCH_PROBED(5, (block_chain_kernel<1, RUMPY_FMT_BF16, true>), grid, dim3(BTHREADS), s, d);
  } else if (p->backward) RUMPY_LAUNCH_PROBED(5, (block_chain_kernel<3>), grid, dim3(BTHREADS), s, d);
  else if (p->fmt == RUMPY_FMT_F16) RUMPY_LAUNCH_PROBED(5, (block_chain_kernel<1, RUMPY_FMT_F16>), grid, dim3(BTHREADS), s, d);
  else RUMPY_LAUNCH_PROBED(5, (block_chain_kernel<1>), grid, dim3(BTHREADS), s, d);
  return rumpy_check_launch("rumpy_res_chain");
}

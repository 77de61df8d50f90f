/*
 * rumpy_experimental.h - C ABI of the EXPERIMENTAL persistent-chain kernels (not part of the product library, not launched by the engine):
 * a whole chain of 64->64 layers / residual blocks in ONE launch with the strip resident in LDS and halo rows exchanged between
 * workgroups as epoch-tagged records.  Built, parity-tested and measured (tests/test_kernels_gpu.py, tests/tools/kbench.py chain|bchain);
 * they tie with or lose to the per-block launches, so they live with the measurement tools:
 * tests/tools/csrc/ -> tests/tools/csrc/librumpy_exp.so.
 */
#ifndef RUMPY_EXPERIMENTAL_H
#define RUMPY_EXPERIMENTAL_H
#include "rumpy_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- layer-resident chain of 64->64 3x3 convs (EDSR body, forward or data-gradient direction) in one launch ----
 * Every layer has rumpy_conv3x3 semantics with cin_chunks = cout_tiles = 1; layer l+1's input is layer l's output, kept
 * in LDS by the workgroup that owns the strip; halo rows travel between vertical neighbours through `xchg` as
 * epoch-tagged 8-byte records (see conv_chain.hip).
 * Requirements: W <= 48 and N * ceil(H/6) <= rumpy_device_cus() (all strips co-resident), nothing else running on the GPU.
 * *status != 0 afterwards means a hand-off timed out (results invalid). */
typedef struct {
  const void* w;        /* packed filter (fwd or dgrad image) */
  const float* bias;    /* packed bias or NULL */
  void* out;            /* [N,H,W,64] bf16, always written */
  const void* mask;     /* as rumpy_conv_args */
  const void* res1;
  const void* res2;
  int32_t relu;
  float scale;
} rumpy_chain_layer;
typedef struct {
  const void* x;                    /* [N,H,W,64] bf16 input of layer 0 */
  const rumpy_chain_layer* layers;  /* DEVICE array */
  int32_t nlayers;
  int32_t N, H, W;
  void* xchg;                       /* rumpy_conv_chain_xchg_bytes(N*ceil(H/6)) bytes: zeroed ONCE by the caller when
                                       allocated, then owned by the library (halo-row records + epoch header) */
  uint32_t* status;                 /* 1 word (zeroed by the call) */
  uint64_t* stamps;                 /* NULL, or diagnostics: [strip][wave 8][layer < 8][8] s_memrealtime stamps */
} rumpy_chain_args;
int rumpy_conv_chain(const rumpy_chain_args* a, void* stream);
int64_t rumpy_conv_chain_xchg_bytes(int32_t nstrips);

/* ---- a chain of residual blocks in one launch, the strip resident in LDS from block to block (conv_block_chain.hip) ----
 * blocks: DEVICE array of rumpy_block_args, block b+1's input is block b's output (x is read from blocks[0] only; res2 must be
 * NULL).  Needs N*ceil(H/6) <= CUs (every strip co-resident), W <= 48 and nothing else occupying CUs while it runs.
 * xchg: rumpy_block_chain_xchg_bytes(N*ceil(H/6)) bytes, zeroed ONCE by the caller at allocation, then owned by the library.
 * status: one device word, 0 after a clean run (a neighbour hand-off that timed out stores 0x200 + block index). */
typedef struct {
  const rumpy_block_args* blocks;
  int32_t nblocks;
  int32_t N, H, W;
  int32_t masked;      /* 1: the blocks carry ReLU masks (data-gradient chains), 0: none does (the masks are not even loaded) */
  void* xchg;
  uint32_t* status;
} rumpy_block_chain_args;
int rumpy_block_chain(const rumpy_block_chain_args* a, void* stream);
int64_t rumpy_block_chain_xchg_bytes(int32_t nstrips);

/* ---- the same chain, second design (conv_body_chain.hip): OUT rows leave as whole write-through lines, ONE flag word per (strip, row half)
 * and block announces them, the neighbour reads its two halo rows back from the OUT tensor (sc1 loads).  Placement-independent.
 * `blocks` = DEVICE array of rumpy_chain_block; block b's x MUST be block b-1's out.  Forms of rumpy_conv_block's ResBlock launches:
 * backward = 0: t = relu(conv1(x)+b1) [mask bytes -> maskbits], out = x + scale2*(conv2(t)+b2) [+ res2]; backward = 1: t = maskbits . scale1*convA(x),
 * out = x + scale2*convB(t) [+ res2].  flags = rumpy_body_chain_flag_bytes(N, H) bytes, epoch / status = device words, all zeroed once.
 * *status = 0x500 + block after a hand-off that timed out. */
typedef struct {
  const void* x; const void* w1; const float* b1; const void* w2; const float* b2;
  const void* res2; void* t; void* out; void* maskbits;
  float scale1, scale2;
} rumpy_chain_block;
typedef struct {
  const void* blocks;
  int32_t nblocks, N, H, W;
  int32_t backward, fmt;   /* fmt: RUMPY_FMT_*; F16 with backward = 0 only */
  void* flags; void* epoch; void* status;
} rumpy_body_chain_args;
int rumpy_body_chain(const rumpy_body_chain_args* a, void* stream);
int64_t rumpy_body_chain_flag_bytes(int32_t N, int32_t H);

/* ---- round 3: half-strip residual-block launches, two chains side by side (conv_hblock.hip; measured NOT to pay) ----
 * The two ResBlock forms of a training step (forward: relu1, scale1 = 1, maskbits written; data gradient: !relu1, maskbits read; bf16,
 * W <= 48) as HALF-strip launches (3 output rows per 256-thread workgroup, 76.8 KB of LDS: two workgroups per CU; csrc/conv_hblock.hip):
 * images [0, ceil(N/2)) on `stream`, the rest on the device's second stream - two launch chains whose boundaries, tile loads and
 * epilogues run under each other's MFMA sweeps.  Results are bitwise rumpy_conv_block's.  A run of consecutive split launches is opened
 * by RUMPY_SPLIT_FORK on its first launch (the second stream waits for everything queued on `stream` so far) and closed by
 * RUMPY_SPLIT_JOIN on its last (`stream` waits for the second stream); every launch between the two must be a split launch on the same
 * `stream`.  RUMPY_SPLIT_ONE_STREAM: both halves on `stream` (A/B, tests). */
#define RUMPY_SPLIT_FORK 1
#define RUMPY_SPLIT_JOIN 2
#define RUMPY_SPLIT_ONE_STREAM 4
typedef struct {
  rumpy_block_args block;
  int32_t flags;
  int32_t pad_;
} rumpy_block_split_args;
int rumpy_conv_block_split(const rumpy_block_split_args* a, void* stream);

/* ---- round 5, moved here in round 6 (measured slower than the per-block launches; never launched by the engine): a chain of RCABs in one persistent launch
 * (conv_rcab_chain.hip; work buffer, claims and hand-offs as rumpy_res_chain of the product library): the blocks of rumpy_rcab_fwd / rumpy_rcab_bwd (same arithmetic, bitwise the per-block launches), block b's x
 * BEING block b - 1's out; the strips of an image exchange their pool sums inside the launch through `xchg` (N * ceil(H/6) * 512 bytes, zeroed once; its own
 * buffer, not one shared with rumpy_rcab_fwd).  backward = 0: t = t1, t2 = conv2's output (both stored when set), mean / hidden / gate: out.
 * backward = 1: x = dy, w1 / w2 = DATA-GRADIENT images of conv2 / conv1, t = gt1 out, t2 = d_t2 out, t2_in = the forward pass's t2, maskbits, hidden / gate: in,
 * dz [, dzq]: out.  W <= 48, N * ceil(H/6) <= CUs, Cr <= 4, bf16.  *status: 0x4ff / 0x500 + block (hand-off) or 0x600 + block (pool exchange) after a time-out.
 * Replaces: the RCABs of ResidualGroup.body (rumpy/SISR/models/advanced/architectures.py:107-119: n_resblocks RCAB + conv; forward :121-124) run back to
 * back, and their autograd backward.  Measured at parity forward (19.7 against 19.8 us per RCAB) and slower backward (21.6 against 19.0): profiles/r05_rcab_chain.txt. */
typedef struct {
  const void* x; const void* w1; const float* b1; const void* w2; const float* b2;
  void* t; void* t2; const void* t2_in; const void* res2; void* out; void* maskbits;
  const float* ca_w1; const float* ca_b1; const float* ca_w2; const float* ca_b2;
  float* mean; float* hidden; float* gate; const float* qgate; float* dz; float* dzq;
} rumpy_rcab_chain_block;
typedef struct {
  const void* blocks;      /* DEVICE array of rumpy_rcab_chain_block */
  int32_t nblocks, N, H, W, cr, backward;
  void* work; int64_t work_bytes;      /* rumpy_rcab_chain_work_bytes(N, H), zeroed once */
  void* xchg; int64_t xchg_bytes; void* status;
  int32_t fake_xcc, force_sc1;         /* test hooks, as in rumpy_res_chain_args */
} rumpy_rcab_chain_args;
int rumpy_rcab_chain(const rumpy_rcab_chain_args* a, void* stream);
int64_t rumpy_rcab_chain_work_bytes(int32_t N, int32_t H);
int rumpy_debug_rcc_stamps(void* buf);   /* -DRCC_STAMPS builds only */

/* ---- round 6 (measured, not shipped - profiles/r06_block_body.txt): rumpy_res_chain of the product library with ONE wave per SIMD (conv_chain1.hip: 256-thread workgroups, 512 registers per wave, every per-lane address computed once in
 * front of the block loop, four- and six-row sweeps).  Same arguments, work buffer and results (bitwise); hand-offs always through the memory side (force_sc1 is
 * ignored); edge_w must be NULL. */
int rumpy_res_chain1(const rumpy_res_chain_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif

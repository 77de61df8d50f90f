/*
 * rumpy_amd.h - C ABI of the MI355X (gfx950) kernels behind the RUMpy SISR model-handler hot path.
 *
 * Boundary: the reference has NO native layer - below its Python handlers
 * (rumpy/shared_framework/models/base_architecture.py:442-485) sits ATen.  Each entry point here
 * replaces the ATen op(s) one reference call site dispatches; the call site is cited per function.
 * Conventions: every function is `int fn(const <args>* a, void* stream)`; returns 0 or a negative
 * RUMPY_E_* code (text via rumpy_last_error()); all pointers are DEVICE pointers owned by the caller
 * (PyTorch tensors); nothing is allocated; everything is asynchronous on `stream` (a hipStream_t);
 * re-entrant across streams; no global mutable state except the thread-local error string.
 *
 * Data layout: activations are NHWC bf16 (raw uint16), 64 channels per pixel = one 128-byte line
 * (tensors with 64*k channels are k such chunks per pixel).  Master weights stay fp32 OIHW (the
 * reference state_dict layout); rumpy_pack_weights produces the bf16 MFMA-fragment-ordered copies.
 */
#ifndef RUMPY_AMD_H
#define RUMPY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RUMPY_OK 0
#define RUMPY_E_ARG (-1)     /* invalid argument / unsupported shape */
#define RUMPY_E_LAUNCH (-2)  /* HIP launch error */

/* Element format of the stored activations and packed filters (`fmt` fields).  Training is bf16 throughout; the kernels an
 * EVALUATION plan launches (rumpy_head_fwd, rumpy_conv3x3, rumpy_conv_block / rumpy_rcab_fwd forward forms, rumpy_ca_fwd_fused,
 * rumpy_tail_fwd, rumpy_pack_weights) also take IEEE fp16: same bytes and MFMA rate, 11 instead of 8 significant bits, which is
 * what keeps the evaluation PSNR of a trained network within +-0.02 dB of the fp32 reference. */
#define RUMPY_FMT_BF16 0
#define RUMPY_FMT_F16 1
#define RUMPY_FMT_F16_RESIDUAL 2   /* rumpy_pack_weights, kind 0 only: the fp16 image of w - fp16(w), the rounding residual of the RUMPY_FMT_F16 image */
#define RUMPY_FMT_F32 3            /* fp32 elements: the conv outputs z of the degradation encoder's TRAINING forward pass only (rumpy_head_fwd /
                                      rumpy_enc_conv out, rumpy_enc_bn_train_keep x, rumpy_enc_bn_bwd z; round 4) */

#define RUMPY_TILE_H 8
#define RUMPY_TILE_W 16

const char* rumpy_last_error(void);
int rumpy_abi_version(void);
/* number of persistent workgroups the conv kernels want for `tiles` pixel tiles (host sizes workspaces with it) */
int rumpy_device_cus(void);

/* ---- 3x3 same-padding convolution, Cin = 64*cin_chunks, Cout = 64*cout_tiles, bf16 MFMA, fp32 accumulate ----
 * Replaces nn.Conv2d(k=3,p=1) + the elementwise ops fused around it:
 *   forward : common.py:6-9 (default_conv), common.py:62-73 (ReLU / .mul(res_scale) / res += x),
 *             architectures.py:72-83,118-123,238-239 (RCAB / group / global skips), common.py:30-33 (PixelShuffle)
 *   backward: the autograd dgrad of the same layers (base_architecture.py:432 loss.backward()) - call with the
 *             dgrad-packed filter; `mask` applies the ReLU derivative, in_mode=1 applies PixelShuffle^T.
 * v = acc + bias; if relu v = max(v,0); v *= scale; if mask v = mask>0 ? v : 0;
 * pool[n][tile][c] = sum over the tile of v (optional); v += res1 + res2; out = bf16(v)
 */
typedef struct {
  const void* x;        /* in_mode 0: [N,H,W,64*cin_chunks] ; in_mode 1: [N,2H,2W,64], chunk q=2i+j read at (2h+i,2w+j) */
  const void* w;        /* packed filter from rumpy_pack_weights (fwd or dgrad image) */
  const float* bias;    /* packed-order bias [64*cout_tiles] or NULL */
  void* out;            /* out_mode 0: [N,H,W,64*cout_tiles] ; out_mode 1: [N,2H,2W,64], tile q=2i+j stored at (2h+i,2w+j) */
  const void* mask;     /* bf16, layout of out (out_mode 0 only) or NULL */
  const void* res1;     /* bf16, layout of out, or NULL; with out_mode 1 (cin_chunks 1 only, res2 NULL): conv-output order [N,H,W,256] */
  const void* res2;     /* bf16, layout of out, or NULL */
  float* pool;          /* [N][rumpy_conv_pool_tiles()][64*cout_tiles] per-tile channel sums, or NULL */
  int32_t N, H, W;
  int32_t cin_chunks;   /* 1 or 4 */
  int32_t cout_tiles;
  int32_t in_mode, out_mode;
  int32_t relu;
  float scale;
  int32_t grid_x;       /* persistent workgroups per cout tile; 0 = library default */
  int32_t fmt;          /* RUMPY_FMT_*: format of x, w, out, res1, res2 (F16: cin_chunks 1, no mask) */
  const void* w_lo;     /* NULL, or (fmt F16, Cin = 64 forward launches that conv_up.hip takes: the upsampler convs of an evaluation plan) the
                           filter's rounding-residual image (rumpy_pack_weights fmt RUMPY_FMT_F16_RESIDUAL): out = epilogue(conv(x, w_lo) +
                           conv(x, w)), both sums in the same fp32 accumulators - one launch and one pass over x instead of a residual launch
                           into a scratch tensor that the main launch reads back (round 3) */
} rumpy_conv_args;
int rumpy_conv3x3(const rumpy_conv_args* a, void* stream);
/* number of per-image pool partial rows ("tiles") rumpy_conv3x3 writes for an H x W image */
int rumpy_conv_pool_tiles(int32_t H, int32_t W, int32_t cin_chunks);

/* ---- residual block in one launch: two 3x3 convs 64 -> 64, the activation between them stays in LDS (conv_block.hip) ----
 *   T = post1(convA(X)),  post1 = [+b1] [ReLU] [* scale1] [zero where mask <= 0];   OUT = X + scale2 * (convB(T) + b2) [+ res2]
 * forward of ResBlock (rumpy/SISR/models/advanced/common.py ResBlock, used by EDSR architectures.py:196-241):
 *   w1/b1 = conv1 forward image, relu1 = 1, scale1 = 1, mask = NULL, w2/b2 = conv2 forward image, scale2 = res_scale, T = saved activation
 * its data gradient: X = dOUT, w1 = conv2's data-gradient image, b1 = NULL, relu1 = 0, scale1 = res_scale, mask = saved T,
 *   w2 = conv1's data-gradient image, b2 = NULL, scale2 = 1, res2 = extra skip gradient or NULL; T = gradient w.r.t. the activation.
 * RCAB (architectures.py:60-84) forward: as ResBlock with res_mode = 1, scale2 = 1 and `pool` set (the channel-attention gate
 * and the skip follow in rumpy_ca_*); its data gradient: X = gradient after the gate, res_mode = 2, res1 = dOUT of the RCAB.
 * All tensors [N,H,W,64] bf16 (fp16: `fmt`); `t` may be NULL (inference: the activation is not stored).  Any W: up to 48 columns a
 * strip of 6 rows spans the image; wider images (ABI 3) are cut into column tiles of 32 or 48 output columns whose workgroups also
 * compute the intermediate activation on one halo column per side (the reference's shipped training crops are 64 x 64,
 * Documentation/sample_config_files/div2k/edsr.toml:16,26; every evaluation image is wider than 48). */
typedef struct {
  const void* x;
  const void* w1; const float* b1;
  const void* w2; const float* b2;
  const void* mask;
  const void* res2;
  void* t;
  void* out;
  int32_t N, H, W;
  int32_t relu1;
  float scale1, scale2;
  /* RCAB form (CALayer between the second conv and the skip, architectures.py:60-84): */
  int32_t res_mode;    /* 0: OUT = X + ..., the residual operand is the block input (taken from LDS) ; 1: no residual ;
                          2: the residual operand is `res1` */
  const void* res1;    /* [N,H,W,64] bf16, res_mode 2 */
  float* pool;         /* NULL, or per-(column tile, strip, row half) channel sums of scale2*(convB(T)+b2):
                          [N][rumpy_block_pool_tiles(H, W)][64] fp32, the partial sums rumpy_ca_mlp_fwd / rumpy_ca_fwd_fused reduce */
  void* maskbits;      /* NULL, or [N,H,W,8] bytes = the ReLU mask of T, one bit per channel (ResBlock form only): WRITTEN by a forward
                          launch (relu1 = 1), READ instead of `mask` by a data-gradient launch (relu1 = 0) - 1/16 of the mask traffic */
  int32_t fmt;         /* RUMPY_FMT_*; F16 for the forward forms (ResBlock: relu1 = 1, scale1 = 1, no mask; RCAB: res_mode 1 / 2) */
  int32_t col_tile;    /* 0: automatic (one strip across the image up to W = 48, column tiles of 32 or 48 columns beyond);
                          2 / 3: column tiles of 32 / 48 columns whatever W (tests: the two geometries against each other); not with `pool` */
  /* ABI 4 - precision 'fp8' (BASELINE.json config 5: "fp8 MFMA conv"; an opt-in of its own accuracy class).  With w1_f8
   * set, both sweeps of the launch run on the block-scaled fp8 MFMA (conv_block_fp8.hip): ResBlock forward form (relu1 = 1) or its
   * mask-byte data-gradient form (relu1 = 0, maskbits), W <= 48, res_mode 0, bf16 tensors in HBM as before; w1 / w2 are not read. */
  const void* w1_f8;        /* fp8 filter images of rumpy_fp8_pack (forward images, or the data-gradient images of conv2 / conv1), or NULL */
  const void* w2_f8;
  const uint32_t* f8_sw1;   /* their e8m0 scale exponents (rumpy_fp8_pack_item.exponent of the same conv) */
  const uint32_t* f8_sw2;
  uint32_t* f8_site;        /* this launch's record: [0] / [1] e8m0 exponents of the X / T images (read; written by rumpy_fp8_rotate), [2] number
                               of amax entries (set by the host, >= rumpy_fp8_site_entries), [3] unused, then `entries` pairs {amax of X, amax of
                               T} as fp32 bit patterns, one pair per (workgroup, row half), written by the launch */
  int32_t f8_entries;       /* entries the record has room for */
} rumpy_block_args;
int rumpy_conv_block(const rumpy_block_args* a, void* stream);

/* ---- precision 'fp8': filter images and delayed scaling, all on the device (conv_block_fp8.hip) ----
 * rumpy_fp8_pack: for every item, fp32 OIHW [64][64][3][3] master filter -> e4m3 images in the order the block-scaled MFMA sweeps read
 *   (forward and data-gradient image, 40960 bytes each) of w / 2^(e - 127), one exponent e per conv such that amax / scale is in [128, 256);
 *   e -> *exponent.  After every optimizer step / weight load (replaces nothing in the reference: torch has no fp8 conv; call site of the
 *   arithmetic it feeds: common.py:6-9 default_conv).
 * rumpy_fp8_rotate: n site records of `words` words each: exponent of each image tensor <- from the amax its launch left in the previous
 *   pass (amax / 2^(e - 127) in [128, 256); unchanged when nothing was recorded), entries cleared. */
#define RUMPY_FP8_SITE_HEAD 4
#define RUMPY_FP8_IMAGE_BYTES 40960
typedef struct {
  const float* w;
  void* img_fwd;            /* may be NULL */
  void* img_dgrad;          /* may be NULL */
  uint32_t* exponent;
} rumpy_fp8_pack_item;
int rumpy_fp8_pack(const rumpy_fp8_pack_item* items, int32_t n, void* stream);
int rumpy_fp8_rotate(void* sites, int32_t n, int32_t words, void* stream);
int rumpy_fp8_site_entries(int32_t N, int32_t H, int32_t W);   /* amax entries one fp8 launch on [N,H,W,64] writes */
int rumpy_block_pool_tiles(int32_t H, int32_t W);   /* rows of `pool` per image: 2 * ceil(H/6) * column tiles */

/* ---- a whole residual channel-attention block per launch (conv_rcab.hip): RCAB, rumpy/SISR/models/advanced/architectures.py:60-84, and
 * QRCAB, rumpy/SISR/models/attention_manipulators/architectures.py:154-228 (gate_q = `qgate`).
 *   rumpy_rcab_fwd:  t1 = relu(conv1(x)+b1) ; t2 = conv2(t1)+b2 ; gate = sigmoid(W2 relu(W1 mean_hw(t2) + b1c) + b2c) ; out = x + gate*[qgate*]t2
 *                    w1 / w2 = forward filter images of conv1 / conv2; t, t2 = stores of t1, t2 (training) or NULL; mean, hidden, gate: out.
 *   rumpy_rcab_bwd:  x = dy ; ds = sum_hw(dy*t2_in) -> dz [, dzq] ; d_t2 = dy*gate[*qgate] + dp/HW -> t2 ; t = [mask > 0] . conv2^T(d_t2) ;
 *                    out = dy + conv1^T(t).  w1 / w2 = DATA-GRADIENT filter images of conv2 / conv1; hidden, gate: in (from the forward launch).
 * The strips of an image (6 rows each; times column tiles when W > 48) exchange 64 partial sums through `xchg` (rumpy_rcab_xchg_bytes(N, H, W) bytes, zeroed ONCE at
 * allocation; one buffer can serve every block of a network, launches on one stream are serialised) as records tagged
 * (*epoch << 12) + seq: `epoch` is a device word the caller advances between passes (rumpy_rcab_epoch_advance), `seq` < 4096 must differ
 * between the launches of one pass.  Needs rumpy_rcab_strips(H, W) <= CUs, and the GPU to itself while a launch runs (kernels of other
 * processes / streams on the same XCDs can make the strips of an image wait for each other in a circle).  *status (device word, zero it
 * once) becomes 0x300 + seq if an exchange timed out: the results of that launch are invalid. */
typedef struct {
  const void* x; const void* w1; const float* b1; const void* w2; const float* b2;
  void* t; void* t2; const void* t2_in; const void* mask;
  const void* res2;    /* backward only: one more gradient added into `out` (a skip connection joining here), or NULL */
  void* out;
  int32_t N, H, W, cr;
  const float* ca_w1; const float* ca_b1; const float* ca_w2; const float* ca_b2;   /* conv_du.0 [cr,64],[cr] ; conv_du.2 [64,cr],[64] */
  float* mean; float* hidden; float* gate;       /* [N,64], [N,cr], [N,64] */
  const float* qgate; float* dz; float* dzq;     /* [N,64] each */
  void* xchg; int64_t xchg_bytes; const void* epoch; void* status;
  uint32_t seq;
  int32_t fmt;         /* RUMPY_FMT_*; F16: rumpy_rcab_fwd only */
  void* maskbits;      /* NULL, or [N,H,W,8] bytes: ReLU mask of t1, written by rumpy_rcab_fwd and read (instead of `mask`) by rumpy_rcab_bwd */
  /* ABI 4 - precision 'fp8' (as in rumpy_block_args; conv_rcab_fp8.hip): with w1_f8 set both sweeps run on the block-scaled fp8 MFMA;
   * W <= 48, bf16 tensors, backward with maskbits.  Image tensors of the site record: forward x / t1 (e4m3), backward d_t2 / gt1 (e5m2). */
  const void* w1_f8; const void* w2_f8;
  const uint32_t* f8_sw1; const uint32_t* f8_sw2;
  uint32_t* f8_site;
  int32_t f8_entries;
} rumpy_rcab_args;
int rumpy_rcab_fwd(const rumpy_rcab_args* a, void* stream);
int rumpy_rcab_bwd(const rumpy_rcab_args* a, void* stream);
int64_t rumpy_rcab_xchg_bytes(int32_t N, int32_t H, int32_t W);
int rumpy_rcab_strips(int32_t H, int32_t W);      /* workgroups per image = ceil(H/6) * column tiles: must be <= the device's CUs */
int rumpy_rcab_epoch_advance(void* epoch, void* stream);

/* ---- ABI 5: the same blocks per launch WITHOUT any exchange between workgroups (conv_rcab2.hip) - the attention gate is applied by the launch
 * that CONSUMES a block's output, from partial sums the producing launch stored; no residency requirement, no epoch / status words, any image size.
 *   rumpy_rcab2_fwd:  [u_in: x' = x + gate(part_in) * [qgate *] u_in, own rows -> x_out; mean / hidden / gate of THAT block: out]
 *                     t = relu(conv1(x') + b1) (stored when t is set; + maskbits) ; u_out = conv2(t) + b2, UNGATED ;
 *                     part_out[n][2 strip + row half][64] = its channel sums (rumpy_rcab2_partials(N, H, W) rows per image).
 *                     The attention MLP arguments (ca_*, cr, qgate, mean, hidden, gate) describe the block that produced u_in.
 *   rumpy_rcab2_bwd:  x = G = dL/d(block output) ; part_in = rows of sum_hw(G * U) of THIS block (written by the launch that produced G, or by
 *                     rumpy_ca_bwd_reduce) -> dz [, dzq] ; x_out = dU = G * gate [* qgate] + dp / HW ; t = [mask bits] . conv2^T(dU) ;
 *                     u_out = G + conv1^T(t) [+ res2] ; with u_in (the forward pass's U of the PREVIOUS block): part_out = rows of sum_hw(u_out * u_in).
 *                     w1 / w2 = DATA-GRADIENT filter images of conv2 / conv1; the attention MLP arguments describe this block (hidden, gate: in).
 * np_in > 64 (whole-image evaluation): the launch first folds the rows into one with a small kernel and needs part_scratch ([N][64] floats).
 * x + gate * u of a chain's LAST block: rumpy_ca_fwd_fused on part_out.
 * Replaces: RCAB.forward (rumpy/SISR/models/advanced/architectures.py:60-84: body = conv, ReLU, conv, CALayer; res += x) with CALayer
 * (architectures.py:24-44: AdaptiveAvgPool2d(1), conv_du = 1x1 conv, ReLU, 1x1 conv, Sigmoid; x * y) and its autograd backward; QRCAB's 'standard' /
 * 'modulate' styles (attention_manipulators/architectures.py:154-228) through qgate. */
typedef struct {
  const void* x; const void* u_in; const float* part_in; float* part_out; float* part_scratch;
  const void* w1; const float* b1; const void* w2; const float* b2;
  void* x_out; void* t; void* u_out;
  const void* res2;    /* backward only */
  void* maskbits;      /* [N,H,W,8] bytes: written by the forward launch (training), read by the backward launch */
  const float* ca_w1; const float* ca_b1; const float* ca_w2; const float* ca_b2;   /* conv_du.0 [cr,64],[cr] ; conv_du.2 [64,cr],[64] */
  float* mean; float* hidden; float* gate;       /* [N,64], [N,cr], [N,64] */
  const float* qgate; float* dz; float* dzq;     /* [N,64] each */
  int32_t N, H, W, cr, np_in;
  int32_t fmt;         /* RUMPY_FMT_*; F16: rumpy_rcab2_fwd only */
} rumpy_rcab2_args;
int rumpy_rcab2_fwd(const rumpy_rcab2_args* a, void* stream);
int rumpy_rcab2_bwd(const rumpy_rcab2_args* a, void* stream);
int rumpy_rcab2_partials(int32_t N, int32_t H, int32_t W);      /* rows of part_out per image */

/* ---- ABI 5: a CHAIN of residual blocks in one persistent launch (conv_chain.hip): the ResBlock forms of rumpy_conv_block - backward = 0:
 * t = relu(conv1(x) + b1) [mask bytes -> maskbits], out = x + scale2 * (conv2(t) + b2) [+ res2]; backward = 1: t = maskbits . scale1 * convA(x),
 * out = x + scale2 * convB(t) [+ res2] - for `nblocks` blocks, block b's x BEING block b - 1's out.  A workgroup keeps its strip in LDS from block to block;
 * the halo rows travel between vertical neighbours through the XCD's L2 (strips are claimed per XCD: all strips of an image run behind one L2) or, for a
 * strip that had to be claimed from another XCD, through the memory side.  Bitwise the per-block launches.  Needs rumpy_res_chain_strips(N, H, W) <= CUs, W <= 64 (round 6; W <= 48 before), and
 * `work` = rumpy_res_chain_work_bytes(N, H) bytes, zeroed once.  *status (device word, zero it once) becomes 0x4ff / 0x500 + block after a hand-off that
 * timed out: the results of that launch are invalid.  fake_xcc / force_sc1: test hooks (0 in production).
 * Replaces: the ResBlocks of EDSR.body (rumpy/SISR/models/advanced/architectures.py:218-224, 233: nn.Sequential of n_resblocks common.ResBlock;
 * common.py:62-73: conv, ReLU, conv, .mul(res_scale), res += x) run back to back by EDSR.forward (architectures.py:236-241), and their autograd data gradients. */
typedef struct {
  const void* x; const void* w1; const float* b1; const void* w2; const float* b2;
  const void* res2; void* t; void* out; void* maskbits;
  float scale1, scale2;
} rumpy_res_chain_block;
typedef struct {
  const void* blocks;      /* DEVICE array of rumpy_res_chain_block */
  int32_t nblocks, N, H, W;
  int32_t backward, fmt;   /* fmt: RUMPY_FMT_*; F16 with backward = 0 only */
  void* work; int64_t work_bytes; void* status;
  int32_t fake_xcc;        /* test hook: > 0 = pretend workgroup b runs on XCD b % fake_xcc (claim bookkeeping under oversubscription); needs force_sc1 */
  int32_t force_sc1;       /* test hook / A-B: every hand-off through the memory side */
  /* optional single 3x3 conv at the chain's OUTER end (edge_w != NULL): the body-end conv of EDSR (architectures.py:224: m_body.append(default_conv)).
   * backward = 0: after the last block, edge_out = conv(out_last, edge_w) + edge_b [+ edge_res] - `res = self.body(x); res += x` (architectures.py:238-239);
   * backward = 1: in front of the first block, blocks[0].x (a tensor the launch WRITES then) = conv(edge_x, edge_w), edge_w = the conv's data-gradient
   * filter image.  Bitwise rumpy_conv3x3 in front of / behind the chain.  nblocks <= 254 with it. */
  const void* edge_w; const float* edge_b; const void* edge_x; const void* edge_res; void* edge_out;
} rumpy_res_chain_args;
int rumpy_res_chain(const rumpy_res_chain_args* a, void* stream);
int64_t rumpy_res_chain_work_bytes(int32_t N, int32_t H);
/* ABI 6 (round 6): strips of an [N, H, W] launch - all of them must be co-resident (<= rumpy_device_cus()): 6 rows x 48 columns for W <= 48, 4 rows x 64 columns for
 * 48 < W <= 64 (the reference's shipped 64-pixel training crops: 16 crops = 256 strips; no edge conv at this geometry); 0: W is beyond the kernel */
int32_t rumpy_res_chain_strips(int32_t N, int32_t H, int32_t W);
int rumpy_device_xcds(void);   /* accelerator dies (XCDs, each with its own L2) of the current device: 8 on MI355X */

/* ---- head conv: Cin = C (<=4) fp32 NCHW image -> 64*cout_tiles ch NHWC bf16 (exact fp32 arithmetic) ----
 * Replaces nn.Conv2d(in_features, n_feats, 3, p=1): architectures.py:216,232 (EDSR head), :153,167 (RCAN head). */
typedef struct {
  const float* x;   /* [N,C,H,W] fp32 */
  const float* w;   /* [64*cout_tiles, C, 3, 3] fp32 OIHW master weights */
  const float* b;   /* [64*cout_tiles] */
  void* out;        /* [N,H,W,64*cout_tiles] bf16 */
  int32_t N, C, H, W, cout;
  float neg_slope_m1;  /* negative-side slope MINUS ONE of the activation applied to the output: 0 = none (SR head),
                          -0.9f = LeakyReLU(0.1) (first conv of the degradation encoder), -1 = ReLU */
  int32_t fmt;         /* RUMPY_FMT_* of `out` (incl. RUMPY_FMT_F32: fp32 [N,H,W,cout]) */
  int32_t pad_;
  const float* const* x_ind;   /* NULL, or a device word holding the address to read instead of `x` (rumpy_set_pointers): a captured
                                  hipGraph of the training step then follows the caller's batch without a copy into a fixed buffer */
} rumpy_head_fwd_args;
int rumpy_head_fwd(const rumpy_head_fwd_args* a, void* stream);

/* weight/bias gradient of the head conv (no data gradient: the image needs none).
 * Two launches inside: persistent partial sums into `slab` (>= rumpy_head_wgrad_slab_floats floats), then a
 * deterministic reduction into gw [cout,C,3,3] / gb [cout], multiplied by `scale`. */
typedef struct {
  const float* x;   /* [N,C,H,W] fp32 */
  const void* dy;   /* [N,H,W,cout] bf16 */
  float* slab;
  float* gw;        /* NULL: the slab reduction is left to rumpy_finish_reduce */
  float* gb;
  int32_t N, C, H, W, cout;
  float scale;
  const float* const* x_ind;   /* as in rumpy_head_fwd_args */
} rumpy_head_wgrad_args;
int rumpy_head_wgrad(const rumpy_head_wgrad_args* a, void* stream);
int64_t rumpy_head_wgrad_slab_floats(int32_t C, int32_t cout);

/* ---- tail conv: 64 ch NHWC bf16 -> C (<=4) fp32 NCHW, optionally fused with nn.L1Loss and its derivative ----
 * Replaces nn.Conv2d(n_feats, out_features, 3, p=1) (architectures.py:229,165), BaseModel.find_loss with
 * nn.L1Loss (base_architecture.py:40,448-449) and d|o-y|/do = sign(o-y) (the 1/numel factor is applied by
 * rumpy_wgrad_reduce / rumpy_head_wgrad through `scale`, so the stored gradient is exactly +-1 or 0). */
typedef struct {
  const void* x;         /* [N,H,W,64] bf16 */
  const void* w;         /* packed tail filter (rumpy_pack_weights kind 2, fwd image) */
  const float* bias;     /* [C] */
  float* out;            /* [N,C,H,W] fp32 */
  const float* target;   /* [N,C,H,W] fp32 or NULL */
  void* dy4;             /* [N,H,W,4] bf16 sign(out-target) or NULL */
  float* loss_partial;   /* [grid] per-workgroup sums of |out-target| (needs target) */
  float* loss;           /* scalar: sum(loss_partial)/numel, written by a second tiny launch (needs target) */
  int32_t N, C, H, W;
  int32_t grid_x;        /* persistent workgroups (<= capacity of loss_partial); 0 = default */
  float* wslab;          /* NULL, or (needs target) rumpy_tail_fwd_grid(..) * rumpy_wgrad_slab_floats(1) floats: the weight /
                            bias gradient of this conv w.r.t. the L1 loss is accumulated in the same pass (one slab per
                            workgroup; rumpy_tail_wgrad_reduce adds them up) instead of re-reading x in rumpy_wgrad_grouped */
  uint32_t* nonfinite;   /* NULL, or a device word that is OR-ed with 1 when an output value is not finite (an fp16 evaluation plan
                            that overflowed: the host re-runs the image in bf16) */
  int32_t fmt;           /* RUMPY_FMT_* of x and w (F16: no dy4 / wslab) */
  int32_t pad_;
  const float* const* target_ind;   /* NULL, or a device word holding the address to read instead of `target` (which must still be non-NULL:
                                       it says that there IS a target); see rumpy_head_fwd_args.x_ind */
} rumpy_tail_fwd_args;
int rumpy_tail_fwd(const rumpy_tail_fwd_args* a, void* stream);
/* table[0] = p0, table[1] = p1 in stream order (one tiny launch; the values travel as kernel arguments, no host buffer to keep alive):
 * the pointer table behind x_ind / target_ind.  Replaces the two device-to-device copies of the batch into the plan's fixed buffers that a
 * captured training step otherwise needs (base_architecture.py:425-431 hands run_train a new x / y every call). */
int rumpy_set_pointers(void* table, const void* p0, const void* p1, void* stream);
int rumpy_tail_fwd_grid(int32_t N, int32_t H, int32_t W, int32_t grid_x);   /* workgroups rumpy_tail_fwd launches = slabs written */
/* gw [C,64,3,3] / gb [C] = scale * sum of the slabs (fixed order) */
int rumpy_tail_wgrad_reduce(const float* wslab, int32_t nslabs, int32_t C, float scale, float* gw, float* gb, void* stream);

/* data gradient of the tail conv: dy4 [N,H,W,4] bf16 -> dx [N,H,W,64] bf16 */
typedef struct {
  const void* dy4;
  const void* w;    /* packed tail filter, dgrad image */
  void* dx;
  int32_t N, H, W;
} rumpy_tail_dgrad_args;
int rumpy_tail_dgrad(const rumpy_tail_dgrad_args* a, void* stream);

/* ABI 5: the tail conv's data gradient and the data gradient of the LAST upsampler stage (conv 64 -> 256 + PixelShuffle(2)) in one launch:
 *   dx [N,2H,2W,64] bf16 = rumpy_tail_dgrad(dy4 [N,2H,2W,4], w_tail)    (still written: the upsampler conv's weight gradient reads it)
 *   out [N,H,W,64]  bf16 = rumpy_conv3x3(x = dx, w, cin_chunks 4, in_mode 1, no bias / ReLU / residuals)
 * without reading dx back: every workgroup makes its input tiles from its window of dy4.  Bitwise the two launches.  H, W = the conv's grid
 * (half the tail's).  Replaces: autograd's backward of `x = self.tail(res)` (rumpy/SISR/models/advanced/architectures.py:226-230, 240: the last
 * Upsampler stage, common.py:30-33 (conv(n_feat, 4 n_feat) + PixelShuffle(2)), followed by default_conv(n_feats, out_features)) from the loss gradient down to the stage's input. */
typedef struct {
  const void* dy4;      /* [N,2H,2W,4] bf16 */
  const void* w_tail;   /* packed tail filter, dgrad image (rumpy_tail_dgrad_args.w) */
  void* dx;             /* [N,2H,2W,64] bf16, out */
  const void* w;        /* packed upsampler filter, dgrad image (rumpy_conv_args.w of the data-gradient launch) */
  void* out;            /* [N,H,W,64] bf16 */
  int32_t N, H, W;
  int32_t grid_x;       /* persistent workgroups; 0 = one per CU */
} rumpy_conv4d_tail_args;
int rumpy_conv4d_tail(const rumpy_conv4d_tail_args* a, void* stream);

/* fp32 NCHW [N,C<=4,H,W] -> bf16 [N,H,W,4] (zero padded): an upstream gradient entering the backward pass */
typedef struct {
  const float* src;
  void* dst;
  int32_t N, C, H, W;
} rumpy_nchw_to_nhwc4_args;
int rumpy_nchw_to_nhwc4(const rumpy_nchw_to_nhwc4_args* a, void* stream);

/* ---- beyond "64 features, PixelShuffle(2)" (round 2): EDSR at the reference's shipped width (Documentation/sample_config_files/div2k/edsr.toml:43-45,
 * 256 features) and the x3 upsampler (rumpy/SISR/models/advanced/common.py:39-44) ----
 * nn.PixelShuffle(r) (common.py:33,42) on an NHWC 16-bit map: dst[n, r h + i, r w + j, c] = src[n, h, w, c r^2 + i r + j] ; inverse != 0: the
 * opposite direction (what the backward pass applies to the incoming gradient).  N, H, W, F describe the LOW-resolution side: lo = [N,H,W,F r^2],
 * hi = [N,rH,rW,F].  The 64-feature x2 path fuses this into the conv kernels instead. */
typedef struct {
  const void* src;
  void* dst;
  int32_t N, H, W, F, r, inverse;
} rumpy_pixel_shuffle_args;
int rumpy_pixel_shuffle(const rumpy_pixel_shuffle_args* a, void* stream);
/* tail conv F -> C (C <= 4) for F = 64 k up to 512, on the fp32 VALU from the fp32 master filter w [C,F,3,3] (architectures.py:229):
 * rumpy_tail_fwd_wide: x = [N,H,W,F] bf16, out = fp32 NCHW [N,C,H,W] (+ bias) ; nonfinite: optional device flag, set when an output is not finite
 * rumpy_tail_dgrad_wide: x = dy4 [N,H,W,4] bf16 (rumpy_nchw_to_nhwc4), out = dx [N,H,W,F] bf16 ; bias / nonfinite unused
 * (the weight gradient is rumpy_wgrad_grouped with mt = 1 jobs per 64-channel chunk, as for F = 64) */
typedef struct {
  const void* x;
  const float* w;
  const float* bias;
  void* out;
  int32_t* nonfinite;
  int32_t N, H, W, F, C;
  int32_t fmt;            /* RUMPY_FMT_* of x (rumpy_tail_fwd_wide; evaluation plans store fp16) ; the data gradient is bf16 */
} rumpy_tail_wide_args;
int rumpy_tail_fwd_wide(const rumpy_tail_wide_args* a, void* stream);
int rumpy_tail_dgrad_wide(const rumpy_tail_wide_args* a, void* stream);

/* ---- weight gradient of the 3x3 convs: grouped launch over a job table + deterministic slab reduction ----
 * Replaces the wgrad/bias-grad half of loss.backward() (base_architecture.py:432) for every nn.Conv2d(k=3)
 * with 64-multiple input channels.  One job = one (layer, cin chunk, cout tile, image range); it leaves
 * fp32 partial sums [16*mt][9][64] (+ [16*mt] bias sums) in its slab. */
typedef struct {
  const void* x;     /* [N,H,W,x_cstride] bf16, channels x_coff..x_coff+63 used */
  const void* dy;    /* dy_mode 0: [N,H,W,dy_cstride] bf16 at channel offset dy_coff ;
                        dy_mode 1: [N,2H,2W,64], sub-pixel q = dy_coff ; dy_mode 2: [N,H,W,4] (mt must be 1) */
  float* slab;       /* rumpy_wgrad_slab_floats(mt) floats */
  int32_t n0, n1;    /* image range */
  int32_t t0, t1;    /* tile sub-range inside that image range: tiles [t0, t1) of (n1-n0)*tiles_y*tiles_x, row-major */
  int32_t H, W;
  int32_t x_cstride, x_coff;
  int32_t dy_mode, dy_cstride, dy_coff;
  int32_t mt;        /* 4: 64 output channels ; 1: <=4 output channels (tail) */
} rumpy_wgrad_job;
/* variant 0 = streaming LDS-DMA kernel (mt == 1 needs an even image width: its dy pieces are pixel pairs);
 * variant 1 = register-staged kernel (any width; also the A/B reference) */
int rumpy_wgrad_grouped(const rumpy_wgrad_job* jobs_device, int32_t njobs, int32_t mt, int32_t variant, void* stream);
/* The same jobs (mt = 4, streaming kernel) as per-workgroup LISTS: workgroup w of `nshares` runs jobs first[w] .. first[w+1]-1 one after the
 * other (first: DEVICE array of nshares + 1 offsets into jobs_device).  The engine cuts the cost-weighted concatenated tile sequence of all
 * layers into equal shares, one per CU; a share's part of one layer is one job with its own slab. */
int rumpy_wgrad_shares(const rumpy_wgrad_job* jobs_device, const int32_t* first_device, int32_t nshares, void* stream);
int64_t rumpy_wgrad_slab_floats(int32_t mt);

typedef struct {
  const float* slab;    /* first slab of this item's jobs */
  int64_t slab_stride;  /* floats between consecutive jobs' slabs */
  int32_t njobs;
  int32_t mt;
  int32_t co_count;     /* real output channels in the slab (64, or C for the tail) */
  int32_t co_mode;      /* 0: co = co_off + c ; 1: co = 4*c + co_off (PixelShuffle order) */
  int32_t co_off;
  int32_t ci_total;     /* Cin of the layer */
  int32_t ci_off;       /* channel offset of this item's cin chunk */
  int32_t write_bias;   /* 1: also reduce the bias sums into gb */
  float scale;
  float* gw;            /* [Cout,Cin,3,3] fp32 (OIHW) */
  float* gb;            /* [Cout] */
} rumpy_reduce_item;
int rumpy_wgrad_reduce(const rumpy_reduce_item* items_device, int32_t nitems, void* stream);

/* ---- filter packing: fp32 OIHW master weights -> bf16 MFMA-fragment images (run after every optimizer step) ---- */
typedef struct {
  const float* w;      /* [Cout,Cin,3,3] */
  const float* b;      /* [Cout] */
  void* w_fwd;         /* kind 0: Cout*Cin*9 elements (format `fmt`) in MFMA A-fragment order [cout_tile][cin_chunk][co quarter 4][s = tap*2 +
                          ci half, 18][lane 64][8] ; kind 2: [18][64][8] - with fmt RUMPY_FMT_F16 TWICE that: the fp16 filter image, then the image of
                          its rounding residual w - fp16(w) (evaluation plans: rumpy_tail_fwd multiplies both) */
  void* w_dgrad;       /* kind 0: same for the transposed, flipped filter ; kind 2: [4][2][64][8] ; may be NULL */
  float* b_packed;     /* kind 0: [Cout] in packed channel order ; else NULL */
  int32_t cout, cin;
  int32_t kind;        /* 0: 64-multiple conv ; 2: tail conv (cout<=4, cin=64) */
  int32_t shuffle;     /* kind 0: 1 = output channels grouped by PixelShuffle sub-pixel (co = 4c+q -> tile q, channel c) */
  int32_t fmt;         /* RUMPY_FMT_* of the images */
  int32_t pad_;
} rumpy_pack_item;
int rumpy_pack_weights(const rumpy_pack_item* items_device, int32_t nitems, void* stream);

/* ---- channel attention (RCAN CALayer, architectures.py:24-44) ---- */
typedef struct {
  const float* pool;   /* [N][ntiles][C] per-tile sums from rumpy_conv3x3 */
  const float* w1;     /* [Cr,C,1,1] */
  const float* b1;     /* [Cr] */
  const float* w2;     /* [C,Cr,1,1] */
  const float* b2;     /* [C] */
  float* mean;         /* [N][C]  out: pooled mean */
  float* hidden;       /* [N][Cr] out: relu(W1 mean + b1) */
  float* gate;         /* [N][C]  out: sigmoid(W2 hidden + b2) */
  int32_t N, C, Cr, ntiles;
  float inv_hw;
} rumpy_ca_mlp_fwd_args;
int rumpy_ca_mlp_fwd(const rumpy_ca_mlp_fwd_args* a, void* stream);

/* out = res + t * gate[n][c]  (x*y at architectures.py:44 fused with `res += x` at :83) */
typedef struct {
  const void* t;       /* [N,HW,C] bf16 */
  const void* res;     /* [N,HW,C] bf16 or NULL */
  const float* gate;   /* [N][C] */
  void* out;
  int32_t N, HW, C;
} rumpy_ca_scale_args;
int rumpy_ca_scale_res_fwd(const rumpy_ca_scale_args* a, void* stream);

/* dgate_partial[n][chunk][c] = sum over 128-pixel chunks of dy * t */
typedef struct {
  const void* dy;
  const void* t;
  float* partial;      /* [N][nchunks][C], nchunks = ceil(HW/128) */
  int32_t N, HW, C;
} rumpy_ca_bwd_reduce_args;
int rumpy_ca_bwd_reduce(const rumpy_ca_bwd_reduce_args* a, void* stream);

typedef struct {
  const float* partial;  /* from rumpy_ca_bwd_reduce */
  const float* mean;
  const float* hidden;
  const float* gate;
  const float* w1;
  const float* w2;
  float* dpool;          /* [N][C] out: d(mean)/HW, to be broadcast-added by rumpy_ca_bwd_apply */
  float* gw1;
  float* gb1;
  float* gw2;
  float* gb2;            /* parameter gradients (overwritten), multiplied by scale */
  int32_t N, C, Cr, nchunks;
  float inv_hw;
  float scale;
} rumpy_ca_mlp_bwd_args;
int rumpy_ca_mlp_bwd(const rumpy_ca_mlp_bwd_args* a, void* stream);   /* gw1 = gb1 = gw2 = gb2 = NULL: dpool only (parameter
                                                                          gradients deferred to rumpy_ca_mlp_bwd_params) */
/* the parameter gradients of `nitems` channel-attention layers in one launch; items: DEVICE array of the arguments the
 * per-layer calls were made with (same N, C, Cr), now with the four gradient pointers set; run after all of them */
int rumpy_ca_mlp_bwd_params(const rumpy_ca_mlp_bwd_args* items_device, int32_t nitems, int32_t N, int32_t C, int32_t Cr, void* stream);

/* dt = dy * gate[n][c] + dpool[n][c] */
typedef struct {
  const void* dy;
  const float* gate;
  const float* dpool;
  void* dt;
  int32_t N, HW, C;
} rumpy_ca_bwd_apply_args;
int rumpy_ca_bwd_apply(const rumpy_ca_bwd_apply_args* a, void* stream);

/* ---- fused forms (what the engine launches): the MLP is recomputed per workgroup inside the streaming kernel ----
 * forward: rumpy_ca_mlp_fwd + rumpy_ca_scale_res_fwd in one launch; backward: the dpool part of rumpy_ca_mlp_bwd +
 * rumpy_ca_bwd_apply in one launch, dz[n][c] (gradient before the sigmoid) is stored for rumpy_ca_mlp_bwd_params
 * (pass it there as `partial` with nchunks = 1).  Results are identical to the separate calls. */
typedef struct {
  const float* pool; const float* w1; const float* b1; const float* w2; const float* b2;
  float* mean; float* hidden; float* gate;     /* [N,C], [N,Cr], [N,C]: saved for the backward pass */
  /* pool: [N,ntiles,C] partial sums from the conv epilogue; SCRATCH - with more than 32 rows per image they are folded in place */
  const void* t; const void* res; void* out;   /* out = res + t * gate (res may be NULL) */
  int32_t N, HW, C, Cr, ntiles;
  float inv_hw;
  const float* qgate;                          /* NULL, or the meta-attention gate [N,C] of a QRCAB: out = res + t * gate * qgate */
  int32_t fmt;                                 /* RUMPY_FMT_* of t, res, out */
  int32_t pad_;
} rumpy_ca_fwd_fused_args;
int rumpy_ca_fwd_fused(const rumpy_ca_fwd_fused_args* a, void* stream);
typedef struct {
  const void* dy; const float* partial; const float* hidden; const float* gate; const float* w1; const float* w2;
  float* dz;                                   /* [N,C] out */
  void* dt;                                    /* dt = dy * gate + dpool */
  int32_t N, HW, C, Cr, nchunks;
  float inv_hw;
  const float* qgate;                          /* NULL, or the meta-attention gate [N,C] used in the forward pass */
  float* dzq;                                  /* with qgate: [N,C] out, gradient before the meta-attention sigmoid */
} rumpy_ca_bwd_fused_args;
int rumpy_ca_bwd_fused(const rumpy_ca_bwd_fused_args* a, void* stream);

/* ---- meta-attention q-layers (ParaCALayer, rumpy/SISR/models/attention_manipulators/q_layer.py:5-45; QRCAB architectures.py:154-228):
 * gate_q[n][c] = sigmoid(W2 relu(W1 m_n + b1) + b2), m_n = metadata vector [M] of image n.  All layers of a network in one launch
 * (items: DEVICE array); the parameter gradients likewise, from dzq written by rumpy_ca_bwd_fused. */
typedef struct {
  const float* w1; const float* b1;     /* [Hq,M], [Hq]  (attribute_integrator.0) */
  const float* w2; const float* b2;     /* [C,Hq], [C]   (attribute_integrator.2) */
  float* hidden;                        /* [N,Hq] out of the forward launch */
  float* gate;                          /* [N,C]  out of the forward launch */
  const float* dzq;                     /* [N,C]  in of the backward launch */
  float* gw1; float* gb1; float* gw2; float* gb2;
  float scale;
  int32_t pad_;
} rumpy_q_mlp_item;
int rumpy_q_mlp_fwd(const rumpy_q_mlp_item* items_device, int32_t nitems, const float* meta, int32_t N, int32_t M, int32_t Hq, int32_t C, void* stream);
int rumpy_q_mlp_bwd_params(const rumpy_q_mlp_item* items_device, int32_t nitems, const float* meta, int32_t N, int32_t M, int32_t Hq, int32_t C, void* stream);
/* gradient at the metadata input: dmeta[N, M] = sum over the layers of scale * W1^T (relu'(hidden) * W2^T dzq) - what autograd hands to whatever
 * produced the metadata (the degradation encoder of the blind pipeline when its trunk trains jointly: contrastive_blind_sr.py:337) */
int rumpy_q_mlp_bwd_meta(const rumpy_q_mlp_item* items_device, int32_t nitems, int32_t N, int32_t M, int32_t Hq, int32_t C, float* dmeta, void* stream);

/* ABI 5: the same q-layer with ParaCALayer's `num_layers` (q_layer.py:13,22-41) other than 2: layer l = [n[l+1], n[l]] weights + bias, ReLU behind every
 * layer but the last (nonlinearity=True, as QRCAB builds it: architectures.py:182-183), sigmoid behind the last.  n[0] = M (metadata), n[nlayers] = C;
 * every q-layer of a network has the same shape, stated by the caller (n, nlayers: HOST values).  acts: the post-ReLU outputs of layers 0 .. L-2 per
 * image, concatenated ([N, n[1] + .. + n[L-1]]).  N <= 64, every n <= 256, n[1] + .. + n[L] <= 448. */
#define RUMPY_QN_MAX_LAYERS 8      /* (4 until round 6; the item's layout changed with it) */
typedef struct {
  const float* w[RUMPY_QN_MAX_LAYERS]; const float* b[RUMPY_QN_MAX_LAYERS];
  float* gw[RUMPY_QN_MAX_LAYERS]; float* gb[RUMPY_QN_MAX_LAYERS];
  float* acts;                          /* out of the forward launch */
  float* gate;                          /* [N,C] out of the forward launch */
  const float* dzq;                     /* [N,C] in of the backward launches (gradient before the sigmoid) */
  int32_t n[RUMPY_QN_MAX_LAYERS + 1];
  int32_t nlayers;
  float scale;
  int32_t pad_;
} rumpy_q_mlpn_item;
int rumpy_q_mlpn_fwd(const rumpy_q_mlpn_item* items_device, int32_t nitems, const float* meta, int32_t N, const int32_t* n, int32_t nlayers, void* stream);
int rumpy_q_mlpn_bwd_params(const rumpy_q_mlpn_item* items_device, int32_t nitems, const float* meta, int32_t N, const int32_t* n, int32_t nlayers, void* stream);
int rumpy_q_mlpn_bwd_meta(const rumpy_q_mlpn_item* items_device, int32_t nitems, int32_t N, const int32_t* n, int32_t nlayers, float* dmeta, void* stream);

/* ---- the other QCALayer styles (rumpy/SISR/models/attention_manipulators/architectures.py:41-136: 'max_concat', 'mini_concat',
 * 'extended_attention', 'softmax'): the gate of a block is an MLP of at most four layers over the block's channel means and the image's
 * attribute vector (qca_style.hip).  Layer l: v = [previous output (n_prev) ; attr (M) if cat]; v = relu(v) if relu_in; out = act(W v + b),
 * act: 0 none, 1 ReLU, 2 sigmoid, 3 sigmoid then softmax over the outputs (last layer only).  These launches stand where rumpy_ca_mlp_fwd /
 * rumpy_ca_mlp_bwd stand for the plain CALayer: pool -> rumpy_qca_gate_fwd -> rumpy_ca_scale_res_fwd; rumpy_ca_bwd_reduce ->
 * rumpy_qca_gate_bwd -> rumpy_ca_bwd_apply; then ONE rumpy_qca_bwd_params over all layers of a network (items: DEVICE array of the
 * arguments of the gate launches, with gw / gb / scale set).  fp32 throughout. */
#define RUMPY_QCA_ACT_STRIDE 320   /* saved floats per image: the C means + every layer's outputs */
typedef struct {
  const float* w; const float* b;   /* [n_out][n_prev + (cat ? M : 0)], [n_out] */
  float* gw; float* gb;             /* gradients, written by rumpy_qca_bwd_params */
  int32_t n_prev, n_out, cat, relu_in, act, pad_;
} rumpy_qca_layer;
typedef struct {
  rumpy_qca_layer layers[4];
  int32_t nlayers, N, C, M, ntiles, nchunks;
  float inv_hw, scale;
  const float* pool;     /* fwd in: [N][ntiles][C] per-tile channel sums of the block's second conv */
  const float* attr;     /* [N][M] attribute vectors */
  float* acts;           /* [N][RUMPY_QCA_ACT_STRIDE] fwd out / bwd in */
  float* gate;           /* [N][C] fwd out / bwd in */
  const float* partial;  /* bwd in: [N][nchunks][C] from rumpy_ca_bwd_reduce */
  float* dpool;          /* bwd out: [N][C] d(mean) / HW, for rumpy_ca_bwd_apply */
  float* delta;          /* bwd out: [N][RUMPY_QCA_ACT_STRIDE] pre-activation gradients, layout of acts */
} rumpy_qca_args;
int rumpy_qca_gate_fwd(const rumpy_qca_args* a, void* stream);
int rumpy_qca_gate_bwd(const rumpy_qca_args* a, void* stream);
int rumpy_qca_bwd_params(const rumpy_qca_args* items_device, int32_t nitems, void* stream);

/* ---- degradation encoder of the blind-SR pipeline (frozen; rumpy/regression/models/contrastive_learning/encoding_models.py:5-55,
 * called by ContrastiveBlindSRPipeline.forward, rumpy/SISR/models/blur_kernel_blind_sr/contrastive_blind_sr.py:241-329):
 * nn.Conv2d(cin, cout, 3, stride, padding=1) + eval-mode BatchNorm2d (folded into w / bias by the host) + LeakyReLU, NHWC bf16 in and
 * out, cin and cout multiples of 64; and nn.AdaptiveAvgPool2d(1) -> fp32 [N, C]. */
typedef struct {
  const void* x;       /* [N,H,W,cin] bf16 */
  const void* w;       /* packed by rumpy_pack_weights (kind 0, shuffle 0): w_fwd */
  const float* bias;   /* [cout] */
  void* out;           /* [N,Ho,Wo,cout] bf16, Ho = (H-1)/stride + 1 */
  int32_t N, H, W, cin, cout, stride;
  float neg_slope;     /* LeakyReLU slope (1 = no activation, 0 = ReLU) */
  int32_t fmt;     /* RUMPY_FMT_BF16 | RUMPY_FMT_F16: element format of x, w and out (ABI 3: the TRAINING forward pass of the encoder stores its
                      filters, conv outputs and stage outputs as fp16 - the gradient of this BatchNorm + LeakyReLU network is 3-4 x closer to the
                      fp32 reference's than with 8-bit mantissas; gradients and the data-gradient convs stay bf16) */
  const void* w_lo;    /* ABI 4: NULL, or (fmt F16) the filter's rounding-residual image (rumpy_pack_weights fmt F16_RESIDUAL): swept into the same
                          accumulators, i.e. the conv runs on the unrounded filter (22 significant bits) */
  int32_t out_fmt;     /* 0: `out` has format `fmt`; RUMPY_FMT_F32: `out` is fp32 [N,Ho,Wo,cout] (the accumulators as they are) */
  int32_t pad_;
} rumpy_enc_conv_args;
int rumpy_enc_conv(const rumpy_enc_conv_args* a, void* stream);
int rumpy_enc_pool(const void* x, float* out, int32_t N, int32_t HW, int32_t C, int32_t fmt, void* stream);
/* training-mode nn.BatchNorm2d + LeakyReLU in place on a conv output (the reference runs its frozen encoder under net.train() inside
 * run_train: base_architecture.py:472): batch statistics over P = N*H*W values per channel (biased variance), running statistics updated
 * with `momentum` (unbiased variance) and the counter incremented, as torch does.  Deterministic. */
typedef struct {
  void* x;                       /* [P, C] bf16 (fmt: fp16), in and out */
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;   /* [C], updated; both NULL = leave untouched */
  int64_t* num_batches_tracked;  /* device scalar, += 1; may be NULL */
  float* partial;                /* scratch, >= rumpy_enc_bn_partial_floats(P, C) floats */
  float* scale_shift;            /* scratch [2, C] */
  int32_t P, C;
  float eps, momentum, neg_slope;
  int32_t fmt;                   /* RUMPY_FMT_*: format of x and of the output */
  int32_t x_fmt;                 /* ABI 4: 0 = x has format `fmt`; RUMPY_FMT_F32: x is fp32 [P, C] (rumpy_enc_bn_train_keep only: out of place) */
} rumpy_enc_bn_args;
int rumpy_enc_bn_train(const rumpy_enc_bn_args* a, void* stream);
int64_t rumpy_enc_bn_partial_floats(int32_t P, int32_t C);
/* The same, out of place and keeping what the backward pass needs: `out` [P, C] (a->fmt) = LeakyReLU(BN(x)), x untouched; `out_bf16` = NULL or
 * a second copy of the output rounded to bf16 (the weight-gradient kernels take bf16 operands: with fp16 stage outputs the next conv reads
 * `out`, its weight gradient `out_bf16`); `saved` [2, C] = batch mean, 1 / sqrt(biased variance + eps); a->scale_shift ([2, C]) must stay
 * alive until rumpy_enc_bn_bwd has run. */
int rumpy_enc_bn_train_keep(const rumpy_enc_bn_args* a, void* out, void* out_bf16, float* saved, void* stream);

/* ---- training the degradation encoder (MoCo / SupMoCo: rumpy/regression/models/contrastive_learning/moco.py:132-187, supmoco.py:52-128) ----
 * Backward of one BatchNorm2d(train) + LeakyReLU stage: from the gradient at the stage's output to the gradient at the conv output feeding
 * it, plus dgamma / dbeta.  The conv gradients are rumpy_enc_conv on the dgrad filter image (data) and rumpy_wgrad_grouped / _reduce
 * (weights; rumpy_head_wgrad for the first layer).  `up` = 2 writes pixel (oy, ox) at (2 oy, 2 ox) of an [N, Hz, Wz, C] grid the caller
 * zeroed once: the stride-1 data / weight gradient over that grid is the stride-2 convolution's.  Deterministic. */
typedef struct {
  const void* z;             /* [N*Ho*Wo, C] bf16 (fmt: fp16): the conv output the forward pass normalised (x of rumpy_enc_bn_train_keep) */
  const void* da;            /* [N*Ho*Wo, C] bf16 gradient at the LeakyReLU output ; NULL: */
  const float* dpool;        /* [N, C] gradient at the AdaptiveAvgPool2d(1) output behind the stage: da[n, p, c] = dpool[n, c] / (Ho*Wo) */
  const float* scale_shift;  /* [2, C] of the forward pass */
  const float* saved;        /* [2, C] of the forward pass */
  const float* gamma;        /* [C] */
  float* dgamma; float* dbeta;   /* [C], overwritten (x scale); may be NULL */
  void* dz;                  /* [N, Hz, Wz, C] bf16 */
  float* partial;            /* scratch, >= rumpy_enc_bn_partial_floats(N*Ho*Wo, C) floats */
  float* coef;               /* scratch [3, C] */
  int32_t N, Ho, Wo, C, up, Hz, Wz;
  float neg_slope, scale;
  int32_t fmt;               /* RUMPY_FMT_* (incl. RUMPY_FMT_F32): format of z (da and dz are bf16) */
} rumpy_enc_bn_bwd_args;
int rumpy_enc_bn_bwd(const rumpy_enc_bn_bwd_args* a, void* stream);
/* key-encoder momentum update over flat fp32 buffers: k = k * m + q * one_minus_m (two rounded products, then the sum: moco.py:71) */
int rumpy_ema(float* k, const float* q, int64_t n, float m, float one_minus_m, void* stream);

/* ---- the contrastive head of the encoder-training step (csrc/contrastive.hip; ABI 3): mlp head, L2 normalisation, MoCo / SupMoCo logits and
 * the softmax cross-entropy - rumpy/regression/models/contrastive_learning/encoding_models.py:43-55, moco.py:147-177, supmoco.py:75-119,
 * handlers.py:57 - forward and backward, exact fp32 (the reference's precision: these are cosines / 0.07), fixed summation orders.
 * rumpy_sgemm: C[m, n] = act(alpha * sum_k A(m, k) B(k, n) + bias[n]) [+ C], A(m, k) = A[m*sam + k*sak], B(k, n) = B[k*sbk + n*sbn], row
 * stride ldc; act = LeakyReLU(leaky_slope) (1 = none).  Products with few outputs and a long K (rumpy_sgemm_partial_floats(M, N, K) > 0)
 * run split over K and need `partial` of that many floats. */
typedef struct {
  const float* A; const float* B; float* C;
  const float* bias;     /* [N] or NULL */
  float* partial;        /* scratch for the split-K form, or NULL */
  int32_t M, N, K, ldc;
  int64_t sam, sak, sbk, sbn;
  float alpha, leaky_slope;
  int32_t accumulate;    /* 1: C += result */
  int32_t pad_;
} rumpy_sgemm_args;
int rumpy_sgemm(const rumpy_sgemm_args* a, void* stream);
int64_t rumpy_sgemm_partial_floats(int32_t M, int32_t N, int32_t K);
/* y = x / max(||x||_2, 1e-12) per row of [N, C] (nn.functional.normalize(dim=1)), inv[n] = the factor; backward dx = (dy - y (y . dy)) inv */
int rumpy_l2norm_rows(const float* x, float* y, float* inv, int32_t N, int32_t C, void* stream);
int rumpy_l2norm_rows_bwd(const float* dy, const float* y, const float* inv, float* dx, int32_t N, int32_t C, void* stream);
/* out[n * ldo] = scale * a[n, :] . b[n, :] */
int rumpy_rowdot(const float* a, const float* b, float* out, int32_t N, int32_t C, int32_t ldo, float scale, void* stream);
/* same[n, j] = (labels[n] == queue_labels[j]) as 0 / 1 floats ([N, K]; int64 labels), cnt[n] = row sum (supmoco.py:93-97) */
int rumpy_label_match(const void* labels, const void* queue_labels, float* same, float* cnt, int32_t N, int32_t K, void* stream);
/* v[n, :] = (sum_p k[n * P + p, :] + s[n, :]) * rscale / (P + cnt[n])  (s = cnt = NULL: MoCo) - the vector the positive logit is q . v with */
int rumpy_pos_vector(const float* k, const float* s, const float* cnt, float* v, int32_t N, int32_t P, int32_t C, float rscale, void* stream);
/* nn.CrossEntropyLoss(reduction='mean') over logits [N, M], int64 targets: lse [N], rowloss [N], loss [1]; backward dlogits = (softmax - onehot) * *gout / N */
int rumpy_ce_rows(const float* logits, const void* target, float* lse, float* rowloss, float* loss, int32_t N, int32_t M, void* stream);
int rumpy_ce_rows_bwd(const float* logits, const void* target, const float* lse, const float* gout, float* dlogits, int32_t N, int32_t M, void* stream);
/* out[c] = sum_n x[n, c] ; dh[i] *= (h[i] > 0 ? 1 : slope) with h the LeakyReLU output */
int rumpy_colsum(const float* x, float* out, int32_t N, int32_t C, void* stream);
int rumpy_lrelu_bwd(float* dh, const float* h, int64_t n, float slope, void* stream);
/* dst[n, :] += coef[n * ldcoef] * src[n, :] */
int rumpy_row_axpy(float* dst, const float* src, const float* coef, int32_t N, int32_t C, int32_t ldcoef, void* stream);
/* MoCo's enqueue (moco.py:74-89, supmoco.py:34-50) in one launch: queue[:, slots[i]] = keys[i * key_stride, :] (queue [C, K] fp32; one key per
 * query), queue_labels[slots[i]] = labels[i] (both NULL: no label track), then slots[i] = (slots[i] + n) % K and *queue_ptr = (*queue_ptr + n) % K
 * (int64 device words; queue_ptr may be NULL) */
int rumpy_moco_enqueue(float* queue, const float* keys, void* slots, void* queue_ptr, void* queue_labels, const void* labels,
                       int32_t n, int32_t key_stride, int32_t C, int32_t K, void* stream);

/* ---- optimizer: torch.optim.Adam semantics (base_architecture.py:93-95,437), flat fp32 buffers ---- */
typedef struct {
  float lr, beta1, beta2, eps;
  float bias_c1;        /* 1 - beta1^t */
  float sqrt_bias_c2;   /* sqrt(1 - beta2^t) */
  float grad_mult;      /* multiplies every gradient first (1, 1/world, or a clipping coefficient) */
  float max_norm;       /* > 0: clip_grad_norm_ semantics using *sumsq (base_architecture.py:434-435) */
} rumpy_adam_hyper;
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  const rumpy_adam_hyper* hyper;  /* DEVICE pointer: lets a captured graph replay with new hyper-parameters */
  const float* sumsq;             /* device scalar (sum of g^2) when hyper->max_norm > 0, else NULL */
  rumpy_adam_hyper hyper_value;   /* hyper == NULL: the hyper-parameters travel BY VALUE with the launch (eager steps: no staging copy) */
  const uint32_t* skip_if;        /* ABI 6: NULL, or a device word; when it is non-zero the launch leaves p, m and v untouched - the watchdog word of the step's
                                     persistent launches (rumpy_res_chain / rumpy_rcab_fwd status): a step whose hand-off timed out never reaches the weights */
} rumpy_adam_args;
int rumpy_adam_step(const rumpy_adam_args* a, void* stream);

/* ---- end-of-step housekeeping in two launches (finish.hip) ----
 * rumpy_finish_reduce: rumpy_wgrad_reduce over `items` + the tail conv's slab reduction (rumpy_tail_wgrad_reduce) + the head conv's
 * (the second half of rumpy_head_wgrad, which skips it when called with gw == NULL) in ONE launch; any part may be absent (NULL slabs / 0
 * items).  Same summation order as the separate entry points: bitwise the same gradients. */
typedef struct {
  const rumpy_reduce_item* items; int32_t nitems; int32_t pad_;
  const float* tail_slabs; int32_t tail_nslabs, tail_C; float tail_scale; int32_t pad2_; float* tail_gw; float* tail_gb;
  const float* head_slabs; int32_t head_nslabs, head_C, head_cout; float head_scale; float* head_gw; float* head_gb;
} rumpy_finish_reduce_args;
int rumpy_finish_reduce(const rumpy_finish_reduce_args* a, void* stream);
/* slabs rumpy_head_wgrad writes for an N x H x W image batch (= head_nslabs above) */
int rumpy_head_wgrad_slabs(int32_t N, int32_t H, int32_t W);

/* rumpy_adam_pack: rumpy_adam_step over the parameters named by `items` AND rumpy_pack_weights of the convs among them, in ONE launch
 * (one workgroup per item; every parameter must be covered by exactly one item).  Offsets are in floats from the start of the flat
 * buffers p / g / m / v.  kind 0: 16 output channels (quarter q of cout tile ct) x 32 input channels (half hf of cin chunk ch) x 9 taps of a
 * conv [cout, cin, 3, 3] at woff -> Adam + the vectors of both packed images that this set makes up; kind 1: n <= any plain values at
 * woff; kind 2: the n = cout <= 4096 biases of such a conv at woff -> Adam + b_packed; kind 3: the tail conv, weights [cout <= 4, 64, 3, 3]
 * at woff (n = cout * 576), bias at boff -> Adam + its two images (rumpy_pack_item kind 2). */
typedef struct {
  int32_t kind, woff, n, cout, cin, shuffle, ct, ch, q, hf, boff, pad_;
  void* w_fwd; void* w_dgrad; float* b_packed;
} rumpy_update_item;
typedef struct {
  const rumpy_update_item* items; int32_t nitems; int32_t pad_;
  float* p; const float* g; float* m; float* v;
  const rumpy_adam_hyper* hyper;  /* DEVICE pointer or NULL (then hyper_value), as rumpy_adam_args */
  const float* sumsq;
  rumpy_adam_hyper hyper_value;
  const uint32_t* skip_if;        /* ABI 6: as rumpy_adam_args.skip_if (the packed images stay as they are too) */
} rumpy_adam_pack_args;
int rumpy_adam_pack(const rumpy_adam_pack_args* a, void* stream);

/* out[0] = sum(g[i]^2) (deterministic two-pass; `partial` >= 1024 floats) */
typedef struct {
  const float* g;
  int64_t n;
  float* partial;
  float* out;
} rumpy_sumsq_args;
int rumpy_sumsq(const rumpy_sumsq_args* a, void* stream);

/* ---- eval post-processing (SISR/models/interface.py:103-124 + sr_tools/metrics.py:33-44,109-121) ----
 * rgb = clip(out,0,1); ycbcr = 'jpg' matrix of rgb; if ref (RGB in [0,1]): sse += (Y(rgb) - Y(clip(ref)))^2 */
typedef struct {
  const float* out;     /* [N,3,H,W] fp32 */
  const float* ref;     /* [N,3,H,W] fp32 or NULL */
  float* rgb;           /* [N,3,H,W] */
  float* ycbcr;         /* [N,3,H,W] */
  float* sse_partial;   /* >= 1024 floats or NULL */
  float* sse;           /* scalar: sum of squared Y error, or NULL */
  int32_t N, H, W;
} rumpy_eval_post_args;
int rumpy_eval_post(const rumpy_eval_post_args* a, void* stream);
/* image save (rumpy/sr_tools/visualization.py:31-62 safe_image_save): [N,C,H,W] fp32 -> [N,H,W,C] uint8 = trunc(clip(v * 255 / max_val, 0, 255)) */
int rumpy_to_uint8_hwc(const float* src, void* dst, int32_t N, int32_t C, int32_t H, int32_t W, float max_val, void* stream);

/* ---- device-side training-patch pipeline: crop + flips + transpose + uint8 -> float/255 ----
 * Replaces SuperResImages.__getitem__ / image_augment_crop (rumpy/sr_tools/data_handler.py:570-645),
 * random_flip_rotate / image_patch_selection / extract_image_patch (rumpy/image_tools/image_manipulation/
 * image_functions.py:245-362) and torchvision ToTensor (data_handler.py:472-486) for a batch of patches.
 * `images`: one device buffer of uint8 HWC images; an item names its LR / HR image by byte offset (HR = scale x LR size).
 * Augmentation order as the reference: A = transpose(vflip(hflip(I))) when the flags are set, patch = A[:, y:y+crop, x:x+crop];
 * the HR patch is A_hr[:, y*scale : (y+crop)*scale, x*scale : ...].  (y, x) index the AUGMENTED LR image. */
typedef struct {
  int64_t lr_off, hr_off;       /* byte offsets into `images` */
  int32_t lr_h, lr_w;           /* size of the stored (un-augmented) LR image */
  int32_t hflip, vflip, rot;    /* 0 / 1 */
  int32_t y, x;                 /* top-left corner of the LR patch in the augmented image */
  int32_t pad_;
} rumpy_patch_item;
typedef struct {
  const uint8_t* images;
  const rumpy_patch_item* items;   /* DEVICE array, N entries */
  float* out_lr;                   /* [N,C,crop,crop] fp32 */
  float* out_hr;                   /* [N,C,crop*scale,crop*scale] fp32, or NULL */
  int32_t N, C, crop, scale;
} rumpy_patch_args;
int rumpy_patch_gather(const rumpy_patch_args* a, void* stream);

/* ---- SSIM with the reference's settings: skimage.metrics.structural_similarity(data_range, gaussian_weights=True,
 * use_sample_covariance=False, sigma=1.5) as called by Metrics.run_ssim (rumpy/sr_tools/metrics.py:123-149); per plane [H,W] fp32 */
typedef struct {
  const float* a;       /* [P,H,W] */
  const float* b;       /* [P,H,W] reference planes */
  float* partial;       /* rumpy_ssim_partial_floats(P,H,W) floats of scratch */
  float* out;           /* [P] mean SSIM of each plane */
  int32_t P, H, W;
  float data_range;
} rumpy_ssim_args;
int rumpy_ssim(const rumpy_ssim_args* a, void* stream);
int64_t rumpy_ssim_partial_floats(int32_t P, int32_t H, int32_t W);

/* ---- direct fp32 convolution for the reference's "basic" models (SRCNN / VDSR: rumpy/SISR/models/basic/architectures.py:6-77, nn.Conv2d with
 * odd kernel sizes 3..11, padding k/2, 1..64 channels, on single-channel Y images): fp32 NCHW in and out, filters read in the reference's OIHW
 * layout in place.  Exact fp32 arithmetic (fma order differs from ATen's) - these layers have K = 81 / 1 output channel and are not MFMA-shaped. */
typedef struct {
  const float* x;        /* [N,Cin,H,W] */
  const float* w;        /* forward: [Cout,Cin,k,k].  transposed != 0 (data gradient): the reference filter [Cin,Cout,k,k], read flipped */
  const float* bias;     /* [Cout] or NULL */
  const float* mask;     /* optional [N,Cout,H,W]: the output is zeroed where mask <= 0 (ReLU backward, fused into the data gradient) */
  const float* res;      /* optional [N,Cout,H,W] added to the output (VDSR's global residual, architectures.py:77) */
  float* y;              /* [N,Cout,H,W] */
  int32_t N, Cin, Cout, H, W, k;
  int32_t relu;          /* epilogue max(.,0) (F.relu between the layers, architectures.py:50-51) */
  int32_t transposed;
} rumpy_dconv_args;
int rumpy_dconv(const rumpy_dconv_args* a, void* stream);

/* weight + bias gradient of such a layer: gw[o,i,ky,kx] = scale * sum_{n,y,x} dy[n,o,y,x] * x[n,i,y+ky-k/2,x+kx-k/2], gb[o] = scale * sum dy.
 * Deterministic: S pixel slabs of partial sums, reduced in a fixed order. */
typedef struct {
  const float* x;        /* [N,Cin,H,W] layer input */
  const float* dy;       /* [N,Cout,H,W] gradient at the layer's (pre-activation) output */
  float* partial;        /* rumpy_dconv_wgrad_partial_floats(...) floats of scratch */
  float* gw;             /* [Cout,Cin,k,k] */
  float* gb;             /* [Cout] or NULL */
  int32_t N, Cin, Cout, H, W, k;
  float scale;
} rumpy_dconv_wgrad_args;
int rumpy_dconv_wgrad(const rumpy_dconv_wgrad_args* a, void* stream);
int64_t rumpy_dconv_wgrad_partial_floats(int32_t N, int32_t Cin, int32_t Cout, int32_t H, int32_t W, int32_t k);

/* nn.MSELoss (mean) and its gradient in one pass: loss = mean((out - target)^2), grad = 2 (out - target) / numel (basic/handlers.py:14) */
typedef struct {
  const float* out;      /* [n] */
  const float* target;   /* [n] */
  float* grad;           /* [n] or NULL */
  float* partial;        /* 1024 floats of scratch */
  float* loss;           /* [1] */
  int64_t n;
} rumpy_mse_args;
int rumpy_mse_loss(const rumpy_mse_args* a, void* stream);

/* a launch list: fn = address of any `int fn(const <args>*, void* stream)` entry point of this library, args = its argument block */
typedef struct { const void* fn; const void* args; } rumpy_op;
int rumpy_run_list(const rumpy_op* ops, int32_t n, void* stream);

/* (the timing probe of bench.py and the diagnostics of tests / tools are declared in include/rumpy_amd_debug.h: measurement hooks of the
 * same library, not part of the drop-in boundary) */

#ifdef __cplusplus
}
#endif
#endif

"""Host-side executor of the SR conv hot path on MI355X: turns an EDSR / RCAN module tree into a static list of
C-ABI kernel launches (forward, backward, weight gradients, optimizer) over preallocated NHWC-bf16 HBM buffers.

PyTorch is used only for device memory, streams and (elsewhere) torch.distributed; every arithmetic step is a
hand-written HIP kernel reached through ``rumpy_amd._lib`` (include/rumpy_amd.h).  Reference behaviour being
reproduced (paths relative to /root/reference/rumpy):
  SISR/models/advanced/architectures.py:198-241 (EDSR.forward), :140-176 (RCAN.forward), :60-84 (RCAB),
  :107-124 (ResidualGroup), :24-44 (CALayer); SISR/models/advanced/common.py:23-75 (Upsampler, ResBlock);
  shared_framework/models/base_architecture.py:425-485 (loss.backward + Adam step).
"""
import bisect
import collections
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib as L

BF16 = torch.bfloat16


def _ptr(t):
    return None if t is None else t.data_ptr()


class ConvLayer:
    """One nn.Conv2d(k=3, p=1) of the net: fp32 OIHW master parameters + packed bf16 images."""

    def __init__(self, name, weight, bias, kind, shuffle=False):
        self.name, self.weight, self.bias, self.kind, self.shuffle = name, weight, bias, kind, shuffle
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.w_fwd = self.w_dgrad = self.b_packed = None
        self.w_fwd_h = None       # fp16 forward image (evaluation plans), allocated with the first such plan
        self.gw = self.gb = None  # views into the flat gradient buffer


class CALayerParams:
    """RCAN CALayer 1x1 convs (architectures.py:34-39)."""

    def __init__(self, name, w1, b1, w2, b2):
        self.name, self.w1, self.b1, self.w2, self.b2 = name, w1, b1, w2, b2
        self.C, self.Cr = w2.shape[0], w1.shape[0]
        self.gw1 = self.gb1 = self.gw2 = self.gb2 = None


class StyledCAParams:
    """QCALayer in one of the styles whose gate MLP also takes the attribute vector (attention_manipulators/architectures.py:41-136:
    'max_concat', 'mini_concat', 'extended_attention', 'softmax'): a list of layers
    (w [n_out, n_in], b, gw, gb, cat: attributes appended to the input, relu_in: ReLU on the input vector, act: 0 none / 1 ReLU / 2 sigmoid /
    3 sigmoid + softmax) for rumpy_qca_gate_fwd / _bwd (csrc/qca_style.hip)."""

    def __init__(self, name, layers, num_metadata):
        self.name, self.layers, self.M = name, layers, num_metadata
        self.C = layers[-1]['w'].shape[0]
        self.gen = True


class QLayerParams(CALayerParams):
    """Meta-attention q-layer (ParaCALayer, attention_manipulators/q_layer.py:5-45, two FC layers): w1 [Hq,M], w2 [C,Hq]."""

    def __init__(self, name, w1, b1, w2, b2):
        super().__init__(name, w1, b1, w2, b2)
        self.Hq, self.M = w1.shape[0], w1.shape[1]


class QLayerNParams:
    """The same q-layer with ParaCALayer's `num_layers` other than 2 (q_layer.py:13,22-41): layers = [dict(w [n_out, n_in], b, gw, gb)], ReLU behind
    every layer but the last - run by the general-depth launches rumpy_q_mlpn_* (csrc/ca.hip)."""

    def __init__(self, name, layers):
        self.name, self.layers = name, layers
        self.M, self.C = layers[0]['w'].shape[1], layers[-1]['w'].shape[0]
        self.widths = [self.M] + [l['w'].shape[0] for l in layers]
        self.hsum = sum(self.widths[1:-1])
        if len(layers) > L.QN_MAX_LAYERS or max(self.widths) > 256 or sum(self.widths[1:]) > 448:
            raise RuntimeError('rumpy_amd: q-layer %s (widths %s) is beyond the HIP path (<= %d layers of <= 256 units, <= 448 in all)' % (name, self.widths, L.QN_MAX_LAYERS))


class NetSpec:
    """Topology handed over by the architecture module.

    body items: ('resblock', conv1, conv2, res_scale) | ('rcab', conv1, conv2, ca[, q]) | ('group', [items], conv)
    (q: QLayerParams of a QRCAB's meta-attention node or None; num_metadata > 0 makes forward() expect a metadata matrix;
    modulate: QCALayer style 'modulate' - the [N, 64] metadata matrix itself multiplies every block's attention gate)
    """

    def __init__(self, head, body, body_conv, ups, tail, scale, num_metadata=0, modulate=False):
        self.head, self.body, self.body_conv, self.ups, self.tail, self.scale = head, body, body_conv, ups, tail, scale
        self.num_metadata = num_metadata
        self.modulate = modulate

    def convs(self):
        out = [self.head]

        def walk(items):
            for it in items:
                if it[0] in ('resblock', 'rcab'):
                    out.extend([it[1], it[2]])
                else:
                    walk(it[1])
                    out.append(it[2])
        walk(self.body)
        out.append(self.body_conv)
        out.extend(self.ups)
        out.append(self.tail)
        return out

    def cas(self):
        out = []

        def walk(items):
            for it in items:
                if it[0] == 'rcab':
                    out.append(it[3])
                elif it[0] == 'group':
                    walk(it[1])
        walk(self.body)
        return out


class _Plan:
    """Static launch lists + buffers for one (N, H, W, train) shape."""

    def __init__(self):
        self.fwd, self.bwd, self.keep = [], [], []
        self.x_in = self.target = self.out = self.loss = self.dy4 = None
        self.gout_stage = None


def cut_wgrad_shares(units, nsh, job_cost=4.0):
    """Cut the weight-gradient work of one launch into `nsh` shares, one per workgroup (= CU).  units: [(tiles, quad_key or None, q)] in layer order; returns per
    unit its tile ranges [(t0, t1, share, prio)] (prio 0 jobs run first inside a share).  A range is a JOB of the kernel: tiles streamed through the ring, then
    a K-half exchange and a 147 KB slab - a fixed cost of `job_cost` tiles (round 6, measured: a second job per workgroup = +8 us = 4 tiles of 1.9 us,
    profiles/r06_wgrad.txt).

    Plain units (quad_key None) form ONE tile sequence; a share may end one layer and begin the next.  The shares are cut to equal COST, tiles + job_cost per
    job (round 6; equal tile counts until then: the launch ended with the workgroups that crossed a layer boundary, 8 us behind the others).

    (round 6) ALIGNED units: the four output-channel tiles q = 0 .. 3 of an upsampler conv (64 -> 256 + pixel shuffle; same quad_key) read the SAME x tiles.  Cut
    as part of the one sequence they ran on workgroups of different XCDs a third of a launch apart and x came out of HBM four times (170 MB of the 1,020 MB the
    launch of a x4 net reads).  Here nq QUADS of workgroups - four of ONE XCD each (workgroup w runs on XCD w % 8) - take the aligned convs' tile sequence cut
    into nq chunks, output-channel tile q on the q-th workgroup of the quad: the four start together, run in step, and x is read once into that XCD's L2.  The
    other workgroups share the plain units; nothing is cut finer than before."""
    T_all = sum(u[0] for u in units)
    by_key = {}
    for i, (nt, key, q) in enumerate(units):
        if key is not None:
            by_key.setdefault(key, []).append(i)
    quads = [sorted(v, key=lambda i: units[i][2]) for v in by_key.values()
             if len(v) == 4 and sorted(units[i][2] for i in v) == [0, 1, 2, 3] and len(set(units[i][0] for i in v)) == 1]
    TA = sum(units[quad[0]][0] for quad in quads)                     # aligned tiles per output-channel tile
    avg = T_all / nsh + job_cost
    quad_share = lambda j, q: (j % 8) + 8 * ((j // 8) * 4 + q)
    MIN_JOB = max(1, min(int(job_cost), T_all // (4 * nsh)))         # no job shorter than its own fixed cost - unless the launch is that small

    def layout(nq):
        """the cut with nq quads of workgroups on the aligned units (0: none aligned) -> (ranges per unit, cost of the most loaded share)"""
        out = [None] * len(units)
        use = quads if nq else []
        aligned = set(i for quad in use for i in quad)
        a_cost = {}
        if use:
            def chunks(CA):                                # the aligned sequence in chunks of cost <= CA (tiles + job_cost per conv a chunk touches)
                res, c, cap = [[] for _ in use], 0, CA
                for n, quad in enumerate(use):
                    nt, t = units[quad[0]][0], 0
                    while t < nt:
                        if cap < job_cost + min(MIN_JOB, nt - t):
                            c, cap = c + 1, CA
                        take = min(nt - t, max(1, int(cap - job_cost)))
                        if nt - t - take < MIN_JOB:
                            take = nt - t
                        res[n].append((t, t + take, c))
                        cap -= job_cost + take
                        t += take
                return res, c + 1
            lo, hi = 0.0, float(TA + job_cost * len(use) + 1)
            for _ in range(40):
                mid = 0.5 * (lo + hi)
                lo, hi = (lo, mid) if chunks(mid)[1] <= nq else (mid, hi)
            for quad, rs in zip(use, chunks(hi)[0]):
                for q, i in enumerate(quad):
                    out[i] = [(t, t1, quad_share(c, q), 0) for t, t1, c in rs]
                    for t, t1, c in rs:
                        a_cost[quad_share(c, q)] = a_cost.get(quad_share(c, q), 0.0) + t1 - t + job_cost
        # the plain units: every share is filled up to the cost C - the smallest C all the work fits under (bisection over a greedy fill).  A share with
        # aligned work takes plain work only if a quarter of the average is missing; no job of less than MIN_JOB tiles unless that is all a layer has left
        rest = [i for i in range(len(units)) if i not in aligned]

        def fill(C, commit):
            it, t = 0, 0                                   # next plain tile: unit rest[it], tile t
            res = [[] for _ in rest]
            for k in range(nsh):
                cap = C - a_cost.get(k, 0.0)
                if k in a_cost and cap < 0.25 * avg:
                    continue
                while it < len(rest) and cap >= job_cost + min(MIN_JOB, units[rest[it]][0] - t):
                    left = units[rest[it]][0] - t
                    take = min(left, int(cap - job_cost))
                    if left - take < MIN_JOB:
                        take = left                        # (a rest of a layer too small for a job of its own goes along)
                    res[it].append((t, t + take, k, 1))
                    cap -= job_cost + take
                    t += take
                    if t >= units[rest[it]][0]:
                        it, t = it + 1, 0
            if it < len(rest):
                return False
            if commit:
                for i, rs in zip(rest, res):
                    out[i] = rs
            return True
        C = 0.0
        if rest:
            lo, hi = 0.0, 2.0 * (avg + job_cost) + max(units[i][0] for i in rest)
            while not fill(hi, False):
                hi *= 2.0
            for _ in range(40):
                mid = 0.5 * (lo + hi)
                lo, hi = (lo, mid) if fill(mid, False) else (mid, hi)
            fill(hi, True)
            C = hi
        cost = [0.0] * nsh
        for rs in out:
            for t, t1, k, prio in rs:
                cost[k] += t1 - t + job_cost
        return out, max(cost)
    cands = [0]
    if quads and nsh % 32 == 0:
        ideal = (TA + job_cost * len(quads)) / avg                    # quads that would carry exactly the average cost
        top = min(nsh // 4 - (1 if len(units) > 4 * len(quads) else 0), TA)      # (a quad of workgroups less when there is plain work as well)
        cands = sorted(set(min(top, n) for n in list(range(int(ideal) - 1, int(ideal) + 3)) + [int(ideal * f) for f in (1.25, 1.5, 2.0)] if n >= 1)) if top >= 1 else [0]
    best = min((layout(nq) for nq in cands), key=lambda oc: oc[1])     # the launch ends with its most loaded workgroup
    return best[0]


class SREngine:
    def __init__(self, spec, device, wgrad_pixels_per_job=8192):
        self.spec, self.device = spec, device
        self.lib = L.lib()
        self.cus = self.lib.rumpy_device_cus()
        self.plans = {}
        self._tables = {}      # id(launch list) -> (list, ctypes table, its address, length)
        self.wgrad_pixels_per_job = int(os.environ.get('RUMPY_WGRAD_PIXELS', wgrad_pixels_per_job))    # env: A/B runs of the job size
        # Weight-gradient work as equal SHARES of the concatenated tile sequence of all 64-channel layers, one share per CU (rumpy_wgrad_shares:
        # a workgroup runs its share's jobs one after the other; 477 equal jobs on 256 CUs took two rounds of 64 tiles where the average is
        # 119, and left one slab per job to reduce: 288 -> 262 us for the launch, 34 -> 18 us for the reduction).  RUMPY_WGRAD_JOBS=1: fixed-size
        # jobs, one workgroup each (A/B).  Every tile costs the same (weighting the gathered upsampler layers differently was slower either
        # way).  two_phase: the shares are cut separately for the layers of the upper and of the lower part of the gradient buffer, so that the
        # data-parallel form (upper part first, its all-reduce under the rest) and the one-launch form run the SAME jobs: bitwise equal
        # gradients.  Set by the first backward pass with a gradient-ready hook / RUMPY_WGRAD_AB=1.
        self.wgrad_shares = os.environ.get('RUMPY_WGRAD_JOBS') != '1'
        self.wgrad_align = os.environ.get('RUMPY_WGRAD_ALIGN', '1') != '0'       # the output-channel tiles of an upsampler conv on one XCD, in step (_emit_wgrad)
        self.wgrad_two_phase = os.environ.get('RUMPY_WGRAD_AB') == '1'
        self.use_block_kernel = os.environ.get('RUMPY_NO_BLOCK') != '1'    # residual blocks in one launch (conv_block.hip)
        self.batch_by_pointer = os.environ.get('RUMPY_BATCH_COPY') != '1'  # captured training step reads the caller's x / target through a pointer table; =1: A/B, copies into fixed buffers
        self.block_any_width = os.environ.get('RUMPY_BLOCK_W48') != '1'    # ... also for images wider than one strip (column tiles); =1: A/B, two launches per block there
        self.use_rcab_kernel = os.environ.get('RUMPY_NO_RCAB') != '1'      # channel-attention blocks in one launch (conv_rcab.hip)
        # ... and which one-launch form (round 5): 'lazy' = conv_rcab2.hip, the gate applied by the launch that consumes a block's output - no exchange
        # between workgroups, any image size; 'xchg' = conv_rcab.hip, pool sums exchanged inside the launch (A/B; precision 'fp8' runs this form)
        # Default 'auto' = what measured faster per shape (profiles/r05_rcab_forms.txt): 'xchg' while a strip spans the image (W <= 48) and all strips of an
        # image are resident (32 x 48 x 48: 3.02 k against 2.83 k patches/s - the lazy forward launch reads two halo tiles), 'lazy' for every wider image
        # (8 crops of 64 x 64, div2k/rcan.toml: +6.4 %; 96 x 96: +2.6 %; whole-image evaluation: one launch per block instead of two, +2.8 %) and wherever
        # the exchange's residency condition does not hold.
        self.rcab_form = os.environ.get('RUMPY_RCAB_FORM', 'auto')
        if self.rcab_form not in ('auto', 'lazy', 'xchg'):
            raise RuntimeError("rumpy_amd: RUMPY_RCAB_FORM is 'auto', 'lazy' or 'xchg' (got %r)" % self.rcab_form)
        self.use_mask_bytes = os.environ.get('RUMPY_NO_MASKBITS') != '1'   # ReLU mask of the block kernels as one byte per 8 channels
        # runs of consecutive residual-block launches as ONE persistent launch with the halo rows handed over through the XCD's L2 (conv_chain.hip, round 5;
        # bitwise the per-block launches; needs every strip co-resident: N * ceil(H/6) <= CUs, W <= 48); RUMPY_NO_CHAIN=1: one launch per block (A/B)
        self.use_chain = os.environ.get('RUMPY_NO_CHAIN') != '1'
        # the tail conv's data gradient inside the last upsampler stage's data-gradient launch (rumpy_conv4d_tail; RUMPY_NO_TAIL_FUSE=1: two launches, A/B)
        self.fuse_tail_dgrad = os.environ.get('RUMPY_NO_TAIL_FUSE') != '1'
        self.chain_edge = os.environ.get('RUMPY_NO_CHAIN_EDGE') != '1'     # the body-end conv (and its data gradient) inside the chain launch; =1: its own launch (A/B)
        self.chain_force_sc1 = os.environ.get('RUMPY_CHAIN_SC1') == '1'       # A/B: every hand-off of the chain through the memory side
        # Evaluation plans store activations and filters as IEEE fp16 (same MFMA rate and bytes as bf16, 11 instead of 8 significant bits):
        # bf16 storage alone costs a >= 30 dB model 0.02-0.03 dB of Y-PSNR against the fp32 reference (fixtures G17 / G18).
        # Training stays bf16 (gradient range).  An output that is not finite (fp16 overflow) switches the engine back to bf16 for good.
        self.eval_fmt = L.FMT_BF16 if os.environ.get('RUMPY_EVAL_BF16') == '1' else L.FMT_F16

        def styled(items):
            return any((it[0] == 'rcab' and getattr(it[3], 'gen', False)) or (it[0] == 'group' and styled(it[1])) for it in items)
        if styled(spec.body):
            self.eval_fmt = L.FMT_BF16      # the separate channel-attention launches of the styled QCALayers are bf16 only
        self.max_eval_plans = max(1, int(os.environ.get('RUMPY_EVAL_PLANS', '4')))     # LRU bound on cached evaluation plans (one per image size)
        # end-of-step housekeeping as two launches (csrc/finish.hip): one reduction launch for all slab kinds, Adam + re-pack in one
        self.use_finish = os.environ.get('RUMPY_NO_FINISH') != '1'
        self.update_items = None   # device table of rumpy_update_item (build_update_table), n items
        self.pack_gen = 0          # bumped by every repack(); the fp16 images follow lazily (h_gen)
        self.h_gen = -1
        self._pack_items_h = None
        self.feats = spec.head.cout
        if self.feats not in (64, 128, 192, 256):
            raise RuntimeError('rumpy_amd: the HIP path is built for n_feats = 64, 128, 192 and 256 (got %d); other widths are not '
                               'implemented and there is no fallback' % self.feats)
        # wide = more than 64 features, up to the reference's shipped EDSR width (div2k/edsr.toml: 256): every 3x3 conv runs the 2 / 3 / 4
        # input-chunk form of rumpy_conv3x3 (one launch per layer, epilogues fused), the tail its fp32 VALU form; the one-launch
        # residual-block kernels are 64-feature kernels.
        self.wide = self.feats != 64
        # an upsampler stage that is not "64 features, PixelShuffle(2)" (x3: conv F -> 9F, or any stage of a wide net): the conv writes its
        # natural channel order, rumpy_pixel_shuffle permutes (forward) / un-permutes the gradient (backward): csrc/wide.hip
        self.generic_up = self.wide or any(cv.cout != 4 * cv.cin for cv in spec.ups)
        # fp16 evaluation plans of the 64-feature nets: upsampler filters as image + rounding-residual image too (two conv launches per stage)
        self.eval_up_residual = os.environ.get('RUMPY_EVAL_UP_RESIDUAL', '1') == '1'
        self.eval_up_fused = os.environ.get('RUMPY_EVAL_UP_FUSED', '1') == '1'      # ... in ONE launch per stage (conv_up.hip sweeps both images); =0: A/B, two launches
        # evaluation status words (non-finite output, strip-exchange watchdog): read back synchronously after every pass, or - eval_defer,
        # set by run_eval(keep_on_device=True) around its pass - staged into pinned memory and examined at the next pass / check_eval()
        self.eval_defer = False
        self._flag_host = self._flag_pending = None
        if spec.head.cin > 4 or spec.tail.cout > 4:
            raise RuntimeError('rumpy_amd: image channels must be <= 4')
        if spec.tail.cin != self.feats:
            raise RuntimeError('rumpy_amd: tail conv needs %d input features (got %d)' % (self.feats, spec.tail.cin))
        if self.wide and (spec.cas() or any(it[0] != 'resblock' for it in spec.body)):
            # RCAN wider than 64 features (round 5): every conv on the Cin = 128 / 256 form of rumpy_conv3x3, the channel attention as its separate
            # launches (pool sums from the conv's epilogue, rumpy_ca_fwd_fused, rumpy_ca_bwd_reduce / _fused: C <= 256, a multiple of 8; 192 since round 6). The
            # one-launch RCAB kernels and the q-layer launches are 64-feature kernels.
            plain_ca = all(not getattr(ca, 'gen', False) for ca in spec.cas())
            if not plain_ca:
                raise RuntimeError('rumpy_amd: n_feats > 64 is built for EDSR and for RCAN / QRCAN (styles "standard", "modulate") at 128 / 192 / 256 '
                                   'features; styled channel attention is a 64-feature kernel')
        for cv in spec.convs():
            if cv.kind == 'main' and (cv.cin not in (64, 128, 192, 256) or cv.cout % 64):
                raise RuntimeError('rumpy_amd: conv %s %d->%d unsupported (Cin must be 64, 128, 192 or 256, Cout a multiple of 64)'
                                   % (cv.name, cv.cin, cv.cout))
        for cv in spec.ups:
            r2 = cv.cout // cv.cin
            if cv.cout != r2 * cv.cin or r2 not in (4, 9):
                raise RuntimeError('rumpy_amd: upsampler conv %s %d->%d is neither a x2 nor a x3 stage' % (cv.name, cv.cin, cv.cout))
            if self.generic_up:
                cv.shuffle = False          # natural channel order; the permutation is its own pass
        self._alloc_packed()
        self.packed_version = None
        # precision 'fp8' (BASELINE config 5; opt-in, enable_fp8): the one-launch residual-block kernels of TRAINING plans run both sweeps on the
        # block-scaled fp8 MFMA (csrc/conv_block_fp8.hip).  Weight gradients, head / tail / upsampler / body-end convs, the optimizer and every
        # evaluation plan are unchanged.
        self.fp8 = False
        self.f8_gen = -1
        self._f8_items = None
        # the watchdog word of the LAST training pass's plan (or None: no launch of it waits for another workgroup); the fused optimizer launch
        # reads it on the device (rumpy_adam_pack_args.skip_if): a step whose hand-off timed out never reaches the weights (round 6)
        self.step_status = None

    # ------------------------------------------------------------------ precision 'fp8'
    def enable_fp8(self):
        """from now on training plans run their residual blocks on the fp8 matrix instruction (existing training plans are dropped)"""
        if self.fp8:
            return
        if self.wide:
            raise RuntimeError("rumpy_amd: precision 'fp8' is built for the 64-feature one-launch block kernels")
        self.fp8 = True
        for k in [k for k in self.plans if k[3]]:
            old = self.plans.pop(k)
            for ops in (old.fwd, old.bwd):
                self._tables.pop(id(ops), None)
        convs = []

        def walk(items):
            for it in items:
                if it[0] == 'resblock' or (it[0] == 'rcab' and not getattr(it[3], 'gen', False)):
                    convs.extend([it[1], it[2]])
                elif it[0] == 'group':
                    walk(it[1])
        walk(self.spec.body)
        # mixed-precision policy (round 6): blocks that stay on the bf16 kernels (no fp8 filter images -> the plan emits their bf16 launches)
        nblk = len(convs) // 2
        self.f8_bf16_blocks = self._fp8_bf16_blocks(nblk)
        convs = [cv for i, cv in enumerate(convs) if (i // 2) not in self.f8_bf16_blocks]
        self.f8_wscale = torch.full((max(1, len(convs)),), 127, dtype=torch.int32, device=self.device)
        items = []
        for i, cv in enumerate(convs):
            cv.w8_fwd = torch.empty(L.FP8_IMAGE_BYTES, dtype=torch.uint8, device=self.device)
            cv.w8_dgrad = torch.empty(L.FP8_IMAGE_BYTES, dtype=torch.uint8, device=self.device)
            cv.w8_scale = self.f8_wscale[i:i + 1]
            items.append(L.Fp8PackItem(w=_ptr(cv.weight), img_fwd=_ptr(cv.w8_fwd), img_dgrad=_ptr(cv.w8_dgrad), exponent=_ptr(cv.w8_scale)))
        self._f8_convs = convs
        if items:
            self._f8_items_host = (L.Fp8PackItem * len(items))(*items)
            self._f8_items = (self._to_device_bytes(self._f8_items_host), len(items))

    # default mixed-precision policy of precision='fp8' (round 6, profiles/r06_fp8_block_ablation.txt): see _fp8_bf16_blocks
    FP8_POLICY_DEFAULT = 'none'

    def _fp8_bf16_blocks(self, nblk):
        """ordinals (walk order of the body) of the residual blocks / RCABs that precision='fp8' keeps on the bf16 kernels.
        RUMPY_FP8_BF16_BLOCKS overrides the default policy: 'none', a comma list of ordinals ('0,20,199'), 'first:K' / 'last:K' (the first / last K
        blocks of the body) or 'every:K' (every K-th block: with K = n_resblocks each group's first RCAB)."""
        pol = os.environ.get('RUMPY_FP8_BF16_BLOCKS', self.FP8_POLICY_DEFAULT).strip()
        if pol in ('', 'none'):
            return frozenset()
        try:
            if ':' in pol:
                kind, k = pol.split(':', 1)
                k = int(k)
                if k <= 0:
                    raise ValueError
                if kind == 'first':
                    return frozenset(range(min(k, nblk)))
                if kind == 'last':
                    return frozenset(range(max(0, nblk - k), nblk))
                if kind == 'every':
                    return frozenset(range(0, nblk, k))
                raise ValueError
            return frozenset(b for b in (int(t) for t in pol.split(',')) if 0 <= b < nblk)
        except ValueError:
            raise RuntimeError("rumpy_amd: RUMPY_FP8_BF16_BLOCKS is 'none', 'first:K', 'last:K', 'every:K' or a comma list of block ordinals (got %r)" % pol)

    def _repack_f8(self, stream):
        """bring the fp8 filter images (and their per-conv scale exponents) up to date with the master weights; no-op when nothing changed"""
        if self._f8_items is not None and self.f8_gen != self.pack_gen:
            L.check(self.lib.rumpy_fp8_pack(_ptr(self._f8_items[0]), self._f8_items[1], stream), 'rumpy_fp8_pack')
            self.f8_gen = self.pack_gen

    def _f8_site(self, plan, which):
        """one record per fp8 launch of a plan: exponents of its two image tensors, then one amax pair per (workgroup, row half) -> dict of
        the launch's f8_site / f8_entries arguments"""
        key = 'f8_' + which
        buf = getattr(plan, key)
        entries = int(self.lib.rumpy_fp8_site_entries(plan.N, plan.H, plan.W))
        if buf is None:
            n = max(1, len(self._f8_convs) // 2)
            buf = torch.zeros(n, L.FP8_SITE_HEAD + 2 * entries, dtype=torch.int32, device=self.device)
            buf[:, 0:2] = 127
            buf[:, 2] = entries
            setattr(plan, key, buf)
        i = getattr(plan, key + '_n')
        setattr(plan, key + '_n', i + 1)
        return dict(f8_site=buf[i].data_ptr(), f8_entries=entries)

    def _f8_begin(self, plan, which, ops, stream):
        """in front of a pass that contains fp8 launches: the first time the pass runs once to measure (every amax is taken from the values as
        they are, whatever the scales), then - every time - last pass's amax becomes this pass's scale exponents (rumpy_fp8_rotate)"""
        n = getattr(plan, 'f8_' + which + '_n')
        if not n:
            return
        buf = getattr(plan, 'f8_' + which)
        if not getattr(plan, 'f8_' + which + '_cal'):
            # TWO measuring passes: with every exponent at 127 the first one converts at scale 1 - gradients that arrive mean-reduced
            # (1e-7: the generic loss path) flush to zero in e5m2, the intermediate image of a block is then exactly 0 and its amax with it.  The
            # skip connection carries the block's input through unconverted, so pass one leaves every site's INPUT amax right; with those exponents
            # rotated in, pass two measures the intermediate images from values that were converted at their proper scale.
            setattr(plan, 'f8_' + which + '_cal', True)
            for rep in range(2):
                self._run(ops, stream)
                self._advance_epoch(plan, stream)
                if rep == 0:
                    L.check(self.lib.rumpy_fp8_rotate(buf.data_ptr(), n, buf.shape[1], stream), 'rumpy_fp8_rotate')
        L.check(self.lib.rumpy_fp8_rotate(buf.data_ptr(), n, buf.shape[1], stream), 'rumpy_fp8_rotate')

    # ------------------------------------------------------------------ packed filters
    def _alloc_packed(self):
        dev = self.device
        items = []
        for cv in self.spec.convs():
            if cv.kind == 'head':
                continue
            if cv.kind == 'main':
                n = cv.cout * cv.cin * 9
                cv.w_fwd = torch.empty(n, dtype=BF16, device=dev)
                cv.w_dgrad = torch.empty(n, dtype=BF16, device=dev)
                cv.b_packed = torch.empty(cv.cout, dtype=torch.float32, device=dev)
                items.append(L.PackItem(w=_ptr(cv.weight), b=_ptr(cv.bias), w_fwd=_ptr(cv.w_fwd), w_dgrad=_ptr(cv.w_dgrad),
                                        b_packed=_ptr(cv.b_packed), cout=cv.cout, cin=cv.cin, kind=0,
                                        shuffle=1 if cv.shuffle else 0))
            elif self.wide:   # tail of a wide net: rumpy_tail_*_wide read the fp32 master filter
                continue
            else:  # tail
                cv.w_fwd = torch.empty(18 * 64 * 8, dtype=BF16, device=dev)
                cv.w_dgrad = torch.empty(4 * 2 * 64 * 8, dtype=BF16, device=dev)
                items.append(L.PackItem(w=_ptr(cv.weight), b=_ptr(cv.bias), w_fwd=_ptr(cv.w_fwd), w_dgrad=_ptr(cv.w_dgrad),
                                        b_packed=None, cout=cv.cout, cin=cv.cin, kind=2, shuffle=0))
        self._pack_items_host = (L.PackItem * len(items))(*items)
        self._pack_items = self._to_device_bytes(self._pack_items_host)
        self._n_pack = len(items)

    def _to_device_bytes(self, ctypes_array):
        raw = np.frombuffer(bytes(ctypes_array), dtype=np.uint8).copy()
        return torch.from_numpy(raw).to(self.device)

    def repack(self, stream=None):
        """fp32 OIHW master weights -> bf16 MFMA-fragment images (after every optimizer step / weight load)."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.rumpy_pack_weights(_ptr(self._pack_items), self._n_pack, s), 'rumpy_pack_weights')
        self.pack_gen += 1

    def build_update_table(self, flat_p, params, offsets):
        """Item table of rumpy_adam_pack over the flat parameter buffer: every parameter exactly once.  64-multiple convs become sets of
        16 output x 32 input channels (kind 0) + their bias (kind 2), the tail conv one item (kind 3), everything else plain ranges (kind 1,
        neighbours merged: the four tensors of a channel-attention block are one range)."""
        base = flat_p.data_ptr()
        off_of = lambda t: (t.data_ptr() - base) // 4
        roles = {}
        for cv in self.spec.convs():
            if cv.kind == 'main':
                roles[off_of(cv.weight)] = ('w', cv)
                roles[off_of(cv.bias)] = ('b', cv)
            elif cv.kind == 'tail' and not self.wide:      # a wide net's tail has no packed image: its weight and bias are plain ranges
                roles[off_of(cv.weight)] = ('tw', cv)
                roles[off_of(cv.bias)] = ('tb', cv)
        items, plain = [], []
        for p, off in zip(params, offsets):
            role = roles.get(off)
            if role is None:
                if plain and plain[-1][0] + plain[-1][1] == off:
                    plain[-1][1] += p.numel()
                else:
                    plain.append([off, p.numel()])
                continue
            kind, cv = role
            if kind == 'w':
                for ct in range(cv.cout // 64):
                    for ch in range(cv.cin // 64):
                        for q in range(4):
                            for hf in range(2):
                                items.append(L.UpdateItem(kind=0, woff=off, n=16 * 288, cout=cv.cout, cin=cv.cin, shuffle=1 if cv.shuffle else 0, ct=ct,
                                                          ch=ch, q=q, hf=hf, w_fwd=_ptr(cv.w_fwd), w_dgrad=_ptr(cv.w_dgrad)))
            elif kind == 'b':
                if cv.cout > 4096:
                    raise RuntimeError('rumpy_amd: conv %s has more than 4096 output channels' % cv.name)
                items.append(L.UpdateItem(kind=2, woff=off, n=cv.cout, cout=cv.cout, cin=cv.cin, shuffle=1 if cv.shuffle else 0, b_packed=_ptr(cv.b_packed)))
            elif kind == 'tw':
                items.append(L.UpdateItem(kind=3, woff=off, n=cv.cout * 576, cout=cv.cout, cin=cv.cin, boff=off_of(cv.bias), w_fwd=_ptr(cv.w_fwd),
                                          w_dgrad=_ptr(cv.w_dgrad)))
            # 'tb': covered by the tail item
        for off, n in plain:
            for lo in range(0, n, 16384):
                items.append(L.UpdateItem(kind=1, woff=off + lo, n=min(16384, n - lo)))
        covered = sum(it.n for it in items) + sum(it.cout for it in items if it.kind == 3)
        if covered != flat_p.numel():
            raise RuntimeError('rumpy_amd: update table covers %d of %d parameters' % (covered, flat_p.numel()))
        self._update_items_host = (L.UpdateItem * len(items))(*items)
        self.update_items = (self._to_device_bytes(self._update_items_host), len(items))

    def _alloc_packed_h(self):
        """fp16 forward images of every MFMA conv (evaluation plans only: no data-gradient image, the packed biases are shared)"""
        items = []
        for cv in self.spec.convs():
            if cv.kind == 'head' or (cv.kind == 'tail' and self.wide):      # (a wide net's tail reads its fp32 master filter)
                continue
            kind = 0 if cv.kind == 'main' else 2
            cv.w_fwd_h = torch.empty(cv.cout * cv.cin * 9 if kind == 0 else 2 * 18 * 64 * 8, dtype=torch.float16, device=self.device)      # tail: filter + residual image
            items.append(L.PackItem(w=_ptr(cv.weight), b=_ptr(cv.bias), w_fwd=_ptr(cv.w_fwd_h), w_dgrad=None, b_packed=None, cout=cv.cout,
                                    cin=cv.cin, kind=kind, shuffle=1 if (kind == 0 and cv.shuffle) else 0, fmt=L.FMT_F16))
            if (self.generic_up or self.eval_up_residual) and any(cv is u for u in self.spec.ups):
                # the upsampler filters also enter evaluation as image + rounding-residual image (a second conv launch adds the residual's
                # contribution): their fp16 rounding alone moved a 32 dB EDSR 256 x 32 by -0.022 dB, a 64-feature RCAN by -0.011
                cv.w_fwd_h_lo = torch.empty(cv.cout * cv.cin * 9, dtype=torch.float16, device=self.device)
                items.append(L.PackItem(w=_ptr(cv.weight), b=_ptr(cv.bias), w_fwd=_ptr(cv.w_fwd_h_lo), w_dgrad=None, b_packed=None, cout=cv.cout,
                                        cin=cv.cin, kind=0, shuffle=1 if cv.shuffle else 0, fmt=L.FMT_F16_RESIDUAL))
        self._pack_items_h_host = (L.PackItem * len(items))(*items)
        self._pack_items_h = self._to_device_bytes(self._pack_items_h_host)
        self._n_pack_h = len(items)

    def _repack_h(self, stream):
        """bring the fp16 images up to date with the master weights (no-op when nothing was re-packed since)"""
        if self._pack_items_h is None:
            self._alloc_packed_h()
        if self.h_gen != self.pack_gen:
            L.check(self.lib.rumpy_pack_weights(_ptr(self._pack_items_h), self._n_pack_h, stream), 'rumpy_pack_weights')
            self.h_gen = self.pack_gen

    # ------------------------------------------------------------------ plan construction
    def _new(self, plan, *shape, dtype=BF16):
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        plan.keep.append(t)
        return t

    def _conv(self, ops, x, cv, N, H, W, out, dgrad=False, relu=False, scale=1.0, mask=None, res1=None, res2=None,
              pool=None, in_mode=0, out_mode=0, bias=True, fmt=0, w_lo=None):
        if dgrad:
            w, cin_chunks, cout_tiles, b = cv.w_dgrad, cv.cout // 64, cv.cin // 64, None
        else:
            w, cin_chunks, cout_tiles, b = (cv.w_fwd_h if fmt else cv.w_fwd), cv.cin // 64, cv.cout // 64, (cv.b_packed if bias else None)
        a = L.ConvArgs(x=_ptr(x), w=_ptr(w), bias=_ptr(b), out=_ptr(out), mask=_ptr(mask), res1=_ptr(res1), res2=_ptr(res2),
                       pool=_ptr(pool), N=N, H=H, W=W, cin_chunks=cin_chunks, cout_tiles=cout_tiles, in_mode=in_mode,
                       out_mode=out_mode, relu=1 if relu else 0, scale=float(scale), grid_x=0, fmt=fmt, w_lo=_ptr(w_lo))
        ops.append(('rumpy_conv3x3', a))

    def _build(self, N, H, W, train, fmt=0):
        spec, lib = self.spec, self.lib
        F = self.feats
        if fmt and train:
            raise RuntimeError('rumpy_amd: training plans are bf16')
        if fmt and self._pack_items_h is None:
            self._alloc_packed_h()
        wf = (lambda cv: cv.w_fwd_h) if fmt else (lambda cv: cv.w_fwd)       # forward filter image of this plan's format
        plan = _Plan()
        plan.N, plan.H, plan.W, plan.train, plan.fmt = N, H, W, train, fmt
        plan.gen = 0              # forward-pass counter of a training plan: a backward pass must belong to the LAST forward pass
        Cin, Cout = spec.head.cin, spec.tail.cout
        plan.x_in = self._new(plan, N, Cin, H, W, dtype=torch.float32)
        fwd, bwd = plan.fwd, plan.bwd
        tiles = int(lib.rumpy_conv_pool_tiles(H, W, F // 64))     # per-image pool partial rows written by the F->F conv (its kernel's tiling)
        # ... and by the one-launch residual block (column tiles when W > 48: another count of partial rows)
        btiles = int(lib.rumpy_block_pool_tiles(H, W))
        wjobs = []      # (layer, x, dy, H, W, dy_mode, scale, mt)

        free_pool = []
        protected = []          # data_ptrs of live skip sources (eval-mode buffer reuse must not recycle them)
        plan.scaled = []
        plan.ca_param_items = []
        plan.qca_items, plan.qca_dev = [], None      # gate MLPs of the styled QCALayers: parameter gradients in one launch
        plan.q_items, plan.q_shape, plan.q_dev = [], None, None
        plan.qn_items, plan.qn_shape, plan.qn_dev = [], None, None      # q-layers of another depth than 2 (rumpy_q_mlpn_*)
        plan.rcab_n, plan.rcab_xchg, plan.rcab_epoch, plan.rcab_status = 0, None, None, None
        plan.f8_f = plan.f8_b = None              # fp8 site records of the forward / backward launches (precision 'fp8')
        plan.f8_f_n = plan.f8_b_n = 0
        plan.f8_f_cal = plan.f8_b_cal = False
        # device status words read back together: [0] a non-finite output value (rumpy_tail_fwd, evaluation plans), [1] strip-exchange watchdog
        plan.flags = torch.zeros(2, dtype=torch.int32, device=self.device)
        plan.meta = self._new(plan, N, max(1, spec.num_metadata), dtype=torch.float32) if spec.num_metadata else None

        def act():
            if not train and free_pool:
                return free_pool.pop()
            return self._new(plan, N, H, W, F)

        def release(t):
            if not train and t.data_ptr() not in protected:
                free_pool.append(t)

        # ---- head ----
        a0 = act()
        protected.append(a0.data_ptr())
        plan.head_args = L.HeadFwdArgs(x=_ptr(plan.x_in), w=_ptr(spec.head.weight), b=_ptr(spec.head.bias),
                                       out=_ptr(a0), N=N, C=Cin, H=H, W=W, cout=F, fmt=fmt)
        fwd.append(('rumpy_head_fwd', plan.head_args))

        def emit_items(items, cur):
            """Emit forward ops for a chain of body items; returns (output buffer, backward node list).
            A backward node is a callable(g_out, extra) -> g_in for a residual unit, or
            ('group', conv, inner_out, sub_nodes) for a ResidualGroup."""
            nodes = []
            pending = None        # conv_rcab2.hip chain: the last block's (x, ungated branch u, pool partial rows, its attention MLP) - x + gate * u not formed yet

            def flush(pd):
                """x + gate * u of a chain's last block (streaming launch; the MLP is evaluated per workgroup from the partial rows)"""
                y = act()
                fwd.append(('rumpy_ca_fwd_fused', L.CaFwdFusedArgs(
                    pool=_ptr(pd['part']), w1=_ptr(pd['ca'].w1), b1=_ptr(pd['ca'].b1), w2=_ptr(pd['ca'].w2), b2=_ptr(pd['ca'].b2), mean=_ptr(pd['mean']),
                    hidden=_ptr(pd['hid']), gate=_ptr(pd['gate']), t=_ptr(pd['u']), res=_ptr(pd['x']), out=_ptr(y), N=N, HW=H * W, C=F, Cr=pd['ca'].Cr,
                    ntiles=pd['np'], inv_hw=1.0 / (H * W), qgate=_ptr(pd['qg']), fmt=fmt)))
                release(pd['u'])
                release(pd['x'])
                return y

            for it in items:
                lazy = (it[0] == 'rcab' and not getattr(it[3], 'gen', False) and it[3].Cr <= 4 and self.rcab_form != 'xchg' and self.use_rcab_kernel and self.use_block_kernel
                        and not self.wide and (W <= 48 or self.block_any_width) and not (train and (self.fp8 or not self.use_mask_bytes)))
                if lazy and self.rcab_form == 'auto':
                    lazy = W > 48 or int(self.lib.rumpy_rcab_strips(H, W)) > self.cus
                if pending is not None and not lazy:
                    cur = flush(pending)
                    pending = None
                if it[0] == 'resblock':
                    _, c1, c2, rs = it
                    # one launch per block (conv_block.hip; images wider than 48 pixels as column tiles since round 3): the activation
                    # between the two convs stays in LDS (it is still stored when training: the backward pass masks with it and the weight
                    # gradient of conv2 reads it); RUMPY_NO_BLOCK=1 keeps the two-launch path for A/B runs
                    fused = self.use_block_kernel and (W <= 48 or self.block_any_width) and not self.wide
                    t1 = act() if (train or not fused) else None
                    y = act()
                    # the backward launch needs t1 only as a ReLU mask: the forward launch also leaves it as bytes (1/16 of the traffic)
                    mb = self._new(plan, N, H, W, 8, dtype=torch.uint8) if (fused and train and self.use_mask_bytes) else None
                    f8 = fused and train and self.fp8 and W <= 48 and mb is not None and hasattr(c1, 'w8_fwd')
                    f8f = dict(w1_f8=_ptr(c1.w8_fwd), w2_f8=_ptr(c2.w8_fwd), f8_sw1=_ptr(c1.w8_scale), f8_sw2=_ptr(c2.w8_scale),
                               **self._f8_site(plan, 'f')) if f8 else {}
                    if fused:
                        fwd.append(('rumpy_conv_block', L.BlockArgs(
                            x=_ptr(cur), w1=_ptr(wf(c1)), b1=_ptr(c1.b_packed), w2=_ptr(wf(c2)), b2=_ptr(c2.b_packed), mask=None,
                            res2=None, t=_ptr(t1), out=_ptr(y), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=float(rs), maskbits=_ptr(mb), fmt=fmt, **f8f)))
                    else:
                        self._conv(fwd, cur, c1, N, H, W, t1, relu=True, fmt=fmt)
                        self._conv(fwd, t1, c2, N, H, W, y, scale=rs, res1=cur, fmt=fmt)

                    def node(g_out, extra, x_in=cur, t1=t1, c1=c1, c2=c2, rs=rs, fused=fused, mb=mb, f8=f8):
                        # y = x + rs*conv2(relu(conv1 x)):  dt1 = rs*dgrad2(g) masked ; dx = g + dgrad1(dt1) (+ extra)
                        dt1, dx = self._new(plan, N, H, W, F), self._new(plan, N, H, W, F)
                        f8b = dict(w1_f8=_ptr(c2.w8_dgrad), w2_f8=_ptr(c1.w8_dgrad), f8_sw1=_ptr(c2.w8_scale), f8_sw2=_ptr(c1.w8_scale),
                                   **self._f8_site(plan, 'b')) if f8 else {}
                        if fused:
                            bwd.append(('rumpy_conv_block', L.BlockArgs(
                                x=_ptr(g_out), w1=_ptr(c2.w_dgrad), b1=None, w2=_ptr(c1.w_dgrad), b2=None, mask=_ptr(t1),
                                res2=_ptr(extra), t=_ptr(dt1), out=_ptr(dx), N=N, H=H, W=W, relu1=0, scale1=float(rs), scale2=1.0,
                                maskbits=_ptr(mb), **f8b)))
                        else:
                            self._conv(bwd, g_out, c2, N, H, W, dt1, dgrad=True, scale=rs, mask=t1)
                            self._conv(bwd, dt1, c1, N, H, W, dx, dgrad=True, res1=g_out, res2=extra)
                        wjobs.append((c2, t1, g_out, H, W, 0, rs, 4))
                        wjobs.append((c1, x_in, dt1, H, W, 0, 1.0, 4))
                        return dx
                    nodes.append(node)
                    if t1 is not None:
                        release(t1)
                    release(cur)
                    cur = y
                elif it[0] == 'rcab':
                    _, c1, c2, ca = it[:4]
                    q = it[4] if len(it) > 4 else None
                    qh = qg = qdz = None
                    if spec.modulate:       # attention vector * attributes (constant w.r.t. the parameters: no gradient of its own)
                        if q is not None or spec.num_metadata != F:
                            raise RuntimeError('rumpy_amd: style "modulate" takes one attribute per feature channel and no q-layers')
                        qg = plan.meta
                    if q is not None and isinstance(q, QLayerNParams):      # ... with num_layers other than 2: the general-depth launches
                        qa = self._new(plan, N, max(q.hsum, 1), dtype=torch.float32)
                        qg = self._new(plan, N, F, dtype=torch.float32)
                        qdz = self._new(plan, N, F, dtype=torch.float32) if train else None
                        item = L.QMlpNItem(acts=_ptr(qa), gate=_ptr(qg), dzq=_ptr(qdz), nlayers=len(q.layers), scale=1.0)
                        for li, lay in enumerate(q.layers):
                            item.w[li], item.b[li], item.gw[li], item.gb[li] = _ptr(lay['w']), _ptr(lay['b']), _ptr(lay['gw']), _ptr(lay['gb'])
                        for li, wd in enumerate(q.widths):
                            item.n[li] = wd
                        plan.qn_items.append(item)
                        plan.qn_shape = tuple(q.widths)
                    elif q is not None:      # meta-attention gate of this QRCAB: evaluated for all layers by one launch before the forward ops
                        qh = self._new(plan, N, q.Hq, dtype=torch.float32)
                        qg = self._new(plan, N, F, dtype=torch.float32)
                        qdz = self._new(plan, N, F, dtype=torch.float32) if train else None
                        plan.q_items.append(L.QMlpItem(w1=_ptr(q.w1), b1=_ptr(q.b1), w2=_ptr(q.w2), b2=_ptr(q.b2), hidden=_ptr(qh),
                                                       gate=_ptr(qg), dzq=_ptr(qdz), gw1=_ptr(q.gw1), gb1=_ptr(q.gb1), gw2=_ptr(q.gw2),
                                                       gb2=_ptr(q.gb2), scale=1.0))
                        plan.q_shape = (q.M, q.Hq)
                    if getattr(ca, 'gen', False):
                        cur = self._emit_styled_rcab(plan, fwd, bwd, wjobs, nodes, c1, c2, ca, cur, N, H, W, tiles, train, act, release)
                        continue
                    if lazy:
                        # ---- one launch per block, no exchange inside it (conv_rcab2.hip): this launch stores the UNGATED branch u = conv2(t1) + b2 and its
                        # pool partial rows; the NEXT launch (or flush() at the chain's end) evaluates the gate and forms x + gate * u on its way in ----
                        np_out = int(lib.rumpy_rcab2_partials(N, H, W))
                        t1 = act() if train else None
                        u = act()
                        part = self._new(plan, N, np_out, F, dtype=torch.float32)
                        mean = self._new(plan, N, F, dtype=torch.float32)
                        hid = self._new(plan, N, ca.Cr, dtype=torch.float32)
                        gate = self._new(plan, N, F, dtype=torch.float32)
                        mbr = self._new(plan, N, H, W, 8, dtype=torch.uint8) if train else None
                        if not hasattr(plan, 'part_scratch'):
                            plan.part_scratch = self._new(plan, N, F, dtype=torch.float32)
                        common = dict(w1=_ptr(wf(c1)), b1=_ptr(c1.b_packed), w2=_ptr(wf(c2)), b2=_ptr(c2.b_packed), t=_ptr(t1), u_out=_ptr(u),
                                      part_out=_ptr(part), part_scratch=_ptr(plan.part_scratch), maskbits=_ptr(mbr), N=N, H=H, W=W, fmt=fmt)
                        if pending is None:
                            x_k = cur
                            fwd.append(('rumpy_rcab2_fwd', L.Rcab2Args(x=_ptr(cur), u_in=None, part_in=None, np_in=0, x_out=None, cr=ca.Cr, **common)))
                        else:
                            pd = pending
                            x_k = act()
                            fwd.append(('rumpy_rcab2_fwd', L.Rcab2Args(
                                x=_ptr(pd['x']), u_in=_ptr(pd['u']), part_in=_ptr(pd['part']), np_in=pd['np'], x_out=_ptr(x_k), cr=pd['ca'].Cr,
                                ca_w1=_ptr(pd['ca'].w1), ca_b1=_ptr(pd['ca'].b1), ca_w2=_ptr(pd['ca'].w2), ca_b2=_ptr(pd['ca'].b2),
                                mean=_ptr(pd['mean']), hidden=_ptr(pd['hid']), gate=_ptr(pd['gate']), qgate=_ptr(pd['qg']), **common)))
                            release(pd['u'])
                            release(pd['x'])
                        me = dict(x=x_k, u=u, part=part, np=np_out, ca=ca, qg=qg, mean=mean, hid=hid, gate=gate, link={})

                        def node(g_out, extra, me=me, prev=pending, t1=t1, c1=c1, c2=c2, ca=ca, qdz=qdz, mbr=mbr, np_out=np_out):
                            # G = g_out = dL/d(x + gate * u).  sum_hw(G * u) comes as partial rows from the launch that produced G (the next block's
                            # backward launch, which ran before this node) or, at the chain's end, from rumpy_ca_bwd_reduce
                            dz = self._new(plan, N, F, dtype=torch.float32)
                            dt2, dt1, dx = (self._new(plan, N, H, W, F) for _ in range(3))
                            lp = me['link'].get('part')
                            if lp is None:
                                nchunks = (H * W + 127) // 128
                                pb = self._new(plan, N, nchunks, F, dtype=torch.float32)
                                bwd.append(('rumpy_ca_bwd_reduce', L.CaBwdReduceArgs(dy=_ptr(g_out), t=_ptr(me['u']), partial=_ptr(pb), N=N, HW=H * W, C=F)))
                                lp = (pb, nchunks)
                            u_prev = part_out = None
                            if prev is not None:
                                u_prev = prev['u']
                                part_out = self._new(plan, N, np_out, F, dtype=torch.float32)
                                prev['link']['part'] = (part_out, np_out)
                            bwd.append(('rumpy_rcab2_bwd', L.Rcab2Args(
                                x=_ptr(g_out), u_in=_ptr(u_prev), part_in=_ptr(lp[0]), np_in=lp[1], part_out=_ptr(part_out), part_scratch=_ptr(plan.part_scratch),
                                w1=_ptr(c2.w_dgrad), b1=None, w2=_ptr(c1.w_dgrad), b2=None, x_out=_ptr(dt2), t=_ptr(dt1), u_out=_ptr(dx), res2=_ptr(extra),
                                maskbits=_ptr(mbr), ca_w1=_ptr(ca.w1), ca_b1=_ptr(ca.b1), ca_w2=_ptr(ca.w2), ca_b2=_ptr(ca.b2), mean=None,
                                hidden=_ptr(me['hid']), gate=_ptr(me['gate']), qgate=_ptr(me['qg']), dz=_ptr(dz), dzq=_ptr(qdz), N=N, H=H, W=W, cr=ca.Cr, fmt=0)))
                            plan.ca_param_items.append(L.CaMlpBwdArgs(
                                partial=_ptr(dz), mean=_ptr(me['mean']), hidden=_ptr(me['hid']), gate=_ptr(me['gate']), w1=_ptr(ca.w1), w2=_ptr(ca.w2),
                                dpool=_ptr(dz), gw1=_ptr(ca.gw1), gb1=_ptr(ca.gb1), gw2=_ptr(ca.gw2), gb2=_ptr(ca.gb2), N=N, C=F, Cr=ca.Cr,
                                nchunks=1, inv_hw=1.0 / (H * W), scale=1.0))
                            wjobs.append((c2, t1, dt2, H, W, 0, 1.0, 4))
                            wjobs.append((c1, me['x'], dt1, H, W, 0, 1.0, 4))
                            return dx
                        nodes.append(node)
                        if t1 is not None:
                            release(t1)
                        pending = me
                        cur = None
                        continue
                    t1, t2, y = act(), act(), act()
                    fused = self.use_block_kernel and (W <= 48 or self.block_any_width) and not self.wide
                    # the whole RCAB in one launch (conv_rcab.hip): the strips of an image exchange their pool sums, the gate is applied on chip
                    # (every strip of an image - strip rows x column tiles - has to be resident at the same time)
                    rc = fused and self.use_rcab_kernel and int(self.lib.rumpy_rcab_strips(H, W)) <= self.cus and 2 * plan.rcab_n + 2 <= 4096
                    ptiles = btiles if (fused and not rc) else tiles
                    pool = self._new(plan, N, ptiles, F, dtype=torch.float32)
                    mean = self._new(plan, N, F, dtype=torch.float32)
                    hid = self._new(plan, N, ca.Cr, dtype=torch.float32)
                    gate = self._new(plan, N, F, dtype=torch.float32)
                    rc_common, seq = None, None
                    if rc:
                        if plan.rcab_xchg is None:
                            plan.rcab_xchg = torch.zeros(int(self.lib.rumpy_rcab_xchg_bytes(N, H, W)), dtype=torch.uint8, device=self.device)
                            plan.rcab_epoch = torch.zeros(1, dtype=torch.int32, device=self.device)
                            plan.rcab_status = plan.flags[1:2]
                        seq = 2 * plan.rcab_n
                        plan.rcab_n += 1
                        mbr = self._new(plan, N, H, W, 8, dtype=torch.uint8) if (train and self.use_mask_bytes) else None
                        f8 = train and self.fp8 and W <= 48 and mbr is not None and hasattr(c1, 'w8_fwd')
                        f8f = dict(w1_f8=_ptr(c1.w8_fwd), w2_f8=_ptr(c2.w8_fwd), f8_sw1=_ptr(c1.w8_scale), f8_sw2=_ptr(c2.w8_scale),
                                   **self._f8_site(plan, 'f')) if f8 else {}
                        rc_f8b = (lambda c1=c1, c2=c2: dict(w1_f8=_ptr(c2.w8_dgrad), w2_f8=_ptr(c1.w8_dgrad), f8_sw1=_ptr(c2.w8_scale), f8_sw2=_ptr(c1.w8_scale),
                                               **self._f8_site(plan, 'b'))) if f8 else (lambda: {})
                        rc_common = dict(maskbits=_ptr(mbr), N=N, H=H, W=W, cr=ca.Cr, ca_w1=_ptr(ca.w1), ca_b1=_ptr(ca.b1), ca_w2=_ptr(ca.w2), ca_b2=_ptr(ca.b2),
                                         hidden=_ptr(hid), gate=_ptr(gate), qgate=_ptr(qg), xchg=_ptr(plan.rcab_xchg),
                                         xchg_bytes=plan.rcab_xchg.numel(), epoch=_ptr(plan.rcab_epoch), status=_ptr(plan.rcab_status))
                        fwd.append(('rumpy_rcab_fwd', L.RcabArgs(
                            x=_ptr(cur), w1=_ptr(wf(c1)), b1=_ptr(c1.b_packed), w2=_ptr(wf(c2)), b2=_ptr(c2.b_packed),
                            t=_ptr(t1) if train else None, t2=_ptr(t2) if train else None, out=_ptr(y), mean=_ptr(mean), seq=seq, fmt=fmt, **rc_common, **f8f)))
                    elif fused:   # conv -> ReLU -> conv (+ pool partial sums) in one launch, no residual yet (the gate comes first)
                        fwd.append(('rumpy_conv_block', L.BlockArgs(
                            x=_ptr(cur), w1=_ptr(wf(c1)), b1=_ptr(c1.b_packed), w2=_ptr(wf(c2)), b2=_ptr(c2.b_packed), mask=None,
                            res2=None, t=_ptr(t1) if train else None, out=_ptr(t2), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=1.0, res_mode=1,
                            res1=None, pool=_ptr(pool), fmt=fmt)))
                    else:
                        self._conv(fwd, cur, c1, N, H, W, t1, relu=True, fmt=fmt)
                        self._conv(fwd, t1, c2, N, H, W, t2, pool=pool, fmt=fmt)
                    if not rc:
                        # squeeze-excite MLP + gate * t2 + skip in one launch (the MLP is recomputed per workgroup)
                        fwd.append(('rumpy_ca_fwd_fused', L.CaFwdFusedArgs(
                            pool=_ptr(pool), w1=_ptr(ca.w1), b1=_ptr(ca.b1), w2=_ptr(ca.w2), b2=_ptr(ca.b2), mean=_ptr(mean),
                            hidden=_ptr(hid), gate=_ptr(gate), t=_ptr(t2), res=_ptr(cur), out=_ptr(y), N=N, HW=H * W, C=F, Cr=ca.Cr,
                            ntiles=ptiles, inv_hw=1.0 / (H * W), qgate=_ptr(qg), fmt=fmt)))

                    def node(g_out, extra, x_in=cur, t1=t1, t2=t2, c1=c1, c2=c2, ca=ca, mean=mean, hid=hid, gate=gate, fused=fused, qg=qg, qdz=qdz,
                             rc_common=rc_common, rc_seq=seq, rc_f8b=(rc_f8b if rc else None)):
                        # y = x + t2*gate:  dgate = sum(g*t2) -> MLP backward -> dpool ; dt2 = g*gate + dpool
                        nchunks = (H * W + 127) // 128
                        part = self._new(plan, N, nchunks, F, dtype=torch.float32)
                        dz = self._new(plan, N, F, dtype=torch.float32)
                        dt2, dt1, dx = (self._new(plan, N, H, W, F) for _ in range(3))
                        if rc_common is not None:
                            # sum(g * t2) over the image, MLP backward, dt2 = g * gate + dpool and both data gradients in one launch
                            bwd.append(('rumpy_rcab_bwd', L.RcabArgs(
                                x=_ptr(g_out), w1=_ptr(c2.w_dgrad), b1=None, w2=_ptr(c1.w_dgrad), b2=None, t=_ptr(dt1), t2=_ptr(dt2), t2_in=_ptr(t2),
                                mask=_ptr(t1), res2=_ptr(extra), out=_ptr(dx), mean=None, dz=_ptr(dz), dzq=_ptr(qdz),
                                seq=rc_seq + 1, **rc_common, **rc_f8b())))
                        else:
                            bwd.append(('rumpy_ca_bwd_reduce', L.CaBwdReduceArgs(dy=_ptr(g_out), t=_ptr(t2), partial=_ptr(part),
                                                                                  N=N, HW=H * W, C=F)))
                            # MLP backward (dpool) + dt2 = g * gate + dpool in one launch; dz is kept for the parameter gradients
                            # of ALL channel-attention layers, which come from one launch after the backward chain
                            bwd.append(('rumpy_ca_bwd_fused', L.CaBwdFusedArgs(
                                dy=_ptr(g_out), partial=_ptr(part), hidden=_ptr(hid), gate=_ptr(gate), w1=_ptr(ca.w1), w2=_ptr(ca.w2),
                                dz=_ptr(dz), dt=_ptr(dt2), N=N, HW=H * W, C=F, Cr=ca.Cr, nchunks=nchunks, inv_hw=1.0 / (H * W),
                                qgate=_ptr(qg), dzq=_ptr(qdz))))
                        plan.ca_param_items.append(L.CaMlpBwdArgs(
                            partial=_ptr(dz), mean=_ptr(mean), hidden=_ptr(hid), gate=_ptr(gate), w1=_ptr(ca.w1), w2=_ptr(ca.w2),
                            dpool=_ptr(dz), gw1=_ptr(ca.gw1), gb1=_ptr(ca.gb1), gw2=_ptr(ca.gw2), gb2=_ptr(ca.gb2), N=N, C=F, Cr=ca.Cr,
                            nchunks=1, inv_hw=1.0 / (H * W), scale=1.0))
                        if rc_common is not None:
                            pass
                        elif fused:   # both data gradients in one launch; the skip operand is the RCAB's incoming gradient
                            bwd.append(('rumpy_conv_block', L.BlockArgs(
                                x=_ptr(dt2), w1=_ptr(c2.w_dgrad), b1=None, w2=_ptr(c1.w_dgrad), b2=None, mask=_ptr(t1), res2=_ptr(extra),
                                t=_ptr(dt1), out=_ptr(dx), N=N, H=H, W=W, relu1=0, scale1=1.0, scale2=1.0, res_mode=2,
                                res1=_ptr(g_out), pool=None)))
                        else:
                            self._conv(bwd, dt2, c2, N, H, W, dt1, dgrad=True, mask=t1)
                            self._conv(bwd, dt1, c1, N, H, W, dx, dgrad=True, res1=g_out, res2=extra)
                        wjobs.append((c2, t1, dt2, H, W, 0, 1.0, 4))
                        wjobs.append((c1, x_in, dt1, H, W, 0, 1.0, 4))
                        return dx
                    nodes.append(node)
                    release(t1)
                    release(t2)
                    release(cur)
                    cur = y
                else:  # ('group', items, conv): y = conv(chain(x)) + x
                    _, sub, gconv = it
                    x_in = cur
                    protected.append(x_in.data_ptr())
                    inner, sub_nodes = emit_items(sub, x_in)
                    y = act()
                    self._conv(fwd, inner, gconv, N, H, W, y, res1=x_in, fmt=fmt)
                    protected.pop()
                    nodes.append(('group', gconv, inner, sub_nodes))
                    if inner is not x_in:
                        release(inner)
                    release(x_in)
                    cur = y
            if pending is not None:
                cur = flush(pending)
            return cur, nodes

        last, tree = emit_items(spec.body, a0)
        if train and self.fp8 and plan.f8_f_n == 0:
            #  the fp8 kernels are the W <= 48, mask-byte, one-launch forms: a plan that has none of them trains in bf16 - say so
            import warnings
            warnings.warn("rumpy_amd: precision='fp8' has no fp8 launch for %d x %d x %d training batches (the fp8 residual-block kernels are built for "
                          "patches up to 48 pixels wide, mask bytes on, one-launch forms on); this plan trains in bf16" % (N, H, W))
        r = act()
        self._conv(fwd, last, spec.body_conv, N, H, W, r, res1=a0, fmt=fmt)
        # ---- upsampler ----
        ups_in = []
        u, h, w = r, H, W
        for cv in spec.ups:
            rr = 3 if cv.cout == 9 * cv.cin else 2
            nxt = self._new(plan, N, rr * h, rr * w, F)
            if self.generic_up:
                pre = self._new(plan, N, h, w, cv.cout)          # natural channel order, then the PixelShuffle permutation as its own pass
                if fmt:
                    # fp16 evaluation plan: conv(x, w - fp16(w)) FIRST (the filter's rounding residual, _alloc_packed_h: small values, stored with
                    # their own exponent), then the conv on the fp16 filter adds it in fp32 before the one rounding of the result - the other
                    # order loses the correction in the second rounding (it is a quarter of an fp16 step of the result)
                    fwd.append(('rumpy_conv3x3', L.ConvArgs(x=_ptr(u), w=_ptr(cv.w_fwd_h_lo), bias=None, out=_ptr(pre), mask=None, res1=None,
                                                            res2=None, pool=None, N=N, H=h, W=w, cin_chunks=cv.cin // 64, cout_tiles=cv.cout // 64,
                                                            in_mode=0, out_mode=0, relu=0, scale=1.0, grid_x=0, fmt=fmt)))
                    self._conv(fwd, u, cv, N, h, w, pre, res1=pre, fmt=fmt)
                else:
                    self._conv(fwd, u, cv, N, h, w, pre, fmt=fmt)
                fwd.append(('rumpy_pixel_shuffle', L.PixelShuffleArgs(src=_ptr(pre), dst=_ptr(nxt), N=N, H=h, W=w, F=F, r=rr, inverse=0)))
            elif fmt and self.eval_up_residual and self.eval_up_fused:
                # 64 features, fp16 evaluation plan (round 3): ONE launch sweeps every strip with the filter's rounding-residual image and then
                # with the fp16 filter into the same fp32 accumulators (conv_up.hip, w_lo) - no residual launch, no scratch tensor written and
                # read back (4 x the stage's input), one pass over the input; the sum is rounded once, as before
                self._conv(fwd, u, cv, N, h, w, nxt, out_mode=1, fmt=fmt, w_lo=cv.w_fwd_h_lo)
            elif fmt and self.eval_up_residual:
                # RUMPY_EVAL_UP_FUSED=0 (A/B; round 2): the same as two launches with the PixelShuffle fused into the second one's store - the
                # residual's conv leaves its (small) result in conv-output order (channel tile = sub-pixel position, as the packed image has
                # it), the main conv reads it as its residual operand at that very index and adds it in fp32 before the one rounding
                pre = self._new(plan, N, h, w, cv.cout)
                fwd.append(('rumpy_conv3x3', L.ConvArgs(x=_ptr(u), w=_ptr(cv.w_fwd_h_lo), bias=None, out=_ptr(pre), mask=None, res1=None,
                                                        res2=None, pool=None, N=N, H=h, W=w, cin_chunks=cv.cin // 64, cout_tiles=cv.cout // 64,
                                                        in_mode=0, out_mode=0, relu=0, scale=1.0, grid_x=0, fmt=fmt)))
                self._conv(fwd, u, cv, N, h, w, nxt, out_mode=1, res1=pre, fmt=fmt)
            else:
                self._conv(fwd, u, cv, N, h, w, nxt, out_mode=1, fmt=fmt)
            ups_in.append((cv, u, h, w, rr))
            u, h, w = nxt, rr * h, rr * w
        # ---- tail (+ fused L1) ----
        plan.out = self._new(plan, N, Cout, h, w, dtype=torch.float32)
        plan.HR = (h, w)
        tail_grid = min(2 * self.cus, N * ((h + L.TILE_H - 1) // L.TILE_H) * ((w + L.TILE_W - 1) // L.TILE_W))
        plan.loss = self._new(plan, 1, dtype=torch.float32)
        plan.loss_partial = self._new(plan, max(tail_grid, 1), dtype=torch.float32)
        plan.target = self._new(plan, N, Cout, h, w, dtype=torch.float32)
        plan.dy4 = self._new(plan, N, h, w, 4) if train else None
        plan.nonfinite = None if train else plan.flags[0:1]      # raised by the tail kernel (fp16 overflow)
        plan.tail_fused = False
        if self.wide:
            # F -> 3 on the fp32 VALU from the master filter; no fused L1 form (the handlers take the generic loss path for wide nets)
            plan.tail_wide = L.TailWideArgs(x=_ptr(u), w=_ptr(spec.tail.weight), bias=_ptr(spec.tail.bias), out=_ptr(plan.out),
                                            nonfinite=_ptr(plan.nonfinite), N=N, H=h, W=w, F=F, C=Cout, fmt=fmt)
            plan.tail_plain = plan.tail_loss = None
            plan.tail_slabs, plan.tail_wslab = 0, None
        else:
            plan.tail_plain = L.TailFwdArgs(x=_ptr(u), w=_ptr(wf(spec.tail)), bias=_ptr(spec.tail.bias), out=_ptr(plan.out),
                                            target=None, dy4=None, loss_partial=None, loss=None, N=N, C=Cout, H=h, W=w, grid_x=tail_grid,
                                            nonfinite=_ptr(plan.nonfinite), fmt=fmt)
            # fused L1 training path: the tail conv's weight gradient is accumulated inside the same pass (one slab per workgroup)
            plan.tail_slabs = int(lib.rumpy_tail_fwd_grid(N, h, w, tail_grid))
            plan.tail_wslab = self._new(plan, plan.tail_slabs * int(lib.rumpy_wgrad_slab_floats(1)), dtype=torch.float32) if train else None
            plan.tail_loss = L.TailFwdArgs(x=_ptr(u), w=_ptr(wf(spec.tail)), bias=_ptr(spec.tail.bias), out=_ptr(plan.out),
                                           target=_ptr(plan.target), dy4=_ptr(plan.dy4), loss_partial=_ptr(plan.loss_partial),
                                           loss=_ptr(plan.loss), N=N, C=Cout, H=h, W=w, grid_x=tail_grid, wslab=_ptr(plan.tail_wslab),
                                           nonfinite=_ptr(plan.nonfinite), fmt=fmt)
        if not train:
            self._chain_runs(plan, fwd, 0)
            return plan

        # =============================== backward ===============================
        plan.gout_stage = None
        g = self._new(plan, N, h, w, F)
        if self.wide:
            bwd.append(('rumpy_tail_dgrad_wide', L.TailWideArgs(x=_ptr(plan.dy4), w=_ptr(spec.tail.weight), bias=None, out=_ptr(g), nonfinite=None,
                                                                N=N, H=h, W=w, F=F, C=Cout)))
        else:
            bwd.append(('rumpy_tail_dgrad', L.TailDgradArgs(dy4=_ptr(plan.dy4), w=_ptr(spec.tail.w_dgrad), dx=_ptr(g), N=N, H=h, W=w)))
        wjobs.append((spec.tail, u, plan.dy4, h, w, 2, 1.0, 1))
        fuse_tail = (self.fuse_tail_dgrad and not self.wide and not self.generic_up and F == 64 and bool(ups_in)
                     and ups_in[-1][4] == 2 and N * h * w * F < 2 ** 31)
        for cv, uin, uh, uw, rr in reversed(ups_in):
            gin = self._new(plan, N, uh, uw, F)
            if fuse_tail:
                # the last stage: its input gradient g = conv^T_tail(dy4) is made tile by tile inside the launch (and written for the weight gradient)
                fuse_tail = False
                assert bwd[-1][0] == 'rumpy_tail_dgrad'
                bwd[-1] = ('rumpy_conv4d_tail', L.Conv4dTailArgs(dy4=_ptr(plan.dy4), w_tail=_ptr(spec.tail.w_dgrad), dx=_ptr(g), w=_ptr(cv.w_dgrad),
                                                                  out=_ptr(gin), N=N, H=uh, W=uw, grid_x=0))
                wjobs.append((cv, uin, g, uh, uw, 1, 1.0, 4))
                g = gin
                continue
            if self.generic_up:
                # the gradient un-permuted into the conv's natural channel order: data and weight gradient are a plain conv's then.  The data
                # gradient has Cin = r^2 F input channels - beyond rumpy_conv3x3's 64 / 256 - and runs the encoder's general conv kernel.
                gpre = self._new(plan, N, uh, uw, cv.cout)
                bwd.append(('rumpy_pixel_shuffle', L.PixelShuffleArgs(src=_ptr(g), dst=_ptr(gpre), N=N, H=uh, W=uw, F=F, r=rr, inverse=1)))
                if not hasattr(plan, 'zero_bias'):
                    plan.zero_bias = torch.zeros(max(64, F), dtype=torch.float32, device=self.device)
                bwd.append(('rumpy_enc_conv', L.EncConvArgs(x=_ptr(gpre), w=_ptr(cv.w_dgrad), bias=_ptr(plan.zero_bias), out=_ptr(gin), N=N, H=uh, W=uw,
                                                            cin=cv.cout, cout=cv.cin, stride=1, neg_slope=1.0)))
                wjobs.append((cv, uin, gpre, uh, uw, 0, 1.0, 4))
            else:
                self._conv(bwd, g, cv, N, uh, uw, gin, dgrad=True, in_mode=1)
                wjobs.append((cv, uin, g, uh, uw, 1, 1.0, 4))
            g = gin
        g_r = g                                      # grad wrt r = body_conv(last) + a0
        g_last = self._new(plan, N, H, W, F)
        self._conv(bwd, g_r, spec.body_conv, N, H, W, g_last, dgrad=True)
        wjobs.append((spec.body_conv, last, g_r, H, W, 0, 1.0, 4))

        def run_tree(nodes, g_out, extra_first):
            for idx in range(len(nodes) - 1, -1, -1):
                nd = nodes[idx]
                extra = extra_first if idx == 0 else None
                if callable(nd):
                    g_out = nd(g_out, extra)
                else:
                    _, gconv, inner_out, sub = nd
                    # y = gconv(inner_out) + x_in ; g_out = dL/dy
                    g_inner = self._new(plan, N, H, W, F)
                    self._conv(bwd, g_out, gconv, N, H, W, g_inner, dgrad=True)
                    wjobs.append((gconv, inner_out, g_out, H, W, 0, 1.0, 4))
                    if sub:
                        g_in = run_tree(sub, g_inner, g_out)      # skip gradient joins at the group's first block
                        if extra is not None:
                            g_in = self._add_extra(plan, bwd, g_in, extra, N, H, W)
                    else:
                        g_in = self._add_extra(plan, bwd, g_inner, g_out, N, H, W)
                        if extra is not None:
                            g_in = self._add_extra(plan, bwd, g_in, extra, N, H, W)
                    g_out = g_in
            return g_out

        if tree:
            g_a0 = run_tree(tree, g_last, g_r)       # global skip gradient g_r joins at the first body block
        else:
            g_a0 = self._add_extra(plan, bwd, g_last, g_r, N, H, W)

        # ---- head weight gradient ----
        slab = self._new(plan, int(lib.rumpy_head_wgrad_slab_floats(Cin, F)), dtype=torch.float32)
        # with the one-launch reduction (finish.hip) the head conv's slabs are added up there: gw = NULL defers it
        a = L.HeadWgradArgs(x=_ptr(plan.x_in), dy=_ptr(g_a0), slab=_ptr(slab), gw=None if self.use_finish else _ptr(spec.head.gw),
                            gb=_ptr(spec.head.gb), N=N, C=Cin, H=H, W=W, cout=F, scale=1.0)
        plan.head_slab, plan.head_nslabs = slab, int(lib.rumpy_head_wgrad_slabs(N, H, W))
        bwd.append(('rumpy_head_wgrad', a))
        plan.scaled.append(a)
        plan.head_wgrad_args = a

        # ---- grouped weight gradients ----
        self._emit_wgrad(plan, wjobs, N)
        self._chain_runs(plan, fwd, 0)
        self._chain_runs(plan, bwd, 1)
        return plan

    def _chain_runs(self, plan, ops, backward):
        """replace every maximal run (>= 2) of consecutive ResBlock-form rumpy_conv_block launches, each reading the previous one's output, by ONE
        rumpy_res_chain launch (conv_chain.hip).  In place: `ops` is the plan's launch list."""
        N, H, W = plan.N, plan.H, plan.W
        strips = int(self.lib.rumpy_res_chain_strips(N, H, W))      # 6-row x 48-column strips for W <= 48, 4-row x 64-column strips for 48 < W <= 64 (round 6); 0: wider
        if not self.use_chain or self.wide or strips == 0 or strips > self.cus:
            return
        # The 4-row strips of the 64-column geometry carry 1.5 x their MFMAs in halo rows (the 6-row strips 1.33 x) and have no conv at the outer end: against one
        # launch per block over column tiles they win only where the chip is (nearly) full - EDSR x4, 64-px crops, same box: 192 strips +3.6 %, 256 +1 %, but
        # 128 strips -7 %, 64 -9 % (profiles/r06_negative_results.txt).  The 48-column geometry wins at every batch measured (64 strips +8 %, 128 +7 %, 192 +7 %).
        if W > 48 and 4 * strips < 3 * self.cus and os.environ.get('RUMPY_CHAIN_ANY_FILL') != '1':
            return

        def chainable(a):
            if a.res_mode != 0 or a.pool or a.w1_f8 or a.col_tile or a.N != N or a.H != H or a.W != W:
                return False
            if backward:
                return (not a.relu1) and bool(a.maskbits) and a.fmt == 0
            return bool(a.relu1) and a.scale1 == 1.0 and not a.mask
        i = 0
        while i < len(ops):
            j = i
            if ops[i][0] == 'rumpy_conv_block' and chainable(ops[i][1]):
                j = i + 1
                while j < len(ops) and ops[j][0] == 'rumpy_conv_block' and chainable(ops[j][1]) and ops[j][1].x == ops[j - 1][1].out and ops[j][1].fmt == ops[i][1].fmt:
                    j += 1
            if j - i >= 2:
                blocks = [a for _, a in ops[i:j]]
                tab = (L.ResChainBlock * len(blocks))(*[L.ResChainBlock(x=a.x, w1=a.w1, b1=a.b1, w2=a.w2, b2=a.b2, res2=a.res2, t=a.t, out=a.out,
                                                                          maskbits=a.maskbits, scale1=a.scale1, scale2=a.scale2) for a in blocks])
                dev = self._to_device_bytes(tab)
                if getattr(plan, 'chain_work', None) is None:
                    plan.chain_work = torch.zeros(int(self.lib.rumpy_res_chain_work_bytes(N, H)), dtype=torch.uint8, device=self.device)
                    plan.rcab_status = plan.flags[1:2]          # the hand-off watchdog shares the strip-exchange status word (read back with the loss)
                plan.keep += [dev, plan.chain_work]
                args = L.ResChainArgs(blocks=_ptr(dev), nblocks=len(blocks), N=N, H=H, W=W, backward=backward, fmt=blocks[0].fmt, work=_ptr(plan.chain_work),
                                      work_bytes=plan.chain_work.numel(), status=_ptr(plan.flags[1:2]), fake_xcc=0, force_sc1=1 if self.chain_force_sc1 else 0)
                args._blocks_host = tab            # (kept alive with the argument block)
                ops[i:j] = [('rumpy_res_chain', args)]
                j = i + 1
                # the single conv at the run's outer end - EDSR's body-end conv behind the last block, its data gradient in front of the first - joins
                # the launch (rumpy_res_chain_args.edge_*; bitwise the separate launch)
                def plain_conv(c):
                    return (c.cin_chunks == 1 and c.cout_tiles == 1 and c.in_mode == 0 and c.out_mode == 0 and not c.relu and c.scale == 1.0 and not c.mask
                            and not c.pool and not c.res2 and not c.w_lo and c.N == N and c.H == H and c.W == W and c.fmt == args.fmt)
                if self.chain_edge and len(blocks) <= 254 and W <= 48:       # (the edge conv is built into the 6-row geometry)
                    if not backward and j < len(ops) and ops[j][0] == 'rumpy_conv3x3' and plain_conv(ops[j][1]) and ops[j][1].x == blocks[-1].out:
                        c = ops[j][1]
                        args.edge_w, args.edge_b, args.edge_res, args.edge_out = c.w, c.bias, c.res1, c.out
                        del ops[j]
                    elif (backward and i > 0 and ops[i - 1][0] == 'rumpy_conv3x3' and plain_conv(ops[i - 1][1]) and ops[i - 1][1].out == blocks[0].x
                          and not ops[i - 1][1].bias and not ops[i - 1][1].res1):
                        c = ops[i - 1][1]
                        args.edge_w, args.edge_x = c.w, c.x
                        del ops[i - 1]
                        i, j = i - 1, j - 1            # (the chain op moved one place down)
            i = max(j, i + 1)

    def _emit_styled_rcab(self, plan, fwd, bwd, wjobs, nodes, c1, c2, ca, cur, N, H, W, tiles, train, act, release):
        """A QRCAB whose channel-attention gate also reads the attribute vector (StyledCAParams): the residual block in one launch with pool
        sums (or two conv launches when W > 48), the gate MLP in its own launch, gate * t2 + x by the streaming kernel; backward in the
        same pieces.  bf16 plans only (the engine keeps evaluation of such networks in bf16)."""
        F = self.feats
        fused = self.use_block_kernel and (W <= 48 or self.block_any_width)
        if fused:
            tiles = int(self.lib.rumpy_block_pool_tiles(H, W))
        t1, t2, y = act(), act(), act()
        pool = self._new(plan, N, tiles, F, dtype=torch.float32)
        acts = self._new(plan, N, L.QCA_ACT_STRIDE, dtype=torch.float32)
        gate = self._new(plan, N, F, dtype=torch.float32)
        if fused:
            fwd.append(('rumpy_conv_block', L.BlockArgs(x=_ptr(cur), w1=_ptr(c1.w_fwd), b1=_ptr(c1.b_packed), w2=_ptr(c2.w_fwd), b2=_ptr(c2.b_packed),
                                                        mask=None, res2=None, t=_ptr(t1), out=_ptr(t2), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=1.0,
                                                        res_mode=1, res1=None, pool=_ptr(pool))))
        else:
            self._conv(fwd, cur, c1, N, H, W, t1, relu=True)
            self._conv(fwd, t1, c2, N, H, W, t2, pool=pool)

        def qca_args(**kw):
            a = L.QcaArgs(nlayers=len(ca.layers), N=N, C=F, M=ca.M, ntiles=tiles, inv_hw=1.0 / (H * W), scale=1.0, attr=_ptr(plan.meta),
                          acts=_ptr(acts), gate=_ptr(gate), **kw)
            for i, ly in enumerate(ca.layers):
                a.layers[i] = L.QcaLayer(w=_ptr(ly['w']), b=_ptr(ly['b']), gw=_ptr(ly['gw']), gb=_ptr(ly['gb']), n_prev=ly['n_prev'],
                                         n_out=ly['w'].shape[0], cat=ly['cat'], relu_in=ly['relu_in'], act=ly['act'])
            return a
        fwd.append(('rumpy_qca_gate_fwd', qca_args(pool=_ptr(pool))))
        fwd.append(('rumpy_ca_scale_res_fwd', L.CaScaleArgs(t=_ptr(t2), res=_ptr(cur), gate=_ptr(gate), out=_ptr(y), N=N, HW=H * W, C=F)))

        def node(g_out, extra, x_in=cur, t1=t1, t2=t2):
            nchunks = (H * W + 127) // 128
            part = self._new(plan, N, nchunks, F, dtype=torch.float32)
            dpool = self._new(plan, N, F, dtype=torch.float32)
            delta = self._new(plan, N, L.QCA_ACT_STRIDE, dtype=torch.float32)
            dt2, dt1, dx = (self._new(plan, N, H, W, F) for _ in range(3))
            bwd.append(('rumpy_ca_bwd_reduce', L.CaBwdReduceArgs(dy=_ptr(g_out), t=_ptr(t2), partial=_ptr(part), N=N, HW=H * W, C=F)))
            ga = qca_args(partial=_ptr(part), nchunks=nchunks, dpool=_ptr(dpool), delta=_ptr(delta))
            bwd.append(('rumpy_qca_gate_bwd', ga))
            plan.qca_items.append(ga)
            bwd.append(('rumpy_ca_bwd_apply', L.CaBwdApplyArgs(dy=_ptr(g_out), gate=_ptr(gate), dpool=_ptr(dpool), dt=_ptr(dt2), N=N, HW=H * W, C=F)))
            if fused:
                bwd.append(('rumpy_conv_block', L.BlockArgs(x=_ptr(dt2), w1=_ptr(c2.w_dgrad), b1=None, w2=_ptr(c1.w_dgrad), b2=None, mask=_ptr(t1),
                                                            res2=_ptr(extra), t=_ptr(dt1), out=_ptr(dx), N=N, H=H, W=W, relu1=0, scale1=1.0, scale2=1.0,
                                                            res_mode=2, res1=_ptr(g_out), pool=None)))
            else:
                self._conv(bwd, dt2, c2, N, H, W, dt1, dgrad=True, mask=t1)
                self._conv(bwd, dt1, c1, N, H, W, dx, dgrad=True, res1=g_out, res2=extra)
            wjobs.append((c2, t1, dt2, H, W, 0, 1.0, 4))
            wjobs.append((c1, x_in, dt1, H, W, 0, 1.0, 4))
            return dx
        nodes.append(node)
        release(t1)
        release(t2)
        release(cur)
        return y

    def _add_extra(self, plan, ops, g, extra, N, H, W):
        """g + extra through the CA scale kernel with a unit gate (rare path: group without residual units)."""
        F = self.feats
        if not hasattr(plan, 'ones_gate'):
            plan.ones_gate = torch.ones(N, F, dtype=torch.float32, device=self.device)
        out = self._new(plan, N, H, W, F)
        ops.append(('rumpy_ca_scale_res_fwd', L.CaScaleArgs(t=_ptr(g), res=_ptr(extra), gate=_ptr(plan.ones_gate), out=_ptr(out),
                                                             N=N, HW=H * W, C=F)))
        return out

    def _emit_wgrad(self, plan, wjobs, N):
        lib = self.lib
        jobs = {4: [], 1: []}
        items = []
        slab_floats = {4: int(lib.rumpy_wgrad_slab_floats(4)), 1: int(lib.rumpy_wgrad_slab_floats(1))}
        # first pass: count slabs
        layout = []
        total = 0
        ntiles_of = lambda H, W: N * ((H + L.TILE_H - 1) // L.TILE_H) * ((W + L.TILE_W - 1) // L.TILE_W)
        # ---- shares (mt = 4 units only): group the units (two_phase: layers at / above the split pointer = group 0, finished first), cut each
        # group's tile sequence into `cus` equal shares; a unit's ranges are its intersections with the shares ----
        units4 = [(cv, ch, ct, ntiles_of(H, W), 1)
                  for (cv, x, dy, H, W, dy_mode, scale, mt) in wjobs if mt == 4 for ch in range(cv.cin // 64) for ct in range(cv.cout // 64)]
        share_ranges, share_split_ptr, share_counts = None, None, None
        if self.wgrad_shares and units4:
            group_of = [0] * len(units4)
            if self.wgrad_two_phase:
                per_layer = {}
                for cv, ch, ct, nt, wt in units4:
                    per_layer[_ptr(cv.gw)] = per_layer.get(_ptr(cv.gw), 0) + nt
                if len(per_layer) >= 2:
                    tot, acc = sum(per_layer.values()), 0
                    for ptr in sorted(per_layer, reverse=True):
                        acc += per_layer[ptr]
                        share_split_ptr = ptr
                        if 2 * acc >= tot:
                            break
                    group_of = [0 if _ptr(u[0].gw) >= share_split_ptr else 1 for u in units4]
                    if len(set(group_of)) < 2:
                        group_of, share_split_ptr = [0] * len(units4), None
            share_ranges, share_counts, base_share = [None] * len(units4), {}, 0
            for grp in sorted(set(group_of)):
                idx = [i for i in range(len(units4)) if group_of[i] == grp]
                T_all = sum(units4[i][3] for i in idx)
                nsh = max(1, min(int(os.environ.get('RUMPY_WGRAD_NSH', self.cus)), T_all))      # env: A/B runs of the share count
                # the four output-channel tiles of an upsampler conv (64 -> 256 + pixel shuffle) read the same x tiles: aligned units of cut_wgrad_shares
                quad_of = lambda u: (id(u[0]), u[1]) if (self.wgrad_align and u[0].shuffle and u[0].cout == 256) else None
                cut = cut_wgrad_shares([(units4[i][3], quad_of(units4[i]), units4[i][2]) for i in idx], nsh, float(os.environ.get('RUMPY_WGRAD_JOB_COST', 4.0)))
                for i, rs in zip(idx, cut):
                    share_ranges[i] = [(t, t1, base_share + k, prio) for t, t1, k, prio in rs]
                share_counts[grp] = (base_share, nsh)
                base_share += nsh
        u4 = 0
        for (cv, x, dy, H, W, dy_mode, scale, mt) in wjobs:
            # split the layer's N*tiles_y*tiles_x pixel tiles into jobs of about wgrad_pixels_per_job pixels (fixed-size jobs), or take the
            # unit's share ranges
            ntile = ntiles_of(H, W)
            per = max(1, self.wgrad_pixels_per_job // (L.TILE_H * L.TILE_W))
            njob = max(1, (ntile + per - 1) // per)
            per = (ntile + njob - 1) // njob
            fixed = [(t0, min(ntile, t0 + per), -1, 1) for t0 in range(0, ntile, per)]
            cin_chunks = cv.cin // 64
            cout_tiles = cv.cout // 64 if mt == 4 else 1
            for ch in range(cin_chunks):
                for ct in range(cout_tiles):
                    ranges = fixed
                    if mt == 4 and share_ranges is not None:
                        ranges = share_ranges[u4]
                    if mt == 4:
                        u4 += 1
                    layout.append((cv, x, dy, H, W, dy_mode, scale, mt, ch, ct, ranges, total))
                    total += len(ranges) * slab_floats[mt]
        slabs = self._new(plan, max(total, 1), dtype=torch.float32)
        base = slabs.data_ptr()
        keyed = {4: [], 1: []}
        for li, (cv, x, dy, H, W, dy_mode, scale, mt, ch, ct, ranges, off) in enumerate(layout):
            sf = slab_floats[mt]
            for k, (t0, t1, share, prio) in enumerate(ranges):
                if dy_mode == 0:
                    dcs, dco = cv.cout, ct * 64
                elif dy_mode == 1:
                    dcs, dco = 64, ct
                else:
                    dcs, dco = 4, 0
                # launch order: jobs that read the same x tiles (the cout tiles of one tile range) sit next to each other,
                # so the x halo re-reads of an upsampler conv hit L2; slab addresses do not depend on the order
                keyed[mt].append(((id(x), k, li, share, prio), L.WgradJob(x=_ptr(x), dy=_ptr(dy), slab=base + 4 * (off + k * sf), n0=0, n1=N,
                                                              t0=t0, t1=t1, H=H, W=W, x_cstride=cv.cin, x_coff=ch * 64, dy_mode=dy_mode,
                                                              dy_cstride=dcs, dy_coff=dco, mt=mt)))
            items.append(L.ReduceItem(slab=base + 4 * off, slab_stride=sf, njobs=len(ranges), mt=mt,
                                      co_count=(64 if mt == 4 else cv.cout), co_mode=1 if (mt == 4 and cv.shuffle) else 0,
                                      co_off=(ct if (mt == 4 and cv.shuffle) else ct * 64), ci_total=cv.cin, ci_off=ch * 64,
                                      write_bias=1 if ch == 0 else 0, scale=float(scale), gw=_ptr(cv.gw), gb=_ptr(cv.gb)))
        first_seen = {}
        for mt in (4, 1):
            for key, _ in keyed[mt]:
                first_seen.setdefault(key[0], len(first_seen))
            jobs[mt] = [jb for _, jb in sorted(keyed[mt], key=lambda kj: (first_seen[kj[0][0]], kj[0][1], kj[0][2]))]
        plan.shares = None
        if share_ranges is not None:
            # jobs in share order (within a share: the aligned jobs, then layer order); first[s] = index of share s's first job
            order = sorted(keyed[4], key=lambda kj: (kj[0][3], kj[0][4], kj[0][2]))
            jobs[4] = [jb for _, jb in order]
            nshares = sum(n for _, n in share_counts.values())
            first = [0] * (nshares + 1)
            for key, _ in order:
                first[key[3] + 1] += 1
            for k in range(nshares):
                first[k + 1] += first[k]
            first_dev = torch.tensor(first, dtype=torch.int32, device=self.device)
            plan.keep.append(first_dev)
            plan.shares = dict(first=first_dev, n=nshares, groups=share_counts)
        plan.reduce_scales = [it.scale for it in items]
        plan.reduce_host = (L.ReduceItem * len(items))(*items)
        plan.reduce_dev = torch.empty(C.sizeof(plan.reduce_host), dtype=torch.uint8, device=self.device)
        plan.keep.append(plan.reduce_dev)
        plan.n_reduce = len(items)
        plan.reduce_keep = [i for i, it in enumerate(items) if it.mt != 1]      # items used when the tail gradient came from tail_fwd
        plan.reduce_dev_notail = torch.empty(max(1, len(plan.reduce_keep)) * C.sizeof(L.ReduceItem), dtype=torch.uint8, device=self.device)
        plan.keep.append(plan.reduce_dev_notail)
        plan.job_dev = {}
        for mt in (4, 1):
            if jobs[mt]:
                arr = (L.WgradJob * len(jobs[mt]))(*jobs[mt])
                dev = self._to_device_bytes(arr)
                plan.keep.append(dev)
                plan.job_dev[mt] = (dev, len(jobs[mt]))
        # Two-phase form for data-parallel runs (backward(on_ready=...)): the layers in the upper part of the flat gradient buffer
        # (group A: gradient pointer >= split pointer; about half of the jobs) are finished first, so that their all-reduce can run on
        # the side stream while the jobs of group B are computed.  Same jobs, same slabs, same reduction order: bitwise the same gradients.
        plan.split = None
        per_layer = {}
        for it in items:
            if it.mt == 4:
                per_layer[it.gw] = per_layer.get(it.gw, 0) + it.njobs
        if plan.shares is not None:
            if share_split_ptr is not None:           # two groups of shares: the same job table, two windows of `first`
                in_a = lambda idx: items[idx].gw >= share_split_ptr
                idx_a = [i for i in range(len(items)) if in_a(i)]
                idx_b = [i for i in range(len(items)) if not in_a(i)]
                mk = lambda n: torch.empty(max(1, n) * C.sizeof(L.ReduceItem), dtype=torch.uint8, device=self.device)
                plan.split = dict(ptr=share_split_ptr, shares_a=share_counts[0], shares_b=share_counts[1], idx_a=idx_a,
                                  idx_a_notail=[i for i in idx_a if items[i].mt != 1], idx_b=idx_b,
                                  red_a=mk(len(idx_a)), red_a_notail=mk(len(idx_a)), red_b=mk(len(idx_b)))
                plan.keep += [plan.split['red_a'], plan.split['red_a_notail'], plan.split['red_b']]
        elif len(per_layer) >= 2:
            total4, acc4, split_ptr = sum(per_layer.values()), 0, None
            for ptr in sorted(per_layer, reverse=True):
                acc4 += per_layer[ptr]
                split_ptr = ptr
                if 2 * acc4 >= total4:
                    break
            item_of_slab = {}
            for idx, it in enumerate(items):
                for k in range(it.njobs):
                    item_of_slab[it.slab + 4 * k * it.slab_stride] = idx
            in_a = lambda idx: items[idx].gw >= split_ptr
            ja = [jb for jb in jobs[4] if in_a(item_of_slab[jb.slab])]
            jb_ = [jb for jb in jobs[4] if not in_a(item_of_slab[jb.slab])]
            if ja and jb_:
                dev_a, dev_b = self._to_device_bytes((L.WgradJob * len(ja))(*ja)), self._to_device_bytes((L.WgradJob * len(jb_))(*jb_))
                idx_a = [i for i in range(len(items)) if in_a(i)]          # the tail conv (mt 1) has the highest pointer of the EDSR / RCAN layouts
                idx_b = [i for i in range(len(items)) if not in_a(i)]
                mk = lambda n: torch.empty(max(1, n) * C.sizeof(L.ReduceItem), dtype=torch.uint8, device=self.device)
                plan.split = dict(ptr=split_ptr, jobs_a=(dev_a, len(ja)), jobs_b=(dev_b, len(jb_)), idx_a=idx_a,
                                  idx_a_notail=[i for i in idx_a if items[i].mt != 1], idx_b=idx_b,
                                  red_a=mk(len(idx_a)), red_a_notail=mk(len(idx_a)), red_b=mk(len(idx_b)))
                plan.keep += [dev_a, dev_b, plan.split['red_a'], plan.split['red_a_notail'], plan.split['red_b']]
        plan.grad_scale = None

    def _set_grad_scale(self, plan, gs):
        """The stored activation gradients are unscaled (+-1 at the loss); 1/numel enters once, in fp32, where the
        parameter gradients are written."""
        if plan.grad_scale == gs:
            return
        for it, s in zip(plan.reduce_host, plan.reduce_scales):
            it.scale = s * gs
        raw = np.frombuffer(bytes(plan.reduce_host), dtype=np.uint8).copy()
        plan.reduce_dev.copy_(torch.from_numpy(raw), non_blocking=False)
        sub = (L.ReduceItem * max(1, len(plan.reduce_keep)))(*[plan.reduce_host[i] for i in plan.reduce_keep])
        plan.reduce_dev_notail.copy_(torch.from_numpy(np.frombuffer(bytes(sub), dtype=np.uint8).copy()), non_blocking=False)
        if plan.split is not None:
            for key, idxs in (('red_a', plan.split['idx_a']), ('red_a_notail', plan.split['idx_a_notail']), ('red_b', plan.split['idx_b'])):
                if idxs:
                    tab = (L.ReduceItem * len(idxs))(*[plan.reduce_host[i] for i in idxs])
                    raw_t = torch.from_numpy(np.frombuffer(bytes(tab), dtype=np.uint8).copy())
                    plan.split[key][:raw_t.numel()].copy_(raw_t, non_blocking=False)
        for a in plan.scaled:
            a.scale = gs
        if plan.ca_param_items:
            for a in plan.ca_param_items:
                a.scale = gs
            arr = (L.CaMlpBwdArgs * len(plan.ca_param_items))(*plan.ca_param_items)
            raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            if getattr(plan, 'ca_params_dev', None) is None:
                plan.ca_params_dev = torch.empty(raw.size, dtype=torch.uint8, device=self.device)
            plan.ca_params_dev.copy_(torch.from_numpy(raw), non_blocking=False)
        if plan.q_items or plan.qn_items:
            for a in plan.q_items + plan.qn_items:
                a.scale = gs
            self._upload_q_items(plan)
        if plan.qca_items:
            for a in plan.qca_items:
                a.scale = gs
            plan.qca_dev = self._to_device_bytes((L.QcaArgs * len(plan.qca_items))(*plan.qca_items))
        plan.grad_scale = gs

    def _advance_epoch(self, plan, stream):
        """new tag epoch for the strip exchanges of the RCAB kernels (forward and backward of one pass share it)"""
        if plan.rcab_epoch is not None:
            L.check(self.lib.rumpy_rcab_epoch_advance(_ptr(plan.rcab_epoch), stream), 'rumpy_rcab_epoch_advance')

    def exchange_status(self, plan=None):
        """0, or the code an RCAB launch left when a strip exchange timed out (synchronises)"""
        worst = 0
        for p in ([plan] if plan is not None else list(self.plans.values())):
            if getattr(p, 'rcab_status', None) is not None:
                worst = max(worst, int(p.rcab_status.item()))
        return worst

    # ------------------------------------------------------------------ watchdog of the launches that wait for other workgroups (round 6)
    @staticmethod
    def _waiting_kinds(plan):
        """which kinds of launches of `plan` poll for another workgroup: 'chain' (rumpy_res_chain hand-offs), 'xchg' (the pool exchange of
        rumpy_rcab_fwd / _bwd)"""
        kinds = set()
        for name, _ in list(plan.fwd) + list(getattr(plan, 'bwd', None) or []):
            if name == 'rumpy_res_chain':
                kinds.add('chain')
            if name in ('rumpy_rcab_fwd', 'rumpy_rcab_bwd'):
                kinds.add('xchg')
        return kinds

    @staticmethod
    def watchdog_text(code):
        """what a non-zero status word says (csrc: 0x4ff / 0x500 + b = a hand-off of the persistent block chain, 0x300 + seq = the pool exchange of a
        one-launch RCAB) and which switch selects launches that wait for nobody"""
        if code == 0x4ff or 0x500 <= code < 0x600:
            return ('a halo hand-off of the persistent residual-block chain timed out (code 0x%x: %s); RUMPY_NO_CHAIN=1 selects one launch per block'
                    % (code, 'a neighbour strip never published where it runs' if code == 0x4ff else 'block %d' % (code - 0x500)))
        return ('a strip exchange of the one-launch RCAB kernels timed out (code 0x%x); RUMPY_RCAB_FORM=lazy selects the exchange-free launches, '
                'RUMPY_NO_RCAB=1 the separate ones' % code)

    def degrade(self, plan):
        """A watchdog word of `plan` came back non-zero: the co-residency its waiting launches need did not hold (GPU shared with another job, a CU
        mask).  Switch THIS engine to the launch forms that wait for nobody - the chain -> one launch per block (bitwise the same results), the
        in-launch pool exchange -> the exchange-free RCAB form (or the separate launches where that form does not exist) - and drop the plans built
        with the old forms.  -> text of what was switched, or None when nothing is left to switch (the caller raises)."""
        kinds = self._waiting_kinds(plan)
        done = []
        if 'chain' in kinds and self.use_chain:
            self.use_chain = False
            done.append('persistent block chain -> one launch per block')
        if 'xchg' in kinds and self.use_rcab_kernel:
            if self.rcab_form != 'lazy' and not self.fp8:
                self.rcab_form = 'lazy'
                done.append("RCAB pool exchange -> the exchange-free form ('lazy')")
            else:
                self.use_rcab_kernel = False
                done.append('one-launch RCAB kernels -> separate launches')
        if not done:
            return None
        for k in list(self.plans):
            old = self.plans.pop(k)
            for ops in (old.fwd, getattr(old, 'bwd', None)):
                if ops is not None:
                    self._tables.pop(id(ops), None)
        self.step_status = None
        return '; '.join(done)

    def _upload_q_items(self, plan):
        if plan.q_items:
            arr = (L.QMlpItem * len(plan.q_items))(*plan.q_items)
            raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            if plan.q_dev is None:
                plan.q_dev = torch.empty(raw.size, dtype=torch.uint8, device=self.device)
            plan.q_dev.copy_(torch.from_numpy(raw), non_blocking=False)
        if plan.qn_items:
            arr = (L.QMlpNItem * len(plan.qn_items))(*plan.qn_items)
            raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            if plan.qn_dev is None:
                plan.qn_dev = torch.empty(raw.size, dtype=torch.uint8, device=self.device)
            plan.qn_dev.copy_(torch.from_numpy(raw), non_blocking=False)

    @staticmethod
    def _qn_widths(plan):
        return (C.c_int32 * len(plan.qn_shape))(*plan.qn_shape), len(plan.qn_shape) - 1

    def _q_gates(self, plan, meta, stream, launch=True):
        """metadata [N,M] -> the plan's static metadata buffer, then (launch) the meta-attention gates of every q-layer in one launch"""
        if plan.meta is None:
            return
        if meta is None or tuple(meta.shape) != tuple(plan.meta.shape):
            raise RuntimeError('rumpy_amd: this network needs a metadata matrix of shape %s (got %s)'
                               % (tuple(plan.meta.shape), None if meta is None else tuple(meta.shape)))
        plan.meta.copy_(meta, non_blocking=True)
        if launch:
            self._q_gates_launch(plan, stream)

    def _q_gates_launch(self, plan, stream):
        if not plan.q_items and not plan.qn_items:
            return
        if (plan.q_items and plan.q_dev is None) or (plan.qn_items and plan.qn_dev is None):
            self._upload_q_items(plan)
        if plan.q_items:
            M, Hq = plan.q_shape
            L.check(self.lib.rumpy_q_mlp_fwd(_ptr(plan.q_dev), len(plan.q_items), _ptr(plan.meta), plan.N, M, Hq, self.feats, stream), 'rumpy_q_mlp_fwd')
        if plan.qn_items:
            n, nl = self._qn_widths(plan)
            L.check(self.lib.rumpy_q_mlpn_fwd(_ptr(plan.qn_dev), len(plan.qn_items), _ptr(plan.meta), plan.N, n, nl, stream), 'rumpy_q_mlpn_fwd')

    def _q_param_grads(self, plan, stream):
        if plan.q_items:
            M, Hq = plan.q_shape
            L.check(self.lib.rumpy_q_mlp_bwd_params(_ptr(plan.q_dev), len(plan.q_items), _ptr(plan.meta), plan.N, M, Hq, self.feats, stream),
                    'rumpy_q_mlp_bwd_params')
        if plan.qn_items:
            n, nl = self._qn_widths(plan)
            L.check(self.lib.rumpy_q_mlpn_bwd_params(_ptr(plan.qn_dev), len(plan.qn_items), _ptr(plan.meta), plan.N, n, nl, stream), 'rumpy_q_mlpn_bwd_params')

    def meta_grad(self, plan):
        """d loss / d metadata [N, M] of the backward pass that has just run on `plan` (q-layer networks: the metadata enters through the
        ParaCALayer MLPs only).  Asked for by the autograd node when the metadata itself has a gradient path (a jointly trained encoder)."""
        if not (plan.q_items or plan.qn_items) or plan.qca_items:
            raise RuntimeError('rumpy_amd: the gradient of the metadata input is built for q-layer networks (style "standard" + '
                               'include_q_layer) only')
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if plan.qn_items:
            n, nl = self._qn_widths(plan)
            dmeta = torch.empty(plan.N, plan.qn_shape[0], dtype=torch.float32, device=self.device)
            L.check(self.lib.rumpy_q_mlpn_bwd_meta(_ptr(plan.qn_dev), len(plan.qn_items), plan.N, n, nl, _ptr(dmeta), stream), 'rumpy_q_mlpn_bwd_meta')
            return dmeta
        M, Hq = plan.q_shape
        dmeta = torch.empty(plan.N, M, dtype=torch.float32, device=self.device)
        L.check(self.lib.rumpy_q_mlp_bwd_meta(_ptr(plan.q_dev), len(plan.q_items), plan.N, M, Hq, self.feats, _ptr(dmeta), stream), 'rumpy_q_mlp_bwd_meta')
        return dmeta

    # ------------------------------------------------------------------ execution
    def plan_for(self, N, H, W, train, fmt=0):
        """the cached plan of a shape.  Training plans are kept; evaluation plans (one per image size, each owning its activation
        buffers - hundreds of MB for a full-size image) live in an LRU of `max_eval_plans`."""
        key = (N, H, W, bool(train), int(fmt))
        p = self.plans.pop(key, None)
        if p is None:
            p = self._build(N, H, W, bool(train), int(fmt))
        self.plans[key] = p                 # most recently used last
        if not train:
            ev = [k for k in self.plans if not k[3]]
            for k in ev[:max(0, len(ev) - self.max_eval_plans)]:
                old = self.plans.pop(k)
                for ops in (old.fwd, old.bwd):
                    self._tables.pop(id(ops), None)
        return p

    def _run(self, ops, stream):
        """Launch a list of (entry point name, argument block) in ONE library call (rumpy_run_list walks the table in C: the
        per-launch ctypes cost, 4-5 us, was most of the host time of a step).  The table is rebuilt when the list object changes."""
        if not ops:
            return
        cache = self._tables.get(id(ops))
        if cache is None or cache[0] is not ops or cache[3] != len(ops):
            arr = (L.Op * len(ops))()
            for i, (name, a) in enumerate(ops):
                arr[i].fn = C.cast(getattr(self.lib, name), C.c_void_p).value
                arr[i].args = C.addressof(a)
            cache = (ops, arr, C.addressof(arr), len(ops))
            self._tables[id(ops)] = cache
        rc = self.lib.rumpy_run_list(cache[2], cache[3], stream)
        if rc != 0:      # -(index + 1) of the entry that failed; its message is in rumpy_last_error()
            L.check(rc, ops[-rc - 1][0] if (rc < 0 and -rc - 1 < len(ops)) else 'rumpy_run_list')

    def forward(self, x, train=False, target=None, meta=None):
        """x (and target): contiguous fp32 [N,C,H,W] on the device; meta: fp32 [N,M] metadata (meta-attention nets only).
        Returns (out fp32 [N,C,sH,sW], loss tensor | None, plan)."""
        N, _, H, W = x.shape
        if not train:
            self.check_eval()
        fmt = 0 if train else self.eval_fmt
        plan = self.plan_for(N, H, W, train, fmt)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if fmt:
            self._repack_h(stream)
        plan.gen += 1
        if train:
            self.step_status = plan.rcab_status
        self._q_gates(plan, meta, stream)
        self._advance_epoch(plan, stream)
        # the head / tail kernels read the caller's fp32 NCHW tensors in place and write a fresh output tensor: no copies
        plan.x_ref, plan.target_ref = x, target           # keep them alive until the backward pass has consumed them
        if train:                                         # (two generations: the backward pass of step i may still run when step i + 1 is queued, _bind_batch)
            refs = getattr(plan, 'batch_refs', None)
            if refs is None:
                refs = plan.batch_refs = collections.deque(maxlen=2)
            refs.append((x, target))
        plan.head_args.x = x.data_ptr()
        plan.head_args.x_ind = None           # (a captured step of the same plan reads through its pointer table; its launches keep their own copy of the arguments)
        if train:
            plan.head_wgrad_args.x = x.data_ptr()
            plan.head_wgrad_args.x_ind = None
            if getattr(plan, 'tail_loss', None) is not None:
                plan.tail_loss.target_ind = None
        out = torch.empty_like(plan.out)
        if train and plan.f8_f_n:
            self._repack_f8(stream)
            self._f8_begin(plan, 'f', plan.fwd, stream)
        self._run(plan.fwd, stream)
        if self.wide:
            if target is not None:
                raise RuntimeError('rumpy_amd: the fused L1 pass is a 64-feature kernel; wide nets take the generic loss path')
            plan.tail_wide.out = out.data_ptr()
            L.call('rumpy_tail_fwd_wide', plan.tail_wide, stream)
            loss = None
        elif target is not None:
            plan.tail_loss.out = out.data_ptr()
            plan.tail_loss.target = target.data_ptr()
            L.call('rumpy_tail_fwd', plan.tail_loss, stream)
            plan.tail_fused = plan.tail_wslab is not None      # the tail weight gradient w.r.t. the L1 loss now sits in tail_wslab
            loss = plan.loss
        else:
            plan.tail_plain.out = out.data_ptr()
            plan.tail_fused = False
            L.call('rumpy_tail_fwd', plan.tail_plain, stream)
            loss = None
        if not train and self.eval_defer:
            # the caller keeps the output on the device (run_eval(keep_on_device=True), validation loops): the status words are copied into
            # pinned memory behind the pass and fenced by an event - no host synchronisation here; they are examined at the next
            # evaluation pass or by check_eval() (a non-finite fp16 pass then switches the FOLLOWING passes to bf16: its own output has
            # been handed out already, non-finite values included)
            if self._flag_host is None:
                self._flag_host = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._flag_host.copy_(plan.flags, non_blocking=True)
            plan.flags.zero_()
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._flag_pending = (ev, fmt)
        elif not train:
            # evaluation passes read both status words back here (one small copy; it synchronises - these callers fetch the image next
            # anyway), so that neither problem can go unnoticed on any return path of the handlers
            bad, xch = plan.flags.tolist()
            if xch:
                # an evaluation pass has no side effects: switch to the launches that wait for nobody and run it again
                plan.flags.zero_()
                what = self.degrade(plan)
                if what is None:
                    raise RuntimeError('rumpy_amd: %s; the evaluation output is invalid (GPU shared with another job?)' % self.watchdog_text(xch))
                import warnings
                warnings.warn('rumpy_amd: %s (GPU shared with another job?) - this engine continues with: %s; the pass is run again'
                              % (self.watchdog_text(xch), what), RuntimeWarning)
                return self.forward(x, train=False, target=target, meta=meta)
            if bad and fmt:             # fp16 overflowed somewhere in this network: bf16 has fp32's range
                import warnings
                warnings.warn('rumpy_amd: an fp16 evaluation pass produced a non-finite output; evaluation of this network continues in bf16')
                plan.flags.zero_()
                self.eval_fmt = L.FMT_BF16
                return self.forward(x, train=False, target=target, meta=meta)
            if bad:
                plan.flags.zero_()      # bf16 plan: the fp32 reference would not be finite either; nothing to fall back to
        return out, loss, plan

    def check_eval(self):
        """examine the status words of a deferred evaluation pass (eval_defer): waits for that pass only.  Raises on a strip-exchange
        time-out; a non-finite fp16 output switches the engine to bf16 evaluation plans from here on."""
        pend, self._flag_pending = self._flag_pending, None
        if pend is None:
            return
        ev, fmt = pend
        ev.synchronize()
        bad, xch = self._flag_host.tolist()
        if xch:
            raise RuntimeError('rumpy_amd: %s; the output of the previous evaluation pass (kept on the device, unchecked) is invalid '
                               '(GPU shared with another job?)' % self.watchdog_text(xch))
        if bad and fmt and self.eval_fmt != L.FMT_BF16:
            import warnings
            warnings.warn('rumpy_amd: the previous fp16 evaluation pass produced a non-finite output (it was kept on the device, unchecked); '
                          'evaluation of this network continues in bf16')
            self.eval_fmt = L.FMT_BF16

    def backward(self, plan, grad_scale, gout=None, on_ready=None):
        """Run the backward pass of the last training forward of `plan`.  gout: optional upstream gradient
        [N,C,sH,sW] fp32 (replaces the fused sign gradient).  on_ready(ptr): called on the host once every launch that writes a
        gradient at a device address >= ptr has been queued (data-parallel runs start the all-reduce of that part there and overlap it
        with the remaining weight-gradient launches)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if gout is not None:
            h, w = plan.HR
            L.call('rumpy_nchw_to_nhwc4', L.NchwToNhwc4Args(src=_ptr(gout), dst=_ptr(plan.dy4), N=plan.N, C=gout.shape[1], H=h, W=w), stream)
        self._set_grad_scale(plan, float(grad_scale))
        tail_done = plan.tail_fused and gout is None      # an upstream gradient replaces the sign gradient: separate pass then
        plan.tail_fused = False
        self._f8_begin(plan, 'b', plan.bwd, stream)
        self._run(plan.bwd, stream)
        self._ca_param_grads(plan, stream)
        self._q_param_grads(plan, stream)
        self._qca_param_grads(plan, stream)
        gs = float(grad_scale)
        if on_ready is not None and plan.split is not None:
            sp = plan.split
            self._wgrad4(plan, stream, 'a')
            if 1 in plan.job_dev and not tail_done:
                dev, n = plan.job_dev[1]
                L.check(self.lib.rumpy_wgrad_grouped(_ptr(dev), n, 1, 1 if plan.HR[1] % 2 else 0, stream), 'rumpy_wgrad_grouped')
            idx = sp['idx_a_notail'] if tail_done else sp['idx_a']
            self._reduce(plan, stream, sp['red_a_notail'] if tail_done else sp['red_a'], len(idx), gs, tail=tail_done, head=False)
            on_ready(sp['ptr'])
            self._wgrad4(plan, stream, 'b')
            self._reduce(plan, stream, sp['red_b'], len(sp['idx_b']), gs, tail=False, head=True)
            return
        if 4 in plan.job_dev:
            self._wgrad4(plan, stream)
        if 1 in plan.job_dev and not tail_done:
            dev, n = plan.job_dev[1]
            L.check(self.lib.rumpy_wgrad_grouped(_ptr(dev), n, 1, 1 if plan.HR[1] % 2 else 0, stream), 'rumpy_wgrad_grouped')      # dy4 pixel-pair DMA needs an even width
        if tail_done:
            self._reduce(plan, stream, plan.reduce_dev_notail, len(plan.reduce_keep), gs, tail=True, head=True)
        else:
            self._reduce(plan, stream, plan.reduce_dev, plan.n_reduce, gs, tail=False, head=True)

    def _wgrad4(self, plan, stream, group=None):
        """the weight-gradient launch of the 64-channel layers: all of them, or group 'a' / 'b' of a two-phase plan"""
        dev, n = plan.job_dev[4]
        if plan.shares is not None:
            lo, cnt = (0, plan.shares['n']) if group is None else plan.split['shares_' + group]
            L.check(self.lib.rumpy_wgrad_shares(_ptr(dev), plan.shares['first'].data_ptr() + 4 * lo, cnt, stream), 'rumpy_wgrad_shares')
        elif group is None:
            L.check(self.lib.rumpy_wgrad_grouped(_ptr(dev), n, 4, 0, stream), 'rumpy_wgrad_grouped')
        else:
            gd, gn = plan.split['jobs_' + group]
            L.check(self.lib.rumpy_wgrad_grouped(_ptr(gd), gn, 4, 0, stream), 'rumpy_wgrad_grouped')

    def set_two_phase(self):
        """from now on training plans cut the weight-gradient shares per gradient-buffer half (data-parallel runs); existing ones are dropped"""
        if self.wgrad_two_phase or not self.wgrad_shares:
            return
        self.wgrad_two_phase = True
        for k in [k for k in self.plans if k[3]]:
            old = self.plans.pop(k)
            for ops in (old.fwd, old.bwd):
                self._tables.pop(id(ops), None)

    def _reduce(self, plan, stream, items_dev, nitems, grad_scale, tail, head):
        """Slab reductions -> parameter gradients: `nitems` entries of the grouped weight-gradient table, the fused tail conv's slabs
        (tail) and the head conv's (head).  One launch (rumpy_finish_reduce); RUMPY_NO_FINISH=1: the separate entry points (A/B)."""
        tl, hd = self.spec.tail, self.spec.head
        if self.use_finish:
            a = L.FinishReduceArgs(items=_ptr(items_dev) if nitems else None, nitems=nitems)
            if tail:
                a.tail_slabs, a.tail_nslabs, a.tail_C, a.tail_scale = _ptr(plan.tail_wslab), plan.tail_slabs, tl.cout, grad_scale
                a.tail_gw, a.tail_gb = _ptr(tl.gw), _ptr(tl.gb)
            if head:
                a.head_slabs, a.head_nslabs, a.head_C, a.head_cout, a.head_scale = _ptr(plan.head_slab), plan.head_nslabs, hd.cin, hd.cout, grad_scale
                a.head_gw, a.head_gb = _ptr(hd.gw), _ptr(hd.gb)
            L.call('rumpy_finish_reduce', a, stream)
            return
        if nitems:
            L.check(self.lib.rumpy_wgrad_reduce(_ptr(items_dev), nitems, stream), 'rumpy_wgrad_reduce')
        if tail:
            L.check(self.lib.rumpy_tail_wgrad_reduce(_ptr(plan.tail_wslab), plan.tail_slabs, tl.cout, grad_scale, _ptr(tl.gw), _ptr(tl.gb), stream),
                    'rumpy_tail_wgrad_reduce')

    # ------------------------------------------------------------------ hipGraph replay of the fused L1 training pass
    def train_pass_graphed(self, x, target, meta=None):
        """forward + L1 + full backward of one batch as ONE captured hipGraph (the per-step launch list is static):
        ~150 kernel launches collapse into a graph replay, which removes the host launch gaps between the short
        per-layer kernels.  Inputs are copied into the plan's static buffers; the returned `out` / `loss` tensors are
        the plan's static buffers (valid until the next call).  Returns (out, loss, plan)."""
        N, _, H, W = x.shape
        if self.fp8:
            raise RuntimeError("rumpy_amd: precision 'fp8' runs eager steps (its first pass measures the scales; RUMPY_GRAPH=1 is not supported)")
        plan = self.plan_for(N, H, W, True)
        cur = torch.cuda.current_stream(self.device)
        if getattr(plan, 'graph', None) is None:
            self._set_grad_scale(plan, 1.0 / plan.out.numel())
            plan.head_args.x = plan.x_in.data_ptr()
            plan.head_wgrad_args.x = plan.x_in.data_ptr()
            plan.tail_loss.out = plan.out.data_ptr()
            plan.tail_loss.target = plan.target.data_ptr()
            if self.batch_by_pointer:
                # the captured launches read x / target through a two-word device table (rumpy_set_pointers before every replay):
                # no device-to-device copy of the batch into the plan's buffers (14 MB of target per step on the headline shape)
                plan.batch_ptrs = torch.zeros(2, dtype=torch.int64, device=self.device)
                plan.head_args.x_ind = plan.batch_ptrs.data_ptr()
                plan.head_wgrad_args.x_ind = plan.batch_ptrs.data_ptr()
                plan.tail_loss.target_ind = plan.batch_ptrs.data_ptr() + 8
            self._bind_batch(plan, x, target, cur)
            self._q_gates(plan, meta, None, launch=False)

            if (plan.q_items and plan.q_dev is None) or (plan.qn_items and plan.qn_dev is None):
                self._upload_q_items(plan)          # not inside the capture

            def body(stream):
                self._advance_epoch(plan, stream)
                self._q_gates_launch(plan, stream)  # reads the plan's static metadata buffer (refreshed before every replay)
                self._run(plan.fwd, stream)
                L.call('rumpy_tail_fwd', plan.tail_loss, stream)
                self._backward_launches(plan, stream)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):           # warm-up run outside capture (lazy initialisation, allocator)
                body(side.cuda_stream)
            cur.wait_stream(side)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body(torch.cuda.current_stream(self.device).cuda_stream)
            plan.graph = g
        self._bind_batch(plan, x, target, cur)
        self._q_gates(plan, meta, None, launch=False)
        plan.graph.replay()
        return plan.out, plan.loss, plan

    def _bind_batch(self, plan, x, target, cur):
        """make the captured step read this batch: by pointer when the caller's tensors are what the kernels read (fp32, contiguous, on
        this device - the caller keeps them alive until the step has run, as for any stream-ordered op), by copy otherwise"""
        def direct(t, like):
            return (t.is_cuda and t.device == like.device and t.dtype == torch.float32 and t.is_contiguous() and t.shape == like.shape)
        if getattr(plan, 'batch_ptrs', None) is None:
            plan.x_in.copy_(x, non_blocking=True)
            plan.target.copy_(target, non_blocking=True)
            return
        if not direct(x, plan.x_in):
            plan.x_in.copy_(x, non_blocking=True)
            x = plan.x_in
        if not direct(target, plan.target):
            plan.target.copy_(target, non_blocking=True)
            target = plan.target
        # the replay reads the caller's tensors in place (head_wgrad reads x at the very end of the backward pass): a batch dropped by the
        # caller right after run_train must not be recycled by the caching allocator under the running replay.  Two
        # generations are held: run_train returns behind the forward pass of step i (loss read-back), so when batch i + 1 is bound the
        # backward pass of step i may still run, and it is over when batch i + 2 is bound (same stream, behind forward i + 1).
        plan.x_ref, plan.target_ref = x, target
        refs = getattr(plan, 'batch_refs', None)
        if refs is None:
            refs = plan.batch_refs = collections.deque(maxlen=2)
        refs.append((x, target))
        L.check(self.lib.rumpy_set_pointers(plan.batch_ptrs.data_ptr(), x.data_ptr(), target.data_ptr(), cur.cuda_stream), 'rumpy_set_pointers')

    def _qca_param_grads(self, plan, stream):
        if plan.qca_items:
            L.check(self.lib.rumpy_qca_bwd_params(_ptr(plan.qca_dev), len(plan.qca_items), stream), 'rumpy_qca_bwd_params')

    def _ca_param_grads(self, plan, stream):
        if plan.ca_param_items:
            a0 = plan.ca_param_items[0]
            L.check(self.lib.rumpy_ca_mlp_bwd_params(_ptr(plan.ca_params_dev), len(plan.ca_param_items), a0.N, a0.C, a0.Cr, stream),
                    'rumpy_ca_mlp_bwd_params')

    def _backward_launches(self, plan, stream):
        """backward launch list of the fused L1 pass (captured into the hipGraph): the tail weight gradient came from tail_fwd"""
        self._run(plan.bwd, stream)
        self._ca_param_grads(plan, stream)
        self._q_param_grads(plan, stream)
        self._qca_param_grads(plan, stream)
        if 4 in plan.job_dev:
            self._wgrad4(plan, stream)
        self._reduce(plan, stream, plan.reduce_dev_notail, len(plan.reduce_keep), float(plan.grad_scale), tail=True, head=True)

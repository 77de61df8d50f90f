"""Degradation encoder of the blind-SR pipeline on the MI355X path - mirror of
rumpy/regression/models/contrastive_learning/encoding_models.py:5-55 (``Encoder``, the DASR encoder: six 3x3 convs
3-64-64-128/2-128-256/2-256 with BatchNorm + LeakyReLU(0.1), global average pool, ``mlp`` 256-256-256).

The torch modules below only OWN the parameters and buffers under the reference's keys (``E.{0,1,3,4,...}.*``, ``mlp.{0,2}.*``) and
creation order; the convolutional trunk runs hand-written HIP kernels and fails loudly without a GPU:

* inference (the blind-SR handlers keep the encoder frozen, ``encoder_freeze_mode='all'``): rumpy_head_fwd, rumpy_enc_conv, rumpy_enc_pool,
  with both BatchNorm modes because the reference uses both - running statistics under ``.eval()`` (folded into the filters, one launch
  per layer) and batch statistics + running-statistics update under ``.train()``, which is the mode the reference's run_train leaves the
  frozen encoder in (base_architecture.py:472);
* training (MoCo / SupMoCo, moco.py / supmoco.py): ``forward`` under ``.train()`` with trainable trunk parameters is one autograd node whose
  backward pass is HIP as well - rumpy_enc_bn_bwd per stage (BatchNorm + LeakyReLU + the pool's backward), rumpy_enc_conv on the filters'
  dgrad images for the data gradients (a stride-2 layer's gradient is written at every second pixel of a zeroed stride-1 grid), the SR
  path's rumpy_wgrad_grouped / rumpy_wgrad_reduce and rumpy_head_wgrad for the weight gradients.  Parameters then live in ONE flat fp32
  buffer with a flat gradient buffer beside it (``flatten()``), which the fused Adam launch and the key encoder's momentum update walk.

The ``mlp`` head (two 256 x 256 linear layers on an [N, 256] matrix) is plain torch (rocBLAS) with torch autograd."""
import ctypes as C
import os

import numpy as np
import torch
from torch import nn

from rumpy_amd import _lib as L

BF16 = torch.bfloat16
LAYERS = ((3, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1))
SLOPE = 0.1


def _ptr(t):
    return None if t is None else t.data_ptr()


def _dev_bytes(ctypes_array, dev):
    return torch.from_numpy(np.frombuffer(bytes(ctypes_array), dtype=np.uint8).copy()).to(dev)


class _LaunchGraph:
    """A fixed launch list over fixed buffers: eager the first time (which also warms every lazily initialised piece up), captured as a
    hipGraph the second, replayed from then on - the trunk is ~20 (forward) / ~30 (backward) short launches, and issuing them one by one
    from Python takes longer than they run.  RUMPY_ENC_GRAPH=0 keeps it eager (A/B, debugging)."""
    enabled = os.environ.get('RUMPY_ENC_GRAPH', '1') != '0'

    def __init__(self):
        self.calls, self.graph = 0, None

    def run(self, launches, dev):
        self.calls += 1
        if not self.enabled or self.calls == 1 or torch.cuda.is_current_stream_capturing():
            launches()
            return
        if self.graph is None:
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                launches()
            self.graph = g
        self.graph.replay()


# Storage format of the TRAINING forward pass (filter images, conv outputs z, stage outputs a): IEEE fp16 since round 3.  The gradient of this
# network is dominated by LeakyReLU(0.1) inputs that change sign under the storage rounding (a tenfold change of that element's gradient):
# with 8-bit mantissas (bf16) the trunk's gradient is 5-10 % off the fp32 reference's, with fp16's 11 bits 2-3 % (CPU simulation of the
# rounding points and GPU agree).  Values after BatchNorm are O(1): far inside fp16's range.  The backward pass keeps
# bf16 (gradient range): the weight-gradient kernels read a bf16 copy of every stage output, written by the same BatchNorm apply pass.
# RUMPY_ENC_TRAIN_BF16=1: the round-2 all-bf16 forward pass (A/B).
TRAIN_FMT = L.FMT_BF16 if os.environ.get('RUMPY_ENC_TRAIN_BF16') == '1' else L.FMT_F16
TRAIN_DT = torch.float16 if TRAIN_FMT == L.FMT_F16 else BF16
# Round 4: of the three 11-bit roundings of that fp16 pass (filter, conv output z, stage output a) the filter's and z's carry most of what is
# left - a CPU simulation of the rounding points on five seeded cases (tests/tools/enc_storage_sim.py): all fp16 -> worst parameter tensor
# 5.4-7.7e-2 from the fp32 graph; fp32 z alone 3.0-7.6e-2 (not robust); unrounded filter alone: no gain; BOTH: 3.1-3.9e-2.  So the training
# forward pass keeps every conv output z as fp32 (read by the BatchNorm statistics / apply passes and once more by the backward pass: +2 bytes
# per element on three streaming reads) and runs its convs on the filter AND its rounding-residual image (w = fp16(w) + fp16(w - fp16(w)):
# twice the forward MFMAs of a network whose step is launch- and bandwidth-bound).  RUMPY_ENC_TRAIN_Z16=1: the round-3 all-fp16 pass (A/B).
TRAIN_Z32 = TRAIN_FMT == L.FMT_F16 and os.environ.get('RUMPY_ENC_TRAIN_Z16') != '1'

_SR_CONVS = os.environ.get('RUMPY_ENC_OWN_CONV') != '1'      # A/B: the encoder's own kernel for every launch


def _conv_plain(x, w, bias, out, N, H, W, cin, cout, stream, fmt=0):
    """stride-1 3x3 conv without activation (a training-mode stage, or a data gradient): the SR path's kernels where they take the shape and are
    faster - Cin = 64 (strip kernel: 43 vs 95 us for 64 -> 64 at 256 x 48 x 48) and Cin = 256 on maps of 24+ rows (128 vs 188 us for 256 -> 128 at
    256 x 24 x 24; tests/tools/enc_conv_ab.py) -, else the encoder's general kernel.  Same filter images, same bf16 NHWC layout."""
    if _SR_CONVS and (cin == 64 or (cin == 256 and H >= 24)):
        L.call('rumpy_conv3x3', L.ConvArgs(x=x, w=w, bias=bias, out=out, mask=None, res1=None, res2=None, pool=None, N=N, H=H, W=W,
                                           cin_chunks=cin // 64, cout_tiles=cout // 64, in_mode=0, out_mode=0, relu=0, scale=1.0, grid_x=0, fmt=fmt), stream)
    else:
        L.call('rumpy_enc_conv', L.EncConvArgs(x=x, w=w, bias=bias, out=out, N=N, H=H, W=W, cin=cin, cout=cout, stride=1, neg_slope=1.0, fmt=fmt), stream)


class _TrunkFn(torch.autograd.Function):
    """The six conv + BatchNorm(train) + LeakyReLU stages and the pool as ONE autograd node: both directions are HIP launches.  Parameter
    gradients are written (not accumulated) into the encoder's flat gradient buffer, whose views are the parameters' ``.grad``."""

    @staticmethod
    def forward(ctx, x, enc, *params):
        fea, ctx.token, ctx.x = enc._train_forward(x)
        ctx.enc = enc
        ctx.nparams = len(params)
        return fea

    @staticmethod
    def backward(ctx, dfea):
        ctx.enc._train_backward(ctx.x, ctx.token, dfea)
        return (None, None) + (None,) * ctx.nparams


class _RepackHook:
    """what FlatAdam expects of ``net.engine``: the fused update rewrote flat_p, so the bf16 filter images are stale"""

    def __init__(self, encoder):
        self.encoder = encoder

    def repack(self, stream=None):
        self.encoder.weights_rewritten()

    def exchange_status(self):
        return 0


class Encoder(nn.Module):
    # a flattened encoder as a handler's whole net (SupConHandler): the attributes BaseModel / FlatAdam look for (after ``flatten()``)
    flat_protocol = True
    supports_fused_l1 = False
    use_graph = False
    _evictions = 0                   # bumped whenever cached plans (and with them buffers a captured graph may hold) are dropped

    def __init__(self, dropdown_q=None):
        super(Encoder, self).__init__()
        if dropdown_q is not None:
            raise RuntimeError('rumpy_amd: the encoder drop-down head is outside the MI355X hot path')
        mods = []
        for cin, cout, stride in LAYERS:
            mods += [nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1), nn.BatchNorm2d(cout), nn.LeakyReLU(SLOPE, True)]
        mods.append(nn.AdaptiveAvgPool2d(1))
        self.E = nn.Sequential(*mods)
        self.mlp = nn.Sequential(nn.Linear(256, 256), nn.LeakyReLU(SLOPE, True), nn.Linear(256, 256))
        self.dropdown = False
        self._plans = {}
        self._packed = None          # (key, raw filter images)
        self._folded = None          # (key, folded filter images)
        self._stats_epoch = 0        # bumped whenever a kernel rewrites the running statistics
        self._train_plans = {}
        self._train_images = None    # (key, [(w_fwd, w_dgrad, b_packed)] of convs 1..5, keep-alive)
        self._img_store = {}         # slot -> persistent image buffers + pack table (captured graphs hold their addresses)
        self._token = 0              # forward passes of the training plan: a backward pass must belong to the last one
        self.flat_p = self.flat_g = None
        self.param_list, self.offsets, self.grad_views = None, None, None

    # ------------------------------------------------------------------ flat parameter storage (training)
    def flatten(self):
        """Move every parameter into one flat fp32 buffer (parameters become views, module order) with a flat gradient buffer beside it."""
        self.param_list = list(self.parameters())
        dev = self.param_list[0].device
        total = sum(p.numel() for p in self.param_list)
        flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets, off = [], 0
        with torch.no_grad():
            for p in self.param_list:
                n = p.numel()
                flat_p[off:off + n].copy_(p.data.reshape(-1).float())
                p.data = flat_p[off:off + n].view(p.shape)
                self.offsets.append(off)
                off += n
        self.flat_p = flat_p
        self.grad_views = [self.flat_g[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, self.param_list)]
        self._packed = self._folded = self._train_images = None
        self._train_plans, self._plans, self._img_store = {}, {}, {}       # captured graphs hold the old storages' addresses
        self.attach_grads()

    def attach_grads(self):
        """Gradient views as ``.grad``.  The trunk's are overwritten by every backward pass; the mlp head's are accumulated into by torch
        autograd, so they start every step from zero."""
        n_trunk = 24                          # 6 x (conv weight, bias, BatchNorm weight, bias) come first in module order
        for i, (p, g) in enumerate(zip(self.param_list, self.grad_views)):
            if p.requires_grad:
                if i >= n_trunk:
                    g.zero_()
                p.grad = g

    def _apply(self, fn, recurse=True):
        was_flat = self.flat_p is not None
        super()._apply(fn)
        if was_flat:                          # .to(device) replaced the parameter storages
            self.flatten()
        else:
            self._packed = self._folded = self._train_images = None
            self._train_plans, self._plans, self._img_store = {}, {}, {}
        return self

    def _ensure_engine(self):
        if self.flat_p is None or not self.flat_p.is_cuda:
            raise RuntimeError('rumpy_amd: this network only runs on an MI355X through the HIP extension; there is no CPU path')

    @property
    def engine(self):
        return _RepackHook(self)

    def mark_weights_clean(self):
        pass

    def mark_weights_updated(self):
        self.weights_rewritten()

    def weights_rewritten(self):
        """flat_p was rewritten by a kernel (fused Adam, the momentum update): the bf16 filter images are stale."""
        self._packed = self._folded = self._train_images = None

    # ------------------------------------------------------------------ filter images
    def _convs(self):
        c = self.__dict__.get('_conv_cache')
        if c is None or c[0] is not self.E:               # nn.Sequential indexing is slow enough to show in a 2 ms step
            mods = list(self.E.children())
            c = self.__dict__['_conv_cache'] = (self.E, [mods[3 * i] for i in range(6)], [mods[3 * i + 1] for i in range(6)])
        return c[1], c[2]

    def train(self, mode=True):
        """nn.Module.train walks every sub-module; the handlers call it at every step (base_architecture.py:472) - skipped when nothing changes"""
        if self.training == mode and self.E.training == mode and self.mlp.training == mode and all(b.training == mode for b in self._convs()[1]):
            return self
        return super().train(mode)

    def _key(self, with_stats):
        convs, bns = self._convs()
        ts = [t for c in convs for t in (c.weight, c.bias)]
        if with_stats:
            ts += [t for b in bns for t in (b.weight, b.bias, b.running_mean, b.running_var)]
        return tuple(t._version for t in ts) + tuple(t.data_ptr() for t in ts) + ((self._stats_epoch,) if with_stats else ())

    def _pack(self, weights, biases, dev, dgrad=False, slot='raw', fwd_fmt=0):
        """[(w fp32 OIHW, b)] of the five 64-multiple convs -> MFMA fragment images through rumpy_pack_weights
        (dgrad: also the transposed + flipped image the data gradient convolves with).  The images of a slot live in persistent buffers -
        captured launch graphs read them - and so does the item table as long as the sources' addresses stay."""
        st = self._img_store.get(slot)
        if st is None or st['dev'] != dev:
            out = []
            for w in weights:
                cout, cin = w.shape[:2]
                wf = torch.empty(cout * cin * 9, dtype=torch.float16 if fwd_fmt == L.FMT_F16 else BF16, device=dev)
                wd = torch.empty(cout * cin * 9, dtype=BF16, device=dev) if dgrad else None
                bp = torch.empty(cout, dtype=torch.float32, device=dev)
                if dgrad and fwd_fmt == L.FMT_F16 and TRAIN_Z32:      # + the fp16 filter's rounding-residual image (rumpy_enc_conv.w_lo)
                    out.append((wf, wd, bp, torch.empty(cout * cin * 9, dtype=torch.float16, device=dev)))
                else:
                    out.append((wf, wd, bp) if dgrad else (wf, bp))
            st = self._img_store[slot] = dict(dev=dev, imgs=out, src=None, tab=None)
        src = tuple(_ptr(t) for t in list(weights) + list(biases))
        if st['src'] != src:
            if fwd_fmt and dgrad:      # forward image in fp16, data-gradient image in bf16: two items per conv
                items = [L.PackItem(w=_ptr(w), b=_ptr(b), w_fwd=_ptr(im[0]), w_dgrad=None, b_packed=_ptr(im[2]), cout=w.shape[0], cin=w.shape[1], kind=0,
                                    shuffle=0, fmt=fwd_fmt) for w, b, im in zip(weights, biases, st['imgs'])] + \
                        [L.PackItem(w=_ptr(w), b=_ptr(b), w_fwd=None, w_dgrad=_ptr(im[1]), b_packed=None, cout=w.shape[0], cin=w.shape[1], kind=0,
                                    shuffle=0) for w, b, im in zip(weights, biases, st['imgs'])] + \
                        [L.PackItem(w=_ptr(w), b=_ptr(b), w_fwd=_ptr(im[3]), w_dgrad=None, b_packed=None, cout=w.shape[0], cin=w.shape[1], kind=0,
                                    shuffle=0, fmt=L.FMT_F16_RESIDUAL) for w, b, im in zip(weights, biases, st['imgs']) if len(im) > 3]
            else:
                items = [L.PackItem(w=_ptr(w), b=_ptr(b), w_fwd=_ptr(im[0]), w_dgrad=_ptr(im[1]) if dgrad else None, b_packed=_ptr(im[-1]),
                                    cout=w.shape[0], cin=w.shape[1], kind=0, shuffle=0, fmt=fwd_fmt) for w, b, im in zip(weights, biases, st['imgs'])]
            st['tab'], st['src'], st['n'] = _dev_bytes((L.PackItem * len(items))(*items), dev), src, len(items)
        L.check(L.lib().rumpy_pack_weights(_ptr(st['tab']), st['n'], torch.cuda.current_stream(dev).cuda_stream), 'rumpy_pack_weights')
        return st['imgs'], (st['tab'], weights, biases)

    def _raw_images(self, dev):
        key = self._key(False)
        if self._packed is None or self._packed[0] != key:
            convs, _ = self._convs()
            ws = [c.weight.detach().float().contiguous() for c in convs]
            bs = [c.bias.detach().float().contiguous() for c in convs]
            imgs, keep = self._pack(ws[1:], bs[1:], dev, fwd_fmt=TRAIN_FMT)       # batch-statistics mode = the training forward's storage format
            self._packed = (key, [(ws[0], bs[0])] + imgs, keep)
        return self._packed[1]

    def _folded_images(self, dev):
        """eval-mode BatchNorm folded into filter and bias: w' = w * g / sqrt(var + eps), b' = (b - mean) * g / sqrt(var + eps) + beta"""
        key = self._key(True)
        if self._folded is None or self._folded[0] != key:
            convs, bns = self._convs()
            ws, bs = [], []
            with torch.no_grad():
                for c, bn in zip(convs, bns):
                    s = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
                    ws.append((c.weight.float() * s[:, None, None, None]).contiguous())
                    bs.append(((c.bias.float() - bn.running_mean.float()) * s + bn.bias.float()).contiguous())
            imgs, keep = self._pack(ws[1:], bs[1:], dev, slot='folded')
            head = self._img_store['folded'].setdefault('head', (torch.empty_like(ws[0]), torch.empty_like(bs[0])))
            head[0].copy_(ws[0])
            head[1].copy_(bs[0])
            self._folded = (key, [head] + imgs, keep)
        return self._folded[1]

    # ------------------------------------------------------------------ execution
    def _plan(self, N, H, W, dev):
        k = (N, H, W, dev.index)
        p = self._plans.get(k)
        if p is None:
            acts, h, w = [], H, W
            for cin, cout, stride in LAYERS:
                h, w = (h - 1) // stride + 1, (w - 1) // stride + 1
                acts.append(torch.empty(N, h, w, cout, dtype=BF16, device=dev))
            part = torch.empty(max(int(L.lib().rumpy_enc_bn_partial_floats(a.shape[0] * a.shape[1] * a.shape[2], a.shape[3])) for a in acts),
                               dtype=torch.float32, device=dev)
            p = dict(acts=acts, partial=part, scale_shift=torch.empty(2 * 256, dtype=torch.float32, device=dev),
                     x=torch.empty(N, 3, H, W, dtype=torch.float32, device=dev), fea=torch.empty(N, 256, dtype=torch.float32, device=dev),
                     graphs={}, gkey={})
            if len(self._plans) > 8:
                self._plans.clear()
                self._evictions += 1         # whoever captured launches over these buffers (a handler's step graph) must drop them
            self._plans[k] = p
        return p

    def features(self, x):
        """x [N,3,H,W] fp32 on the GPU -> fea [N,256] fp32; BatchNorm mode follows ``self.training`` like the torch modules would."""
        if not x.is_cuda:
            raise RuntimeError('rumpy_amd: the degradation encoder runs on the GPU only (no CPU fallback)')
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError('rumpy_amd: encoder input must be [N,3,H,W]')
        dev = x.device
        N, _, H, W = x.shape
        train = self.training
        plan = self._plan(N, H, W, dev)
        imgs = self._raw_images(dev) if train else self._folded_images(dev)
        _, bns = self._convs()
        acts = plan['acts']
        w0, b0 = imgs[0]
        xs, fea = plan['x'], plan['fea']
        xs.copy_(x)
        graphable = all(bn.momentum is not None for bn in bns)      # a cumulative average changes its factor every step
        # batch-statistics mode (a MoCo key encoder, the blind pipeline's frozen encoder under net.train()) stores like the training forward pass:
        # fp16 (same 2-byte elements: the plan's buffers serve both modes); evaluation mode (BatchNorm folded into the filters) stays bf16
        fmt = TRAIN_FMT if train else L.FMT_BF16

        def launches():
            stream = torch.cuda.current_stream(dev).cuda_stream
            L.call('rumpy_head_fwd', L.HeadFwdArgs(x=_ptr(xs), w=_ptr(w0), b=_ptr(b0), out=_ptr(acts[0]), N=N, C=3, H=H, W=W, cout=64,
                                                   neg_slope_m1=0.0 if train else SLOPE - 1.0, fmt=fmt), stream)
            h, w = H, W
            for i, (cin, cout, stride) in enumerate(LAYERS):
                if i > 0:
                    wf, bp = imgs[i]
                    if train and stride == 1:
                        _conv_plain(_ptr(acts[i - 1]), _ptr(wf), _ptr(bp), _ptr(acts[i]), N, h, w, cin, cout, stream, fmt=fmt)
                    else:
                        L.call('rumpy_enc_conv', L.EncConvArgs(x=_ptr(acts[i - 1]), w=_ptr(wf), bias=_ptr(bp), out=_ptr(acts[i]), N=N, H=h, W=w, cin=cin,
                                                               cout=cout, stride=stride, neg_slope=1.0 if train else SLOPE, fmt=fmt), stream)
                    h, w = (h - 1) // stride + 1, (w - 1) // stride + 1
                if train:
                    bn = bns[i]
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
                    L.call('rumpy_enc_bn_train', L.EncBnArgs(x=_ptr(acts[i]), gamma=_ptr(bn.weight), beta=_ptr(bn.bias),
                                                             running_mean=_ptr(bn.running_mean), running_var=_ptr(bn.running_var),
                                                             num_batches_tracked=_ptr(bn.num_batches_tracked), partial=_ptr(plan['partial']),
                                                             scale_shift=_ptr(plan['scale_shift']), P=N * h * w, C=cout, eps=bn.eps, momentum=mom,
                                                             neg_slope=SLOPE, fmt=fmt), stream)
            L.check(L.lib().rumpy_enc_pool(_ptr(acts[5]), _ptr(fea), N, h * w, 256, fmt, stream), 'rumpy_enc_pool')

        # the launch list depends on the mode and on the addresses it reads: parameters, statistics and the slot's image buffers
        gkey = (train, _ptr(w0), _ptr(imgs[1][0]), _ptr(bns[0].weight), _ptr(bns[0].running_mean))
        if graphable:
            if plan['gkey'].get(train) != gkey:
                plan['graphs'][train], plan['gkey'][train] = _LaunchGraph(), gkey
            plan['graphs'][train].run(launches, dev)
        else:
            launches()
        if train:
            self._stats_epoch += 1
        return fea.clone()

    # ------------------------------------------------------------------ training: forward that keeps what the backward pass needs
    def _trunk_params(self):
        convs, bns = self._convs()
        c = self.__dict__.get('_trunk_cache')
        if c is None or c[0] is not convs[0].weight:      # Parameter objects are replaced by .to() / flatten() only
            c = self.__dict__['_trunk_cache'] = (convs[0].weight, [t for cv, b in zip(convs, bns) for t in (cv.weight, cv.bias, b.weight, b.bias)])
        return c[1]

    def _training_images(self, dev):
        key = self._key(False)
        if self._train_images is None or self._train_images[0] != key:
            convs, _ = self._convs()
            ws = [c.weight.detach() for c in convs]
            bs = [c.bias.detach() for c in convs]
            imgs, keep = self._pack(ws[1:], bs[1:], dev, dgrad=True, slot='train', fwd_fmt=TRAIN_FMT)
            self._train_images = (key, imgs, keep)
        return self._train_images[1]

    def _train_plan(self, N, H, W, dev):
        key = (N, H, W, dev.index)
        p = self._train_plans.get(key)
        if p is not None:
            return p
        lib = L.lib()
        new = lambda *shape, dtype=BF16: torch.empty(*shape, dtype=dtype, device=dev)
        z, a, abf, dz, da, dims, h, w = [], [], [], [], [], [], H, W
        for cin, cout, stride in LAYERS:
            hi, wi = h, w                                   # this conv's input size = the grid its gradients live on
            h, w = (h - 1) // stride + 1, (w - 1) // stride + 1
            dims.append((hi, wi, h, w))
            z.append(new(N, h, w, cout, dtype=torch.float32 if TRAIN_Z32 else TRAIN_DT))       # conv outputs: fp32 (round 4) / fp16 (TRAIN_FMT)
            a.append(new(N, h, w, cout, dtype=TRAIN_DT))
            abf.append(new(N, h, w, cout) if (TRAIN_FMT and len(abf) < 5) else None)     # bf16 copy of a stage output: the next conv's weight-gradient operand
            # gradient at the conv output on the stride-1 grid of the conv's input (stride 2: every second pixel, the rest stays zero)
            dz.append(torch.zeros(N, hi, wi, cout, dtype=BF16, device=dev) if stride == 2 else new(N, h, w, cout))
            da.append(new(N, h, w, cout))                   # gradient at the stage's (LeakyReLU) output; the last one comes from the pool
        part = new(max(int(lib.rumpy_enc_bn_partial_floats(t.shape[0] * t.shape[1] * t.shape[2], t.shape[3])) for t in z), dtype=torch.float32)
        p = dict(z=z, a=a, abf=abf, dz=dz, da=da, dims=dims, partial=part, coef=new(3 * 256, dtype=torch.float32),
                 ss=[new(2 * c, dtype=torch.float32) for _, c, _ in LAYERS], saved=[new(2 * c, dtype=torch.float32) for _, c, _ in LAYERS],
                 zero_bias=torch.zeros(256, dtype=torch.float32, device=dev), x=new(N, 3, H, W, dtype=torch.float32),
                 fea=new(N, 256, dtype=torch.float32), dfea=new(N, 256, dtype=torch.float32),
                 head_slab=new(max(1, int(lib.rumpy_head_wgrad_slab_floats(3, 64))), dtype=torch.float32))
        # ---- weight-gradient jobs of convs 1..5: unit = (layer, cin chunk, cout tile); a unit's pixel tiles are cut into ranges of `per` tiles,
        # one job + one slab each; one reduce item per unit (rumpy_wgrad_grouped / rumpy_wgrad_reduce of the SR path)
        ntiles = lambda hh, ww: N * ((hh + L.TILE_H - 1) // L.TILE_H) * ((ww + L.TILE_W - 1) // L.TILE_W)
        units = [(i, ch, ct) for i in range(1, 6) for ch in range(LAYERS[i][0] // 64) for ct in range(LAYERS[i][1] // 64)]
        total_tiles = sum(ntiles(*dims[i][:2]) for i, _, _ in units)
        per = max(1, -(-total_tiles // (2 * max(1, int(lib.rumpy_device_cus())))))
        sf = int(lib.rumpy_wgrad_slab_floats(4))
        nslab = sum(-(-ntiles(*dims[i][:2]) // per) for i, _, _ in units)
        slabs = new(nslab * sf, dtype=torch.float32)
        convs, _ = self._convs()
        gidx = {id(q): j for j, q in enumerate(self.param_list)}
        jobs, items, off = [], [], 0
        for i, ch, ct in units:
            cin, cout, _ = LAYERS[i]
            hi, wi = dims[i][:2]
            nt = ntiles(hi, wi)
            ranges = [(t0, min(nt, t0 + per)) for t0 in range(0, nt, per)]
            for k, (t0, t1) in enumerate(ranges):
                jobs.append(L.WgradJob(x=_ptr(abf[i - 1] if TRAIN_FMT else a[i - 1]), dy=_ptr(dz[i]), slab=slabs.data_ptr() + 4 * (off + k * sf), n0=0, n1=N, t0=t0, t1=t1,
                                       H=hi, W=wi, x_cstride=cin, x_coff=ch * 64, dy_mode=0, dy_cstride=cout, dy_coff=ct * 64, mt=4))
            items.append(L.ReduceItem(slab=slabs.data_ptr() + 4 * off, slab_stride=sf, njobs=len(ranges), mt=4, co_count=64, co_mode=0,
                                      co_off=ct * 64, ci_total=cin, ci_off=ch * 64, write_bias=1 if ch == 0 else 0, scale=1.0,
                                      gw=_ptr(self.grad_views[gidx[id(convs[i].weight)]]), gb=_ptr(self.grad_views[gidx[id(convs[i].bias)]])))
            off += len(ranges) * sf
        p.update(slabs=slabs, jobs=_dev_bytes((L.WgradJob * len(jobs))(*jobs), dev), njobs=len(jobs),
                 items=_dev_bytes((L.ReduceItem * len(items))(*items), dev), nitems=len(items), flat_g_ptr=self.flat_g.data_ptr())
        if len(self._train_plans) > 4:
            self._train_plans.clear()
            self._evictions += 1
        self._train_plans[key] = p
        return p

    def _train_forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('rumpy_amd: the degradation encoder runs on the GPU only (no CPU fallback)')
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError('rumpy_amd: encoder input must be [N,3,H,W]')
        if self.flat_p is None or self.E[0].weight.data_ptr() != self.flat_p.data_ptr():
            self.flatten()
        dev = x.device
        N, _, H, W = x.shape
        plan = self._train_plan(N, H, W, dev)
        imgs = self._training_images(dev)
        convs, bns = self._convs()
        z, a = plan['z'], plan['a']
        xs, fea = plan['x'], plan['fea']
        xs.copy_(x.detach())

        def launches():
            stream = torch.cuda.current_stream(dev).cuda_stream
            L.call('rumpy_head_fwd', L.HeadFwdArgs(x=_ptr(xs), w=_ptr(convs[0].weight), b=_ptr(convs[0].bias), out=_ptr(z[0]), N=N, C=3, H=H, W=W,
                                                   cout=64, neg_slope_m1=0.0, fmt=L.FMT_F32 if TRAIN_Z32 else TRAIN_FMT), stream)
            for i, (cin, cout, stride) in enumerate(LAYERS):
                hi, wi, ho, wo = plan['dims'][i]
                if i > 0:
                    wf, _, bp = imgs[i - 1][:3]
                    if TRAIN_Z32:      # unrounded filter (image + residual image), fp32 conv output
                        L.call('rumpy_enc_conv', L.EncConvArgs(x=_ptr(a[i - 1]), w=_ptr(wf), bias=_ptr(bp), out=_ptr(z[i]), N=N, H=hi, W=wi, cin=cin,
                                                               cout=cout, stride=stride, neg_slope=1.0, fmt=TRAIN_FMT, w_lo=_ptr(imgs[i - 1][3]),
                                                               out_fmt=L.FMT_F32), stream)
                    elif stride == 1:
                        _conv_plain(_ptr(a[i - 1]), _ptr(wf), _ptr(bp), _ptr(z[i]), N, hi, wi, cin, cout, stream, fmt=TRAIN_FMT)
                    else:
                        L.call('rumpy_enc_conv', L.EncConvArgs(x=_ptr(a[i - 1]), w=_ptr(wf), bias=_ptr(bp), out=_ptr(z[i]), N=N, H=hi, W=wi, cin=cin,
                                                               cout=cout, stride=stride, neg_slope=1.0, fmt=TRAIN_FMT), stream)
                bn = bns[i]
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
                args = L.EncBnArgs(x=_ptr(z[i]), gamma=_ptr(bn.weight), beta=_ptr(bn.bias), running_mean=_ptr(bn.running_mean),
                                   running_var=_ptr(bn.running_var), num_batches_tracked=_ptr(bn.num_batches_tracked),
                                   partial=_ptr(plan['partial']), scale_shift=_ptr(plan['ss'][i]), P=N * ho * wo, C=cout, eps=bn.eps, momentum=mom,
                                   neg_slope=SLOPE, fmt=TRAIN_FMT, x_fmt=L.FMT_F32 if TRAIN_Z32 else 0)
                L.check(L.lib().rumpy_enc_bn_train_keep(C.byref(args), _ptr(a[i]), _ptr(plan['abf'][i]), _ptr(plan['saved'][i]), stream), 'rumpy_enc_bn_train_keep')
            ho, wo = plan['dims'][5][2:]
            L.check(L.lib().rumpy_enc_pool(_ptr(a[5]), _ptr(fea), N, ho * wo, 256, TRAIN_FMT, stream), 'rumpy_enc_pool')

        gkey = (_ptr(convs[0].weight), _ptr(imgs[0][0]), _ptr(bns[0].weight), _ptr(bns[0].running_mean))
        if plan.get('gkey') != gkey:
            plan['gkey'], plan['fwd_graph'], plan['bwd_graph'] = gkey, _LaunchGraph(), _LaunchGraph()
        if all(bn.momentum is not None for bn in bns):
            plan['fwd_graph'].run(launches, dev)
        else:
            launches()
        self._stats_epoch += 1
        self._token += 1
        plan['token'] = self._token
        return fea.clone(), (N, H, W, self._token), xs

    def _train_backward(self, x, token, dfea):
        N, H, W, tok = token
        dev = x.device
        plan = self._train_plans.get((N, H, W, dev.index))
        if plan is None or plan.get('token') != tok:
            raise RuntimeError('rumpy_amd: backward pass of an encoder forward pass whose activations were overwritten by a later one '
                               '(one training forward per backward on this path)')
        if plan['flat_g_ptr'] != self.flat_g.data_ptr():
            raise RuntimeError('rumpy_amd: the encoder was re-flattened between forward and backward')
        lib = L.lib()
        imgs = self._training_images(dev)
        convs, bns = self._convs()
        gidx = {id(q): j for j, q in enumerate(self.param_list)}
        gv = lambda t: self.grad_views[gidx[id(t)]]
        dpool = plan['dfea']
        dpool.copy_(dfea.detach())
        z, a, dz, da = plan['z'], plan['a'], plan['dz'], plan['da']

        def launches():
            stream = torch.cuda.current_stream(dev).cuda_stream
            for i in range(5, -1, -1):
                cin, cout, stride = LAYERS[i]
                hi, wi, ho, wo = plan['dims'][i]
                bn = bns[i]
                L.call('rumpy_enc_bn_bwd', L.EncBnBwdArgs(z=_ptr(z[i]), da=None if i == 5 else _ptr(da[i]), dpool=_ptr(dpool) if i == 5 else None,
                                                          scale_shift=_ptr(plan['ss'][i]), saved=_ptr(plan['saved'][i]), gamma=_ptr(bn.weight),
                                                          dgamma=_ptr(gv(bn.weight)), dbeta=_ptr(gv(bn.bias)), dz=_ptr(dz[i]),
                                                          partial=_ptr(plan['partial']), coef=_ptr(plan['coef']), N=N, Ho=ho, Wo=wo, C=cout,
                                                          up=stride, Hz=hi if stride == 2 else ho, Wz=wi if stride == 2 else wo,
                                                          neg_slope=SLOPE, scale=1.0, fmt=L.FMT_F32 if TRAIN_Z32 else TRAIN_FMT), stream)
                if i > 0:      # data gradient: the stride-1 convolution of dz (on the input's grid) with the transposed, flipped filter
                    wd = imgs[i - 1][1]
                    _conv_plain(_ptr(dz[i]), _ptr(wd), _ptr(plan['zero_bias']), _ptr(da[i - 1]),
                                N, hi, wi, cout, cin, stream)
            L.check(lib.rumpy_wgrad_grouped(_ptr(plan['jobs']), plan['njobs'], 4, 0, stream), 'rumpy_wgrad_grouped')
            L.check(lib.rumpy_wgrad_reduce(_ptr(plan['items']), plan['nitems'], stream), 'rumpy_wgrad_reduce')
            L.call('rumpy_head_wgrad', L.HeadWgradArgs(x=_ptr(x), dy=_ptr(dz[0]), slab=_ptr(plan['head_slab']), gw=_ptr(gv(convs[0].weight)),
                                                       gb=_ptr(gv(convs[0].bias)), N=N, C=3, H=H, W=W, cout=64, scale=1.0), stream)

        plan['bwd_graph'].run(launches, dev)
        plan['token'] = None
        for t in self._trunk_params():
            if t.requires_grad:
                t.grad = gv(t)

    def forward(self, x):
        """(fea, {'q': mlp(fea)}) like the reference.  Under ``.train()`` with gradients enabled and a trainable trunk the trunk is one
        autograd node (HIP both ways); otherwise it is the inference path (no gradient reaches the trunk: ``encoder_freeze_mode`` 'all' /
        'pre_q', or evaluation)."""
        trunk = self._trunk_params()
        if self.training and torch.is_grad_enabled() and any(t.requires_grad for t in trunk):
            fea = _TrunkFn.apply(x, self, *trunk)
        else:
            fea = self.features(x)
        from .head import mlp_head         # the mlp head forward and backward in HIP (csrc/contrastive.hip)
        return fea, {'q': mlp_head(self.mlp, fea)}

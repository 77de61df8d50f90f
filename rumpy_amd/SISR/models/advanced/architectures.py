"""EDSR / RCAN for the MI355X HIP path - drop-in for rumpy/SISR/models/advanced/architectures.py (EDSR :198-241,
RCAN :140-176, RCAB :60-84, ResidualGroup :107-124, CALayer :24-44) and common.py (ResBlock :51-75, Upsampler :23-48).

The nn.Module tree exists to own the parameters under the reference's exact ``state_dict`` keys and OIHW fp32
layout (checkpoints interchange) and to consume the RNG like the reference's constructors do (same seed -> same
initial weights).  It does NOT compute: ``forward`` hands the whole network to the hand-written HIP kernels
through ``rumpy_amd.engine.SREngine``.  There is no CPU / eager fallback - calling the net without a GPU or
without the built extension raises.

All parameters live in ONE flat fp32 HBM buffer (``flat_p``; gradients in ``flat_g``): each ``nn.Parameter`` is a
view, so the fused Adam kernel and the gradient all-reduce see one contiguous tensor.
"""
import math
import os

import torch
from torch import nn

from rumpy_amd.engine import CALayerParams, ConvLayer, NetSpec, QLayerNParams, QLayerParams, SREngine


def _ceil64(f):
    return (f + 63) // 64 * 64


def _embed(mod, pin=None, pout=None):
    """Feature widths the kernels do not have (not a multiple of 64): the layer is built at the reference's shape - same RNG use, same
    initial values - and then EMBEDDED in the top-left corner of a zero filter of the next kernel width (pin / pout: padded input /
    output channels; the reference accepts any width, architectures.py:145,207).  Padded channels carry exact zeros through every layer
    (zero filter rows and biases, ReLU(0) = 0, residual adds of zeros, PixelShuffle rows c * r^2 + s of c >= f) and receive exactly zero
    gradients (their inputs or their upstream gradients are zero), so Adam leaves them at zero: the network computes the reference's
    function at the cost of the padded width.  ``state_dict`` / ``load_state_dict`` / the optimizer's checkpoint crop and pad
    (``_rumpy_real_shape``), so files interchange with the reference at its shapes."""
    co, ci = mod.weight.shape[:2]
    pin, pout = pin or ci, pout or co
    if (pin, pout) == (ci, co):
        return mod
    with torch.no_grad():
        w = torch.zeros(pout, pin, *mod.weight.shape[2:], dtype=mod.weight.dtype)
        w[:co, :ci] = mod.weight
        b = torch.zeros(pout, dtype=mod.bias.dtype)
        b[:co] = mod.bias
    real_w, real_b = tuple(mod.weight.shape), tuple(mod.bias.shape)
    mod.weight, mod.bias = nn.Parameter(w), nn.Parameter(b)
    mod.weight._rumpy_real_shape, mod.bias._rumpy_real_shape = real_w, real_b
    mod.in_channels, mod.out_channels = pin, pout
    return mod


def _conv(cin, cout, k=3, pin=None, pout=None):
    # default_conv, common.py:6-9 (parameter container; same default init / RNG use as nn.Conv2d there)
    return _embed(nn.Conv2d(cin, cout, k, padding=k // 2, bias=True), pin, pout)


class _Holder(nn.Module):
    """A module whose children only hold parameters (``body`` Sequential naming as in the reference)."""

    def __init__(self, mods):
        super().__init__()
        self.body = nn.Sequential(*mods)


class _ResBlockParams(_Holder):
    # common.py:51-75: body = [conv, ReLU, conv] -> keys body.0.*, body.2.*
    def __init__(self, feats, res_scale, pad=None):
        super().__init__([_conv(feats, feats, pin=pad, pout=pad), nn.ReLU(True), _conv(feats, feats, pin=pad, pout=pad)])
        self.res_scale = res_scale


class _CAParams(nn.Module):
    # architectures.py:24-44: conv_du = [conv1x1, ReLU, conv1x1, Sigmoid] -> keys conv_du.0.*, conv_du.2.*
    def __init__(self, feats, reduction, pad=None):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(_embed(nn.Conv2d(feats, feats // reduction, 1, padding=0, bias=True), pin=pad), nn.ReLU(inplace=True),
                                     _embed(nn.Conv2d(feats // reduction, feats, 1, padding=0, bias=True), pout=pad), nn.Sigmoid())


class _RCABParams(_Holder):
    # architectures.py:60-84: body = [conv, ReLU, conv, CALayer]; res_scale is stored and IGNORED (:79-84)
    def __init__(self, feats, reduction, res_scale, pad=None):
        super().__init__([_conv(feats, feats, pin=pad, pout=pad), nn.ReLU(True), _conv(feats, feats, pin=pad, pout=pad),
                          _CAParams(feats, reduction, pad)])
        self.res_scale = res_scale


class _GroupParams(_Holder):
    # architectures.py:107-124: body = n x RCAB + conv
    def __init__(self, feats, reduction, res_scale, n_resblocks, pad=None):
        super().__init__([_RCABParams(feats, reduction, res_scale, pad) for _ in range(n_resblocks)] + [_conv(feats, feats, pin=pad, pout=pad)])


def _upsampler(scale, feats, pad=None):
    # common.py:23-48, act=False, bn=False.  (Padded widths: output channel c * r^2 + s of the conv becomes channel c of the shuffled image, so
    # the reference's rows are the first r^2 * feats of the padded filter - the same corner embedding as everywhere.)
    mods = []
    if (scale & (scale - 1)) == 0:
        for _ in range(int(math.log(scale, 2))):
            mods += [_conv(feats, 4 * feats, pin=pad, pout=4 * pad if pad else None), nn.PixelShuffle(2)]
    elif scale == 3:
        mods += [_conv(feats, 9 * feats, pin=pad, pout=9 * pad if pad else None), nn.PixelShuffle(3)]
    else:
        raise NotImplementedError
    return nn.Sequential(*mods)


class _NetFn(torch.autograd.Function):
    """Whole-network autograd node for the generic (non-fused-loss) path: forward/backward both run the HIP engine;
    parameter gradients are deposited straight into the flat gradient buffer (``p.grad`` views, overwritten)."""

    @staticmethod
    def forward(ctx, x, net, train, meta, *params):
        net.check_step_status()          # the previous step's watchdog word, if nobody has looked at it yet
        out, _, plan = net.engine_forward(x, train=train, meta=meta)
        ctx.net, ctx.plan, ctx.gen = net, plan, plan.gen
        # a metadata tensor with a gradient path of its own (the embedding of a jointly trained degradation encoder) gets its gradient too
        ctx.meta_shape = tuple(meta.shape) if (meta is not None and ctx.needs_input_grad[3]) else None
        return out

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        if ctx.plan.gen != ctx.gen:
            # the plan's saved activations belong to a LATER forward pass of the same shape (two outputs alive at once, a deferred
            # backward): the reference's autograd would handle it, this static plan cannot - fail instead of returning wrong gradients
            raise RuntimeError('rumpy_amd: backward() of an output whose saved activations were overwritten by a later training forward '
                               'pass of the same shape; run forward and backward of one batch before the next forward pass')
        net.engine.backward(ctx.plan, 1.0, gout=gout.contiguous().float(), on_ready=getattr(net, 'grad_ready_hook', None))
        net.stage_step_status()          # read back behind the backward pass; examined behind the step's loss read-back (BaseModel.run_train) or by the next pass
        net.attach_grads()
        dmeta = net.engine.meta_grad(ctx.plan).reshape(ctx.meta_shape) if ctx.meta_shape is not None else None
        return (None, None, None, dmeta) + tuple(None for _ in net.param_list)


class HipSRNet(nn.Module):
    """Common machinery of the HIP-backed SR nets."""

    def _finalize(self):
        self.param_list = list(self.parameters())
        self._note_real_shapes()
        self.flat_p = self.flat_g = None
        # RUMPY_GRAPH=1 replays the fused training pass as a captured hipGraph.  Off by default: measured on MI355X the
        # step is GPU-bound (dependent kernel boundaries), eager launches keep up and a replay gains nothing (1.99 vs 2.03 ms)
        self.use_graph = os.environ.get('RUMPY_GRAPH') == '1'
        self.engine = None
        self._loss_host, self._loss_event, self.early_loss = None, None, False
        self._packed_version = None
        self._flatten()

    def set_precision(self, precision):
        """None / 'bf16': the default training arithmetic.  'fp8': the one-launch residual-block / RCAB kernels of training plans run their
        3x3 sweeps on the block-scaled fp8 MFMA (BASELINE config 5; its own accuracy class).  Evaluation plans are unaffected."""
        if precision not in (None, 'bf16', 'fp8'):
            raise RuntimeError("rumpy_amd: precision is None, 'bf16' or 'fp8' (got %r)" % (precision,))
        self.precision = None if precision == 'bf16' else precision
        if self.engine is not None and self.precision == 'fp8':
            self.engine.enable_fp8()
        elif self.engine is not None and getattr(self.engine, 'fp8', False):
            self.engine = None          # back to bf16: a fresh engine (plans are rebuilt)
            self._packed_version = None

    # ---- flat parameter storage ----
    def _flatten(self):
        dev = self.param_list[0].device
        total = sum(p.numel() for p in self.param_list)
        flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self.offsets = []
        with torch.no_grad():
            for p in self.param_list:
                n = p.numel()
                flat_p[off:off + n].copy_(p.data.reshape(-1).float())
                p.data = flat_p[off:off + n].view(p.shape)
                self.offsets.append(off)
                off += n
        self.flat_p, self.flat_g = flat_p, flat_g
        self.grad_views = [flat_g[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, self.param_list)]
        self.attach_grads()
        self.engine = None
        self._packed_version = None

    def train(self, mode=True):
        """The handlers call net.train() / net.eval() at every step (base_architecture.py:472,503) and nn.Module.train walks every sub-module -
        0.1-0.3 ms of host time per step here, for modules that only OWN parameters (nothing in them has a mode): walked once per change."""
        if self.training == mode and self.__dict__.get('_mode_walked') == mode:
            return self
        super().train(mode)
        self.__dict__['_mode_walked'] = mode
        return self

    def attach_grads(self):
        for p, g in zip(self.param_list, self.grad_views):
            if p.requires_grad:
                p.grad = g

    def _apply(self, fn, recurse=True):
        had = getattr(self, 'param_list', None) is not None
        super()._apply(fn)
        if had:
            self.param_list = list(self.parameters())   # Parameter objects may have been replaced
            if self.flat_p is None or self.param_list[0].data_ptr() != self.flat_p.data_ptr():
                self._flatten()
        return self

    # ---- widths embedded in the next kernel width (_embed): checkpoints keep the reference's shapes ----
    def _note_real_shapes(self):
        named = list(self.named_parameters())
        self._real_by_name = {k: tuple(p._rumpy_real_shape) for k, p in named if getattr(p, '_rumpy_real_shape', None) is not None}
        self.real_shapes = [self._real_by_name.get(k) for k, _ in named]       # aligned with param_list (None: not padded)
        if self._real_by_name:
            self._register_load_state_dict_pre_hook(self._pad_loaded_state)

    def real_numel(self):
        """parameter count of the reference's network (what its logs print), without the zero padding"""
        return sum(math.prod(shp or p.shape) for shp, p in zip(self.real_shapes, self.param_list))

    def state_dict(self, *args, **kw):
        sd = super().state_dict(*args, **kw)
        prefix = kw.get('prefix', args[1] if len(args) > 1 else '')
        for k, shp in self._real_by_name.items():
            if prefix + k in sd:
                sd[prefix + k] = sd[prefix + k][tuple(slice(0, n) for n in shp)].detach().clone()
        return sd

    def _pad_loaded_state(self, state_dict, prefix, *unused):
        """load_state_dict pre-hook (also reached through a parent module's load): entries at the reference's shapes -> zero-padded"""
        own = dict(self.named_parameters())
        for k, shp in self._real_by_name.items():
            v = state_dict.get(prefix + k)
            if v is not None and tuple(v.shape) == shp:
                full = torch.zeros(own[k].shape, dtype=v.dtype, device=v.device)
                full[tuple(slice(0, n) for n in shp)] = v
                state_dict[prefix + k] = full

    def load_state_dict(self, state_dict, strict=True, **kw):
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        self._packed_version = None
        return res

    # ---- engine ----
    def _spec(self):
        raise NotImplementedError

    def _conv_layer(self, name, mod, kind='main', shuffle=False):
        cv = ConvLayer(name, mod.weight.data, mod.bias.data, kind, shuffle)
        idx = {id(p): i for i, p in enumerate(self.param_list)}
        cv.gw, cv.gb = self.grad_views[idx[id(mod.weight)]], self.grad_views[idx[id(mod.bias)]]
        return cv

    def _ca_layer(self, name, ca):
        c0, c2 = ca.conv_du[0], ca.conv_du[2]
        lp = CALayerParams(name, c0.weight.data, c0.bias.data, c2.weight.data, c2.bias.data)
        idx = {id(p): i for i, p in enumerate(self.param_list)}
        lp.gw1, lp.gb1 = self.grad_views[idx[id(c0.weight)]], self.grad_views[idx[id(c0.bias)]]
        lp.gw2, lp.gb2 = self.grad_views[idx[id(c2.weight)]], self.grad_views[idx[id(c2.bias)]]
        return lp

    def _q_layer(self, name, qn):
        convs = [m for m in qn.attribute_integrator if isinstance(m, nn.Conv2d)]
        idx = {id(p): i for i, p in enumerate(self.param_list)}
        if len(convs) != 2 or convs[-1].weight.shape[0] > 64:       # ParaCALayer's num_layers other than the default, or more than 64 features: the general-depth launches (rumpy_q_mlpn_*)
            return QLayerNParams(name, [dict(w=c.weight.data.reshape(c.weight.shape[0], -1), b=c.bias.data, gw=self.grad_views[idx[id(c.weight)]],
                                             gb=self.grad_views[idx[id(c.bias)]]) for c in convs])
        c0, c2 = convs
        lp = QLayerParams(name, c0.weight.data.reshape(c0.weight.shape[0], -1), c0.bias.data,
                          c2.weight.data.reshape(c2.weight.shape[0], -1), c2.bias.data)
        lp.gw1, lp.gb1 = self.grad_views[idx[id(c0.weight)]], self.grad_views[idx[id(c0.bias)]]
        lp.gw2, lp.gb2 = self.grad_views[idx[id(c2.weight)]], self.grad_views[idx[id(c2.bias)]]
        return lp

    def _ensure_engine(self):
        if not self.flat_p.is_cuda:
            raise RuntimeError('rumpy_amd: this network only runs on an MI355X through the HIP extension; '
                               'there is no CPU path (parameters are on %s)' % self.flat_p.device)
        if self.engine is None:
            self.engine = SREngine(self._spec(), self.flat_p.device)
            if getattr(self, 'precision', None) == 'fp8':
                self.engine.enable_fp8()
            if self.engine.use_finish and all(p.requires_grad for p in self.param_list):
                self.engine.build_update_table(self.flat_p, self.param_list, self.offsets)
        v = self._weights_version()
        if self._packed_version != v:
            self.engine.repack()
            self._packed_version = v

    def _weights_version(self):
        # torch-side in-place writes (load_state_dict, a stock optimizer, manual edits) bump these counters
        return sum(p._version for p in self.param_list)

    def mark_weights_updated(self):
        """Called after flat_p was rewritten behind torch's back: re-pack the bf16 filter images."""
        self._ensure_engine()
        self.engine.repack()
        self._packed_version = self._weights_version()

    def mark_weights_clean(self):
        """The fused optimizer has already re-packed (inside its own launch list / graph)."""
        self._packed_version = self._weights_version()

    def engine_forward(self, x, train, target=None, meta=None):
        self._ensure_engine()
        if not x.is_cuda:
            raise RuntimeError('rumpy_amd: input must be on the GPU')
        self.engine.eval_defer = bool(getattr(self, 'eval_defer', False))      # set by BaseModel.run_eval(keep_on_device=True)
        return self.engine.forward(x.float().contiguous(), train=train, target=target, meta=self._meta_matrix(meta, x))

    @staticmethod
    def _meta_matrix(meta, x):
        """[N,M,1,1] / [N,M] metadata as the Q models receive it (attention_manipulators/__init__.py:84-103) -> fp32 [N,M] on x's device"""
        if meta is None:
            return None
        return meta.reshape(meta.shape[0], -1).to(device=x.device, dtype=torch.float32).contiguous()

    def forward(self, x, metadata=None):
        train = self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.param_list)
        if train:
            return _NetFn.apply(x, self, True, metadata, *self.param_list)
        out, _, _ = self.engine_forward(x, train=False, meta=metadata)
        return out

    def fused_l1_forward_backward(self, x, y, metadata=None, out_to_host=False):
        """forward + nn.L1Loss + full backward in one pass (base_architecture.py:474-480 minus the optimizer).
        Returns (loss device scalar, out).  Gradients land in flat_g / p.grad.
        out_to_host ('inline' | 'side'; run_train(keep_on_device=False), the reference's default call): the output image is final behind the
        forward pass - its copy into pinned host memory is queued there (_stage_out); take_staged_out() hands it over."""
        if self.use_graph:
            self._ensure_engine()
            out, loss, plan = self.engine.train_pass_graphed(x.float().contiguous(), y.float().contiguous(), meta=self._meta_matrix(metadata, x))
            self._stage_loss(loss, plan.rcab_status)       # the replay's loss and strip-exchange watchdog word, fenced like the eager path's
            return loss, out
        if getattr(self, 'grad_ready_hook', None) is not None:
            self._ensure_engine()
            self.engine.set_two_phase()       # data-parallel form: the weight-gradient shares are cut per gradient-buffer half (plans rebuilt once)
        out, loss, plan = self.engine_forward(x, train=True, target=y.float().contiguous(), meta=metadata)
        # The loss is final once the forward pass has run: its read-back is queued HERE (pinned buffer + event), ahead of the
        # backward launches, so that run_train's `loss.cpu().numpy()` (the reference API returns the loss of every step,
        # base_architecture.py:482-485) waits for the forward pass only and the host can queue the next step while the GPU
        # still runs this step's backward pass and optimizer.
        self._stage_loss(loss, plan.rcab_status)
        if out_to_host:
            self._stage_out(out, side=(out_to_host == 'side'))
        self.engine.backward(plan, 1.0 / out.numel(), on_ready=getattr(self, 'grad_ready_hook', None))
        return loss, out

    def _stage_out(self, out, side=False):
        """queue `out` -> a pinned host tensor behind what the current stream has queued so far (the forward pass); the caller waits for that copy's
        event only, i.e. run_train returns while the backward pass and the optimizer still run.  Default: the copy goes to the step's OWN stream
        (0.3 ms of PCIe in front of the backward pass; no second queue in the process).  side=True: on a copy stream, under the backward pass."""
        host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)       # (caching host allocator: the block of an earlier step comes back)
        ev = torch.cuda.Event()
        if side:
            if getattr(self, '_copy_stream', None) is None:
                self._copy_stream = torch.cuda.Stream(out.device)
            self._copy_stream.wait_stream(torch.cuda.current_stream(out.device))
            with torch.cuda.stream(self._copy_stream):
                host.copy_(out, non_blocking=True)
                ev.record(self._copy_stream)
            out.record_stream(self._copy_stream)
        else:
            host.copy_(out, non_blocking=True)
            ev.record()
        self._staged_out = (host, ev)

    def take_staged_out(self):
        """-> the host copy of the last step's output queued by fused_l1_forward_backward(out_to_host=True) (waits for that copy only), or None"""
        st, self._staged_out = getattr(self, '_staged_out', None), None
        if st is None:
            return None
        st[1].synchronize()
        return st[0]

    def _stage_loss(self, loss, status=None):
        """Queue the read-back of a step's loss (and the strip-exchange watchdog word) into pinned memory, fenced by an event."""
        if self._loss_host is None:
            self._loss_host = torch.empty(1, dtype=torch.float32).pin_memory()
            self._loss_event = torch.cuda.Event()
        self._loss_host.copy_(loss.reshape(1), non_blocking=True)
        if status is not None:          # strip-exchange watchdog of the one-launch RCAB kernels, read back with the loss
            if getattr(self, '_status_host', None) is None:
                self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._status_host.copy_(status, non_blocking=True)
        self._loss_event.record()
        self.early_loss = True

    def take_early_loss(self):
        """-> 0-d float32 ndarray of the last fused step's loss (waits for its forward pass only), or None."""
        if not getattr(self, 'early_loss', False):
            return None
        self.early_loss = False
        self._loss_event.synchronize()
        st = getattr(self, '_status_host', None)
        if st is not None and int(st[0]) != 0:
            self._watchdog_fired(int(st[0]))
        return self._loss_host.numpy().copy().reshape(())

    # ---- watchdog of the launches that wait for other workgroups (persistent block chain, in-launch pool exchange) ----
    def _watchdog_fired(self, code):
        """A step's status word came back non-zero: a workgroup gave up waiting for another one, its results are garbage.  The optimizer launch of
        such a step reads the same word on the device and changes nothing (rumpy_adam_pack_args.skip_if), so the weights are intact.  A single
        process then switches its engine to launches that wait for nobody and goes on (the affected steps were skipped, says the warning);
        a data-parallel rank or a captured graph cannot (the replicas would diverge / the graph is fixed): it raises, as before round 6."""
        eng = self.engine
        text = eng.watchdog_text(code)
        st = getattr(self, '_status_host', None)
        if st is not None:
            st.zero_()
        plan_like = next((p for p in list(eng.plans.values())[::-1] if getattr(p, 'rcab_status', None) is not None), None)
        may = (os.environ.get('RUMPY_WATCHDOG_STRICT') != '1' and not getattr(self, 'use_graph', False)
               and not getattr(self, 'data_parallel_rank', False) and getattr(self, 'grad_ready_hook', None) is None)
        what = eng.degrade(plan_like) if (may and plan_like is not None) else None
        if what is None:
            if plan_like is not None:
                plan_like.flags.zero_()
            raise RuntimeError('rumpy_amd: %s.  The optimizer steps of the affected passes were skipped on the device (weights intact); their '
                               'losses and outputs are invalid (GPU shared with another job?)' % text)
        import warnings
        warnings.warn('rumpy_amd: %s (GPU shared with another job?).  The optimizer steps of the affected passes were skipped on the device (weights '
                      'intact, their losses / outputs are invalid); training continues with: %s' % (text, what), RuntimeWarning)

    def stage_step_status(self):
        """generic-loss path: queue the read-back of the step's watchdog word behind what has been launched so far (the fused-L1 path read
        it with the loss, this path never did)"""
        st = getattr(self.engine, 'step_status', None) if self.engine is not None else None
        if st is None:
            return
        if getattr(self, '_status_host', None) is None:
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        if getattr(self, '_status_event', None) is None:
            self._status_event = torch.cuda.Event()
        self._status_host.copy_(st, non_blocking=True)
        self._status_event.record()
        self._status_pending = True

    def check_step_status(self):
        """examine the word staged by stage_step_status (waits for the launches in front of it); called by the handlers behind their loss read-back
        and by the next training pass"""
        if not getattr(self, '_status_pending', False):
            return
        self._status_pending = False
        self._status_event.synchronize()
        if int(self._status_host[0]) != 0:
            self._watchdog_fired(int(self._status_host[0]))

    def l1_eval(self, x, y, metadata=None):
        out, loss, _ = self.engine_forward(x, train=False, target=y.float().contiguous(), meta=metadata)
        return out, loss


class EDSR(HipSRNet):
    """architectures.py:198-241.  Keys: head.0, body.{i}.body.{0,2}, body.{num_blocks}, tail.0.{0,2..}, tail.1"""

    def __init__(self, in_features=3, out_features=3, net_features=64, num_blocks=16, scale=4, res_scale=0.1):
        super().__init__()
        f = net_features
        P = _ceil64(f) if f % 64 else None       # a width between the kernel widths runs embedded in the next one (_embed)
        self.scale, self.res_scale = scale, res_scale
        self.head = nn.Sequential(_conv(in_features, f, pout=P))
        self.body = nn.Sequential(*([_ResBlockParams(f, res_scale, P) for _ in range(num_blocks)] + [_conv(f, f, pin=P, pout=P)]))
        self.tail = nn.Sequential(_upsampler(scale, f, P), _conv(f, out_features, pin=P))
        # the fused forward + L1 + backward pass is built from 64-feature kernels: wider nets train through the generic autograd node
        self.supports_fused_l1 = (P or f) == 64
        self._finalize()

    def _spec(self):
        if self.scale not in (1, 2, 3, 4, 8):
            raise RuntimeError('rumpy_amd: scale %s not supported by the HIP path' % self.scale)
        body = []
        blocks = list(self.body)[:-1]
        for i, b in enumerate(blocks):
            body.append(('resblock', self._conv_layer('body.%d.body.0' % i, b.body[0]),
                         self._conv_layer('body.%d.body.2' % i, b.body[2]), float(b.res_scale)))
        ups = [self._conv_layer('tail.0.%d' % i, m, shuffle=True) for i, m in enumerate(self.tail[0]) if isinstance(m, nn.Conv2d)]
        return NetSpec(self._conv_layer('head.0', self.head[0], kind='head'), body,
                       self._conv_layer('body.%d' % len(blocks), self.body[-1]), ups,
                       self._conv_layer('tail.1', self.tail[1], kind='tail'), self.scale)


class RCAN(HipSRNet):
    """architectures.py:140-176.  Keys: head.0, body.{g}.body.{b}.body.{0,2}, body.{g}.body.{b}.body.3.conv_du.{0,2},
    body.{g}.body.{n_resblocks}, body.{n_resgroups}, tail.0.{0,2}, tail.1"""

    def __init__(self, n_resblocks=20, n_resgroups=10, n_feats=64, in_feats=3, out_feats=3, scale=4, reduction=16,
                 res_scale=1.0, **kwargs):
        super().__init__()
        f = n_feats
        P = _ceil64(f) if f % 64 else None       # a width between the kernel widths runs embedded in the next one (_embed)
        self.supports_fused_l1 = (P or f) == 64   # (wide nets: the handlers take the generic loss path, as for EDSR at 128 .. 256 features)
        self.scale = scale
        self.head = nn.Sequential(_conv(in_feats, f, pout=P))
        self.body = nn.Sequential(*([_GroupParams(f, reduction, res_scale, n_resblocks, P) for _ in range(n_resgroups)]
                                    + [_conv(f, f, pin=P, pout=P)]))
        self.tail = nn.Sequential(_upsampler(scale, f, P), _conv(f, out_feats, pin=P))
        self._finalize()

    def _spec(self):
        if self.scale not in (1, 2, 3, 4, 8):
            raise RuntimeError('rumpy_amd: scale %s not supported by the HIP path' % self.scale)
        body = []
        groups = list(self.body)[:-1]
        for gi, grp in enumerate(groups):
            mods = list(grp.body)
            items = []
            for bi, rb in enumerate(mods[:-1]):
                pre = 'body.%d.body.%d.body' % (gi, bi)
                items.append(('rcab', self._conv_layer(pre + '.0', rb.body[0]), self._conv_layer(pre + '.2', rb.body[2]),
                              self._ca_layer(pre + '.3', rb.body[3])))
            body.append(('group', items, self._conv_layer('body.%d.body.%d' % (gi, len(mods) - 1), mods[-1])))
        ups = [self._conv_layer('tail.0.%d' % i, m, shuffle=True) for i, m in enumerate(self.tail[0]) if isinstance(m, nn.Conv2d)]
        return NetSpec(self._conv_layer('head.0', self.head[0], kind='head'), body,
                       self._conv_layer('body.%d' % len(groups), self.body[-1]), ups,
                       self._conv_layer('tail.1', self.tail[1], kind='tail'), self.scale)

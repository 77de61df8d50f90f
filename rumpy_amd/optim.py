"""Fused Adam over the flat parameter buffer of a HipSRNet.

Semantics: torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0, amsgrad=False) as the reference builds it
(rumpy/shared_framework/models/base_architecture.py:79-99); one HIP kernel updates every parameter, then the
bf16 filter images are re-packed.  It subclasses torch.optim.Adam only so that ``param_groups`` /
``state_dict()`` / ``load_state_dict()`` keep the reference checkpoint layout (``optimizer`` entry of
``train_model_<epoch>``, base_architecture.py:247-249) and torch LR schedulers can drive it.
"""
import math

import torch

from . import _lib as L
from .staging import PinnedRing


class FlatAdam(torch.optim.Adam):
    def __init__(self, net, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.net = net
        params = [p for p in net.param_list if p.requires_grad]
        if len(params) != len(net.param_list):
            raise RuntimeError('FlatAdam: frozen parameters are not supported by the fused update')
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        dev = net.flat_p.device
        self.flat_m = torch.zeros_like(net.flat_p)
        self.flat_v = torch.zeros_like(net.flat_p)
        self._step_t = torch.tensor(0.0)
        self._hyper_dev = torch.zeros(8, dtype=torch.float32, device=dev)
        # the host runs ahead of the GPU: every step's hyper-parameters get their own fenced pinned slot (staging.PinnedRing)
        self._hyper_ring = PinnedRing((8,), torch.float32) if dev.type == 'cuda' else None
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._sumsq_partial = torch.zeros(1024, dtype=torch.float32, device=dev)
        self._graphs = {}
        self._bind_state()

    def _bind_state(self):
        net = self.net
        for p, off in zip(net.param_list, net.offsets):
            n = p.numel()
            self.state[p] = {'step': self._step_t,
                             'exp_avg': self.flat_m[off:off + n].view(p.shape),
                             'exp_avg_sq': self.flat_v[off:off + n].view(p.shape)}

    @property
    def step_count(self):
        return int(self._step_t.item())

    def zero_grad(self, set_to_none=True):
        # gradients are overwritten (not accumulated) by every backward pass of the engine
        self.net.attach_grads()

    def state_dict(self):
        """torch.optim.Adam's layout; moments of a parameter that lives embedded in a wider zero filter (architectures._embed) are written at
        the reference's shape, like the parameter itself"""
        sd = super().state_dict()
        real = getattr(self.net, 'real_shapes', None)
        if real and any(real):
            for idx, st in sd['state'].items():
                shp = real[idx]
                if shp is not None:
                    st = dict(st)
                    for k in ('exp_avg', 'exp_avg_sq'):
                        if k in st:
                            st[k] = st[k][tuple(slice(0, n) for n in shp)].clone()
                    sd['state'][idx] = st
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # torch deep-copied the loaded state: pull it back into the flat buffers
        net = self.net
        step = 0.0

        def padded(t, p):
            if tuple(t.shape) == tuple(p.shape):
                return t
            full = torch.zeros(p.shape, dtype=t.dtype, device=t.device)      # moments at the reference's shape -> the embedded layout
            full[tuple(slice(0, n) for n in t.shape)] = t
            return full
        with torch.no_grad():
            for p, off in zip(net.param_list, net.offsets):
                st = self.state.get(p, None)
                n = p.numel()
                if st and 'exp_avg' in st:
                    self.flat_m[off:off + n].copy_(padded(st['exp_avg'], p).reshape(-1))
                    self.flat_v[off:off + n].copy_(padded(st['exp_avg_sq'], p).reshape(-1))
                    step = float(st['step'])
                else:
                    self.flat_m[off:off + n].zero_()
                    self.flat_v[off:off + n].zero_()
        self._step_t = torch.tensor(step)
        self._bind_state()

    @torch.no_grad()
    def step(self, closure=None, grad_mult=1.0, max_norm=None):
        net = self.net
        if not net.flat_p.is_cuda:
            raise RuntimeError('rumpy_amd FlatAdam: parameters are not on the GPU; there is no CPU path')
        group = self.param_groups[0]
        beta1, beta2 = group['betas']
        self._step_t += 1
        t = float(self._step_t)
        hv = L.AdamHyper(lr=group['lr'], beta1=beta1, beta2=beta2, eps=group['eps'], bias_c1=1.0 - beta1 ** t,
                         sqrt_bias_c2=math.sqrt(1.0 - beta2 ** t), grad_mult=grad_mult, max_norm=float(max_norm) if max_norm else 0.0)
        graphed = bool(getattr(net, 'use_graph', False))
        if graphed:
            # a captured graph replays with new values: they live in device memory and travel through a fenced pinned slot per step
            slot, h = self._hyper_ring.acquire()
            h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7] = (hv.lr, hv.beta1, hv.beta2, hv.eps, hv.bias_c1, hv.sqrt_bias_c2,
                                                              hv.grad_mult, hv.max_norm)
            self._hyper_dev.copy_(h, non_blocking=True)
            self._hyper_ring.sent(slot)
        dev = net.flat_p.device
        n = net.flat_p.numel()

        # the watchdog word of the step's persistent launches (engine.step_status): non-zero -> this launch changes nothing (rumpy_adam_args.skip_if)
        st = getattr(getattr(net, 'engine', None), 'step_status', None)
        skip = st.data_ptr() if (st is not None and not graphed) else None

        def launches(stream):
            sumsq = None
            if max_norm:
                L.call('rumpy_sumsq', L.SumsqArgs(g=net.flat_g.data_ptr(), n=n, partial=self._sumsq_partial.data_ptr(),
                                                  out=self._sumsq.data_ptr()), stream)
                sumsq = self._sumsq.data_ptr()
            upd = getattr(net.engine, 'update_items', None)
            if upd is not None:
                # Adam AND the re-packing of the bf16 filter images in one launch (csrc/finish.hip): every parameter is an item of the table
                L.call('rumpy_adam_pack', L.AdamPackArgs(items=upd[0].data_ptr(), nitems=upd[1], p=net.flat_p.data_ptr(), g=net.flat_g.data_ptr(),
                                                         m=self.flat_m.data_ptr(), v=self.flat_v.data_ptr(),
                                                         hyper=self._hyper_dev.data_ptr() if graphed else None, sumsq=sumsq, hyper_value=hv,
                                                         skip_if=skip), stream)
                return
            L.call('rumpy_adam_step', L.AdamArgs(p=net.flat_p.data_ptr(), g=net.flat_g.data_ptr(), m=self.flat_m.data_ptr(),
                                                 v=self.flat_v.data_ptr(), n=n, hyper=self._hyper_dev.data_ptr() if graphed else None,
                                                 sumsq=sumsq, hyper_value=hv, skip_if=skip), stream)      # eager: by value with the launch
            net.engine.repack(stream)

        net._ensure_engine()
        if hasattr(net.engine, 'pack_gen'):
            net.engine.pack_gen += 1          # new bf16 images: the fp16 images of the evaluation plans follow lazily (also under graph replays)
        if not getattr(net, 'use_graph', False):
            launches(torch.cuda.current_stream(dev).cuda_stream)
            net.mark_weights_clean()
            return None
        # hyper-parameters live in device memory, so ONE captured graph (per clipping mode) serves every step
        key = bool(max_norm)
        if key not in self._graphs:
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                launches(torch.cuda.current_stream(dev).cuda_stream)
            self._graphs[key] = g
            # capture does not execute: fall through to the replay below
        self._graphs[key].replay()
        net.mark_weights_clean()
        return None

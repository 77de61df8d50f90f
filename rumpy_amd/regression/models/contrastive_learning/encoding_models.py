"""Degradation encoder of the blind-SR pipeline on the MI355X path - mirror of
rumpy/regression/models/contrastive_learning/encoding_models.py:5-55 (``Encoder``, the DASR encoder: six 3x3 convs
3-64-64-128/2-128-256/2-256 with BatchNorm + LeakyReLU(0.1), global average pool, ``mlp`` 256-256-256).

Inference only: the blind-SR handlers keep it frozen (``encoder_freeze_mode='all'``); training the encoder itself (MoCo / SupMoCo,
rumpy/regression) is outside the hot path.  The torch modules below only OWN the parameters and buffers under the reference's keys
(``E.{0,1,3,4,...}.*``, ``mlp.{0,2}.*``) and creation order; ``forward`` runs hand-written HIP kernels (rumpy_head_fwd, rumpy_enc_conv,
rumpy_enc_bn_train, rumpy_enc_pool) and fails loudly without a GPU.  Both BatchNorm modes are implemented because the reference uses
both: running statistics under ``.eval()`` (folded into the filters, one launch per layer) and batch statistics + running-statistics
update under ``.train()`` - which is the mode the reference's run_train leaves the frozen encoder in (base_architecture.py:472).

Only the pooled feature vector ``fea`` (embedding_type 'pre-q', the handlers' default) is produced; the ``mlp`` head ('q') is kept as
parameters for checkpoint interchange and refused at run time."""
import numpy as np
import torch
from torch import nn

from rumpy_amd import _lib as L

BF16 = torch.bfloat16
LAYERS = ((3, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1))
SLOPE = 0.1


def _ptr(t):
    return None if t is None else t.data_ptr()


class Encoder(nn.Module):
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

    # ------------------------------------------------------------------ filter images
    def _convs(self):
        return [self.E[3 * i] for i in range(6)], [self.E[3 * i + 1] for i in range(6)]

    def _key(self, with_stats):
        convs, bns = self._convs()
        ts = [t for c in convs for t in (c.weight, c.bias)]
        if with_stats:
            ts += [t for b in bns for t in (b.weight, b.bias, b.running_mean, b.running_var)]
        return tuple(t._version for t in ts) + tuple(t.data_ptr() for t in ts) + ((self._stats_epoch,) if with_stats else ())

    def _pack(self, weights, biases, dev):
        """[(w fp32 OIHW, b)] of the five 64-multiple convs -> MFMA fragment images through rumpy_pack_weights"""
        items, out = [], []
        for w, b in zip(weights, biases):
            cout, cin = w.shape[:2]
            wf = torch.empty(cout * cin * 9, dtype=BF16, device=dev)
            bp = torch.empty(cout, dtype=torch.float32, device=dev)
            items.append(L.PackItem(w=_ptr(w), b=_ptr(b), w_fwd=_ptr(wf), w_dgrad=None, b_packed=_ptr(bp), cout=cout, cin=cin, kind=0, shuffle=0))
            out.append((wf, bp))
        tab = torch.from_numpy(np.frombuffer(bytes((L.PackItem * len(items))(*items)), dtype=np.uint8).copy()).to(dev)
        L.check(L.lib().rumpy_pack_weights(_ptr(tab), len(items), torch.cuda.current_stream(dev).cuda_stream), 'rumpy_pack_weights')
        return out, (tab, weights, biases)

    def _raw_images(self, dev):
        key = self._key(False)
        if self._packed is None or self._packed[0] != key:
            convs, _ = self._convs()
            ws = [c.weight.detach().float().contiguous() for c in convs]
            bs = [c.bias.detach().float().contiguous() for c in convs]
            imgs, keep = self._pack(ws[1:], bs[1:], dev)
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
            imgs, keep = self._pack(ws[1:], bs[1:], dev)
            self._folded = (key, [(ws[0], bs[0])] + imgs, keep)
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
            p = dict(acts=acts, partial=part, scale_shift=torch.empty(2 * 256, dtype=torch.float32, device=dev))
            if len(self._plans) > 8:
                self._plans.clear()
            self._plans[k] = p
        return p

    def features(self, x):
        """x [N,3,H,W] fp32 on the GPU -> fea [N,256] fp32; BatchNorm mode follows ``self.training`` like the torch modules would."""
        if not x.is_cuda:
            raise RuntimeError('rumpy_amd: the degradation encoder runs on the GPU only (no CPU fallback)')
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError('rumpy_amd: encoder input must be [N,3,H,W]')
        x = x.float().contiguous()
        dev = x.device
        N, _, H, W = x.shape
        train = self.training
        if any(p.requires_grad for p in self.parameters()):
            raise RuntimeError('rumpy_amd: the encoder is inference-only on the HIP path (encoder_freeze_mode="all")')
        plan = self._plan(N, H, W, dev)
        imgs = self._raw_images(dev) if train else self._folded_images(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _, bns = self._convs()
        acts = plan['acts']
        w0, b0 = imgs[0]
        L.call('rumpy_head_fwd', L.HeadFwdArgs(x=_ptr(x), w=_ptr(w0), b=_ptr(b0), out=_ptr(acts[0]), N=N, C=3, H=H, W=W, cout=64,
                                               neg_slope_m1=0.0 if train else SLOPE - 1.0), stream)
        h, w = H, W
        for i, (cin, cout, stride) in enumerate(LAYERS):
            if i > 0:
                wf, bp = imgs[i]
                L.call('rumpy_enc_conv', L.EncConvArgs(x=_ptr(acts[i - 1]), w=_ptr(wf), bias=_ptr(bp), out=_ptr(acts[i]), N=N, H=h, W=w, cin=cin,
                                                       cout=cout, stride=stride, neg_slope=1.0 if train else SLOPE), stream)
                h, w = (h - 1) // stride + 1, (w - 1) // stride + 1
            if train:
                bn = bns[i]
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
                L.call('rumpy_enc_bn_train', L.EncBnArgs(x=_ptr(acts[i]), gamma=_ptr(bn.weight), beta=_ptr(bn.bias), running_mean=_ptr(bn.running_mean),
                                                         running_var=_ptr(bn.running_var), num_batches_tracked=_ptr(bn.num_batches_tracked),
                                                         partial=_ptr(plan['partial']), scale_shift=_ptr(plan['scale_shift']), P=N * h * w, C=cout,
                                                         eps=bn.eps, momentum=mom, neg_slope=SLOPE), stream)
        if train:
            self._stats_epoch += 1
        fea = torch.empty(N, 256, dtype=torch.float32, device=dev)
        L.check(L.lib().rumpy_enc_pool(_ptr(acts[5]), _ptr(fea), N, h * w, 256, stream), 'rumpy_enc_pool')
        return fea

    def forward(self, x):
        """(fea, out_dict) like the reference; the 'q' entry (mlp head) is not computed on this path."""
        return self.features(x), {}

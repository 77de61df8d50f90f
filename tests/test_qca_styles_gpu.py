"""-m gpu parity for the QCALayer styles whose gate MLP also reads the attribute vector ('max_concat', 'mini_concat', 'extended_attention',
'softmax'; rumpy/SISR/models/attention_manipulators/architectures.py:41-136): the gate kernels (csrc/qca_style.hip) against plain torch, and
the HIP QRCAN in each style through the QModel handler API against the CPU oracle (pinned on the real reference handler by golden G19)."""
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sr_oracle as O
from rumpy_amd import _lib as L
from rumpy_amd.shared_framework.models import define_model
from tests.test_network_gpu import _grad_check, self_psnr

SCHED = {'scheduler': 'cosine_annealing_warm_restarts', 'scheduler_params': {'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7}}
STYLES = ('max_concat', 'mini_concat', 'extended_attention', 'softmax')


def _meta(seed, n, m):
    return torch.from_numpy(np.random.default_rng(seed).uniform(0, 1, (n, m)).astype(np.float32))


@pytest.mark.parametrize('style', STYLES)
@pytest.mark.parametrize('N,M', [(5, 5), (32, 256), (1, 1)])
def test_gate_kernels_against_torch(style, N, M):
    """forward gate, d(mean), and every parameter gradient of the style's MLP, from the oracle's module on random pool sums"""
    dev, C, HW, T, K = torch.device('cuda:0'), 64, 48 * 48, 16, 18
    ca = O.StyledChannelAttention(C, 16, style, M)
    g = torch.Generator().manual_seed(11)
    for p in ca.parameters():
        p.data = torch.randn(p.shape, generator=g) * (0.5 / np.sqrt(p.shape[1] if p.dim() > 1 else 1.0))
    from rumpy_amd.SISR.models.attention_manipulators.architectures import _QCAParams
    hp = _QCAParams(C, 16, style, M)
    hp.load_state_dict(ca.state_dict())
    pool = (torch.randn(N, T, C, generator=g) * 0.3 + 1.0) * (HW / T)   # per-tile partial sums of a feature map with channel means around 1
    part = torch.randn(N, K, C, generator=g)                      # partial sums of dy * t2
    attr = _meta(12, N, M)
    keep, layers, n_prev = [], [], C
    for conv, cat, relu_in, act in hp.gate_layers():
        w = conv.weight.data.reshape(conv.weight.shape[0], -1).contiguous().to(dev)
        b = conv.bias.data.contiguous().to(dev)
        gw, gb = torch.zeros_like(w), torch.zeros_like(b)
        keep.append((w, b, gw, gb))
        layers.append(L.QcaLayer(w=w.data_ptr(), b=b.data_ptr(), gw=gw.data_ptr(), gb=gb.data_ptr(), n_prev=n_prev, n_out=w.shape[0], cat=cat,
                                 relu_in=relu_in, act=act))
        n_prev = w.shape[0]
    d = {k: t.to(dev).contiguous() for k, t in dict(pool=pool, part=part, attr=attr).items()}
    acts, delta = torch.zeros(N, L.QCA_ACT_STRIDE, device=dev), torch.zeros(N, L.QCA_ACT_STRIDE, device=dev)
    gate, dpool = torch.zeros(N, C, device=dev), torch.zeros(N, C, device=dev)
    a = L.QcaArgs(nlayers=len(layers), N=N, C=C, M=M, ntiles=T, nchunks=K, inv_hw=1.0 / HW, scale=0.25, pool=d['pool'].data_ptr(),
                  attr=d['attr'].data_ptr(), acts=acts.data_ptr(), gate=gate.data_ptr(), partial=d['part'].data_ptr(), dpool=dpool.data_ptr(),
                  delta=delta.data_ptr())
    for i, ly in enumerate(layers):
        a.layers[i] = ly
    s = torch.cuda.current_stream(dev).cuda_stream
    L.call('rumpy_qca_gate_fwd', a, s)
    L.call('rumpy_qca_gate_bwd', a, s)
    import ctypes
    tab = torch.from_numpy(np.frombuffer(bytes((L.QcaArgs * 1)(a)), dtype=np.uint8).copy()).to(dev)
    L.check(L.lib().rumpy_qca_bwd_params(tab.data_ptr(), 1, s), 'params')
    torch.cuda.synchronize()
    # torch, in float64 (the gate is read off as (x * gate) / x on a 1 x 1 map whose value is the channel mean: exact only without rounding noise)
    ca = ca.double()
    mean = (pool.double().sum(1) / HW).clone().requires_grad_(True)
    x = mean.reshape(N, C, 1, 1)
    gt = ca(x, attr.double().reshape(N, M, 1, 1)) / x
    assert torch.allclose(gate.cpu().double(), gt.detach().reshape(N, C), atol=2e-6, rtol=1e-5), style
    (gt.reshape(N, C) * part.double().sum(1)).sum().backward()

    def close(got, ref):
        return float((got.cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-9
    assert close(dpool * HW, mean.grad), style
    for (w, b, gw, gb), (conv, _, _, _) in zip(keep, ca_layers(ca)):
        assert close(gw, 0.25 * conv.weight.grad.reshape(gw.shape)), style
        assert close(gb, 0.25 * conv.bias.grad), style


def ca_layers(ca):
    if ca.style in ('max_concat', 'softmax'):
        return [(ca.conv_du[0], 1, 0, 1), (ca.conv_du[2], 0, 0, 2)]
    if ca.style == 'mini_concat':
        return [(ca.pre_concat, 0, 0, 0), (ca.conv_du[1], 1, 1, 2)]
    return [(m[0], 1, 0, 1) for m in ca.feature_convs] + [(ca.final_conv[0], 0, 0, 2)]


def test_gate_launch_refuses_what_it_does_not_compute():
    a = L.QcaArgs(nlayers=5, N=1, C=64, M=1)
    assert L.lib().rumpy_qca_gate_fwd(a, None) != 0
    a = L.QcaArgs(nlayers=1, N=1, C=64, M=300)
    assert L.lib().rumpy_qca_gate_fwd(a, None) != 0


@pytest.mark.parametrize('style', STYLES)
def test_styled_qrcan_train_steps_and_eval_against_oracle(style):
    names = ['blur_sigma', 'noise_level', 'jpeg_q', 'extra_a', 'extra_b']
    kw = dict(scale=2, n_feats=64, n_resgroups=2, n_resblocks=2, reduction=16)
    h = define_model('qrcan', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False,
                     metadata_list=None, metadata=list(names), style=style, include_q_layer=False, lr=1e-3, **SCHED, **kw)
    onet = O.build_oracle('qrcan', style=style, include_q_layer=False, num_metadata=h.num_metadata, **kw)
    assert list(onet.state_dict().keys()) == list(h.net.state_dict().keys())
    sd = O.seeded_state_dict(onet, 826)
    onet.load_state_dict(sd)
    h.net.load_state_dict(sd)
    oh = O.OracleHandler(onet, lr=1e-3, scheduler=SCHED['scheduler'], scheduler_params=SCHED['scheduler_params'])
    keys = [(n, 'numeric') for n in names]
    for step in range(3):
        x, y = O.synthetic_batch(830 + step, 3, lr_hw=16, scale=2)
        m = _meta(840 + step, 3, len(names))
        loss, out = h.run_train(x=x, y=y, metadata=m, metadata_keys=keys)
        oloss, oout = oh.run_train(x, y, extra_channels=m.unsqueeze(2).unsqueeze(3))
        assert abs(float(loss) - float(oloss)) < (2e-3 if step == 0 else 1e-2) * float(oloss), (style, step)
        if step == 0:
            assert self_psnr(out, oout) >= 50.0
            # 'softmax': the gate is ~ 1/64 per channel, the block's contribution and with it the squeeze-excite gradients are small against the
            # bf16 noise of the pooled means (3.2e-2 on one bias at cosine 1.000000): wider bound for that style
            worst = _grad_check(h, oh, tol=6e-2 if style == 'softmax' else 3e-2)
            print(style, 'worst grad rel err', worst)
            for k, p in h.net.named_parameters():
                if 'final_body' in k and 'body.' in k:
                    assert float(p.grad.abs().max()) > 0, k
    xe, ye = O.synthetic_batch(850, 2, lr_hw=21, scale=2)
    me = _meta(851, 2, len(names))
    ev, evl, _ = h.run_eval(x=xe, y=ye, request_loss=True, metadata=me, metadata_keys=keys)
    oev, oevl, _ = oh.run_eval(xe, ye, request_loss=True, extra_channels=me.unsqueeze(2).unsqueeze(3))
    assert self_psnr(ev, oev) >= 45.0 and abs(float(evl) - float(oevl)) < 1e-2 * float(oevl)
    # a wide image (W > 48): the conv pair runs as two strip launches with pool sums, the gate path is the same
    xw, yw = O.synthetic_batch(852, 1, lr_hw=(13, 70), scale=2)
    mw = _meta(853, 1, len(names))
    ew, ewl, _ = h.run_eval(x=xw, y=yw, request_loss=True, metadata=mw, metadata_keys=keys)
    oew, oewl, _ = oh.run_eval(xw, yw, request_loss=True, extra_channels=mw.unsqueeze(2).unsqueeze(3))
    assert self_psnr(ew, oew) >= 45.0 and abs(float(ewl) - float(oewl)) < 1e-2 * float(oewl)

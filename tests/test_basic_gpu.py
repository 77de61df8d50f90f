"""-m gpu parity of the reference's "basic" models (SRCNN = BASELINE config 0, VDSR) on the direct fp32 convolution kernels:
kernels against torch (ATen CPU ops = the reference's arithmetic library), the handlers against the oracle and against fixture G15
(three training steps + evaluation of the REAL reference handlers).  fp32 arithmetic on both sides, different summation order:
tolerances 1e-5 relative on activations / losses, 1e-4 on gradients (written at each assert)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import sr_oracle as O
from rumpy_amd import _lib as L
from rumpy_amd.shared_framework.models import define_model
from rumpy_amd.SISR.models.interface import SISRInterface

SCHED = dict(scheduler='cosine_annealing_warm_restarts', scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7})
CASES = {'srcnn': dict(kw={}, clip=None),
         'vdsr': dict(kw=dict(kernel_pattern=[3, 3, 3, 3], channel_pattern=[1, 8, 8, 8, 1]), clip=0.1)}


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def y_batch(seed, n, h, w):
    g = np.random.default_rng(seed)
    return (torch.from_numpy(g.uniform(0, 1, (n, 1, h, w)).astype(np.float32)),
            torch.from_numpy(g.uniform(0, 1, (n, 1, h, w)).astype(np.float32)))


@pytest.mark.parametrize('N,Cin,Cout,H,W,k', [(2, 1, 64, 20, 27, 9), (2, 64, 32, 20, 27, 5), (2, 32, 1, 20, 27, 5), (1, 3, 5, 7, 5, 3),
                                              (3, 8, 17, 33, 16, 1), (1, 2, 4, 16, 48, 11), (1, 64, 64, 40, 40, 3), (1, 1, 1, 1, 1, 7)])
def test_direct_conv_forward_dgrad_wgrad_against_torch(N, Cin, Cout, H, W, k):
    g = torch.Generator().manual_seed(N * 1000 + Cin * 37 + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(N, Cout, H, W, generator=g)
    dy = torch.randn(N, Cout, H, W, generator=g)
    below = torch.randn(N, Cin, H, W, generator=g)                   # post-ReLU activation of the layer below (mask = below > 0)
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=k // 2)
    ref.backward(dy)
    xd, wd, bd, resd, dyd, belowd = (t.cuda() for t in (x, w, b, res, dy, below))
    for relu, use_res in ((0, False), (1, False), (0, True)):
        y = torch.empty(N, Cout, H, W, device='cuda')
        L.call('rumpy_dconv', L.DconvArgs(x=xd.data_ptr(), w=wd.data_ptr(), bias=bd.data_ptr(), mask=None, res=resd.data_ptr() if use_res else None,
                                           y=y.data_ptr(), N=N, Cin=Cin, Cout=Cout, H=H, W=W, k=k, relu=relu, transposed=0), _stream())
        want = ref.detach()
        want = torch.relu(want) if relu else want
        want = want + res if use_res else want
        assert _rel(y, want) < 1e-5, (relu, use_res)
    dx = torch.empty(N, Cin, H, W, device='cuda')
    L.call('rumpy_dconv', L.DconvArgs(x=dyd.data_ptr(), w=wd.data_ptr(), bias=None, mask=belowd.data_ptr(), res=None, y=dx.data_ptr(),
                                       N=N, Cin=Cout, Cout=Cin, H=H, W=W, k=k, relu=0, transposed=1), _stream())
    assert _rel(dx, xr.grad * (below > 0)) < 1e-5
    assert bool((dx.cpu()[below <= 0] == 0).all())
    n = int(L.lib().rumpy_dconv_wgrad_partial_floats(N, Cin, Cout, H, W, k))
    part = torch.empty(max(1, n), device='cuda')
    gw, gb = torch.empty(Cout, Cin, k, k, device='cuda'), torch.empty(Cout, device='cuda')
    for rep in range(2):                                              # twice: bitwise reproducible (fixed reduction order)
        L.call('rumpy_dconv_wgrad', L.DconvWgradArgs(x=xd.data_ptr(), dy=dyd.data_ptr(), partial=part.data_ptr(), gw=gw.data_ptr(), gb=gb.data_ptr(),
                                                     N=N, Cin=Cin, Cout=Cout, H=H, W=W, k=k, scale=0.5), _stream())
        if rep == 0:
            first = (gw.clone(), gb.clone())
    assert torch.equal(first[0], gw) and torch.equal(first[1], gb)
    assert _rel(gw, 0.5 * wr.grad) < 1e-4 and _rel(gb, 0.5 * br.grad) < 1e-4


def test_direct_conv_rejects_what_it_does_not_compute():
    x = torch.zeros(1, 1, 4, 4, device='cuda')
    for k in (2, 4, 13):
        a = L.DconvArgs(x=x.data_ptr(), w=x.data_ptr(), bias=None, mask=None, res=None, y=x.data_ptr(), N=1, Cin=1, Cout=1, H=4, W=4, k=k,
                        relu=0, transposed=0)
        with pytest.raises(RuntimeError):
            L.call('rumpy_dconv', a, _stream())
    with pytest.raises(NotImplementedError):
        define_model('srcnn', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=True, checkpoint_load=False, loss_masking=False,
                     padding='valid')


def test_mse_loss_and_gradient_against_torch():
    g = torch.Generator().manual_seed(5)
    for n in (1, 255, 70001):
        a, b = torch.rand(n, generator=g), torch.rand(n, generator=g)
        ar = a.clone().requires_grad_(True)
        want = F.mse_loss(ar, b)
        want.backward()
        ad, bd = a.cuda(), b.cuda()
        grad, part, loss = torch.empty(n, device='cuda'), torch.empty(1024, device='cuda'), torch.empty(1, device='cuda')
        L.call('rumpy_mse_loss', L.MseArgs(out=ad.data_ptr(), target=bd.data_ptr(), grad=grad.data_ptr(), partial=part.data_ptr(),
                                            loss=loss.data_ptr(), n=n), _stream())
        assert abs(float(loss) - float(want.detach())) < 1e-6 * max(1.0, float(want.detach()))
        assert _rel(grad, ar.grad) < 1e-6


def _handler(name, eval_mode=False, **kw):
    return define_model(name, model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=eval_mode, checkpoint_load=False, loss_masking=False,
                        metadata_list=None, **kw)


@pytest.mark.parametrize('name', ['srcnn', 'vdsr'])
def test_basic_handlers_follow_the_real_reference_handlers(golden_dir, name):
    """fixture G15: three run_train steps + run_eval of the REAL reference SRCNNHandler / VDSRHandler, and the oracle beside them"""
    g = np.load(os.path.join(golden_dir, 'g15_basic_small_train.npz'))
    case = CASES[name]
    torch.manual_seed(8)
    h = _handler(name, lr=1e-3, **SCHED, **case['kw'])
    assert list(h.net.state_dict().keys()) == [str(k) for k in g[name + '.keys']]
    init8 = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in h.net.state_dict().values()])
    assert np.allclose(init8, g[name + '.init8'], rtol=0, atol=1e-9)       # same seed -> the reference's initial weights
    assert [h.colorspace, h.im_input, h.model_name, type(h.criterion).__name__, str(h.grad_clip)] == [str(a) for a in g[name + '.attrs']]
    onet = O.build_oracle(name, **case['kw'])
    sd = O.seeded_state_dict(onet, 840)
    onet.load_state_dict(sd)
    h.net.load_state_dict(sd)
    oh = O.OracleHandler(onet, lr=1e-3, criterion='mse', grad_clip=case['clip'], **SCHED)
    for step in range(3):
        xb, yb = y_batch(850 + step, 2, 20, 27)
        loss, out = h.run_train(x=xb, y=yb)
        oloss, oout = oh.run_train(xb, yb)
        assert out.dtype == torch.float32 and not out.is_cuda and loss.shape == ()
        # step 0: fp32 summation order only.  Later steps: Adam normalises by sqrt(v), which amplifies last-bit differences of small gradients
        tol = 1e-5 if step == 0 else 2e-4
        assert abs(float(loss) - float(g['%s.loss%d' % (name, step)])) < tol * float(oloss)
        assert abs(float(loss) - float(oloss)) < tol * float(oloss)
        assert abs(h.get_learning_rate() - float(g['%s.lr_after%d' % (name, step)])) < 1e-12
        if step == 0:
            assert _rel(out, torch.from_numpy(g[name + '.out0'])) < 1e-5
            # the handler clips inside the fused Adam (the stored gradient stays unclipped); the fixture holds the reference's clipped one
            tot = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in h.net.parameters())))
            coef = 1.0 if case['clip'] is None else min(1.0, case['clip'] / (tot + 1e-6))
            for k, p in h.net.named_parameters():
                assert _rel(p.grad * coef, torch.from_numpy(g['%s.grad0.%s' % (name, k)])) < 1e-4, k
            for k, v in h.net.state_dict().items():
                assert np.allclose(v.cpu().numpy(), g['%s.w1.%s' % (name, k)], atol=2e-6), k
    worst_w3 = 0.0
    for k, v in h.net.state_dict().items():
        # a weight whose gradient is ~0 gets an Adam update of arbitrary sign on either side (g / sqrt(v) of rounding noise): bounded by the
        # distance three steps of lr 1e-3 can cover, and rare - the tensors agree in norm
        assert np.allclose(v.cpu().numpy(), g['%s.w3.%s' % (name, k)], atol=6e-3), k
        worst_w3 = max(worst_w3, _rel(v, torch.from_numpy(g['%s.w3.%s' % (name, k)])))
    print('worst relative weight difference after three steps: %.2e' % worst_w3)
    assert worst_w3 < 5e-3
    xe, ye = y_batch(890, 1, 33, 18)
    ev, evl, t = h.run_eval(x=xe, y=ye, request_loss=True, timing=True)
    assert _rel(ev, torch.from_numpy(g[name + '.eval_out'])) < 5e-4 and t > 0               # with the weights after three steps
    assert abs(float(evl) - float(g[name + '.eval_loss'])) < 5e-4 * float(g[name + '.eval_loss'])


@pytest.mark.parametrize('name', ['srcnn', 'vdsr'])
def test_ycbcr_branch_of_net_run_and_process(golden_dir, name):
    """SISRInterface.net_run_and_process for 'ycbcr' models (interface.py:113-121) against what the reference's functions produced"""
    g = np.load(os.path.join(golden_dir, 'g15_basic_small_train.npz'))
    itf = SISRInterface(tempfile.mkdtemp(), 'exp', gpu='single', sp_gpu=0, mode='eval',
                        new_params={'name': name, 'internal_params': dict(CASES[name]['kw'])})
    onet = O.build_oracle(name, **CASES[name]['kw'])
    sd = {k: torch.from_numpy(g['%s.w3.%s' % (name, k)]) for k in onet.state_dict().keys()}
    itf.model.net.load_state_dict(sd)
    assert itf.configuration == {'colorspace': 'ycbcr', 'input': 'interp'}
    rgb, ycbcr, loss, _ = itf.net_run_and_process(lr=torch.from_numpy(g[name + '.post_in']))
    assert loss is None and rgb.shape == g[name + '.post_rgb'].shape
    assert np.allclose(ycbcr, g[name + '.post_ycbcr'], atol=1e-5) and ycbcr.min() >= 0 and ycbcr.max() <= 1
    assert np.allclose(rgb, g[name + '.post_rgb'], atol=1e-5)


def test_other_criteria_run_through_the_autograd_node_and_checkpoints_interchange():
    """a criterion other than the stock nn.MSELoss -> whole-network autograd node over the same kernels; the checkpoint written by the
    handler loads into the oracle (reference key names, OIHW fp32) and back"""
    h = _handler('srcnn', lr=1e-3, kernel_pattern=[5, 3, 3], channel_pattern=[1, 12, 6, 1])
    onet = O.build_oracle('srcnn', kernel_pattern=[5, 3, 3], channel_pattern=[1, 12, 6, 1])
    sd = O.seeded_state_dict(onet, 841)
    onet.load_state_dict(sd)
    h.net.load_state_dict(sd)
    h.criterion = torch.nn.L1Loss()
    oh = O.OracleHandler(onet, lr=1e-3)            # L1
    xb, yb = y_batch(860, 2, 18, 18)
    loss, _ = h.run_train(x=xb, y=yb)
    oloss, _ = oh.run_train(xb, yb)
    assert abs(float(loss) - float(oloss)) < 1e-5 * float(oloss)
    for (k, p), (_, q) in zip(h.net.named_parameters(), onet.named_parameters()):
        assert _rel(p.grad, q.grad) < 1e-4, k
        assert np.allclose(p.detach().cpu().numpy(), q.detach().numpy(), atol=2e-6), k
    h.save_model('train_model')
    state = torch.load(os.path.join(h.model_save_dir, 'train_model_0'), weights_only=False)
    assert state['model_name'] == 'srcnn' and list(state['network'].keys()) == list(onet.state_dict().keys())
    onet2 = O.build_oracle('srcnn', kernel_pattern=[5, 3, 3], channel_pattern=[1, 12, 6, 1])
    onet2.load_state_dict(state['network'])
    h2 = _handler('srcnn', lr=1e-3, kernel_pattern=[5, 3, 3], channel_pattern=[1, 12, 6, 1])
    h2.load_model('train_model', 0, load_override=h.model_save_dir)
    xe, _ = y_batch(861, 1, 21, 30)
    o2, _, _ = h2.run_eval(x=xe)
    with torch.no_grad():
        assert _rel(o2, onet2(xe)) < 1e-5


def test_two_forward_passes_before_a_backward_pass_keep_their_own_activations():
    """the whole-network autograd node keeps the activations of ITS forward pass - a second forward of another batch before
    backward() must not change the first one's gradients (the reference's autograd behaves so)"""
    h = _handler('srcnn', lr=1e-3, kernel_pattern=[5, 3, 3], channel_pattern=[1, 12, 6, 1])
    h.net.train()
    xa, ya = y_batch(880, 2, 18, 18)
    xb, _ = y_batch(881, 2, 18, 18)
    xa, ya, xb = xa.cuda(), ya.cuda(), xb.cuda()
    out = h.net(xa)
    torch.nn.functional.l1_loss(out, ya).backward()
    ref = [p.grad.clone() for p in h.net.parameters()]
    out = h.net(xa)
    h.net(xb)                                  # another training forward in between
    torch.nn.functional.l1_loss(out, ya).backward()
    for p, r in zip(h.net.parameters(), ref):
        assert torch.equal(p.grad, r)


def test_default_vdsr_trains_on_a_full_size_patch_batch():
    """the 20-layer default VDSR (665,921 parameters, grad_clip 0.1) on a 16 x 1 x 64 x 64 batch: loss falls, everything finite"""
    torch.manual_seed(8)
    h = _handler('vdsr', lr=1e-3)
    assert sum(p.numel() for p in h.net.parameters()) == 665921
    xb, yb = y_batch(870, 16, 64, 64)
    yb = (xb + 0.05 * (yb - 0.5)).clamp(0, 1)           # a learnable residual task
    losses = [float(h.run_train(x=xb, y=yb)[0]) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_srcnn_on_the_set5_example_image_eval_psnr_within_0p02_db(golden_dir):
    """BASELINE config 0 (SRCNN x2 on the reference's example data) through SISRInterface.net_run_and_process: output, clipped YCbCr,
    RGB, evaluation loss as the REAL reference produced them (fixture G16), and the Y-PSNR within the north star's 0.02 dB."""
    g = np.load(os.path.join(golden_dir, 'g16_srcnn_set5_eval.npz'))
    itf = SISRInterface(tempfile.mkdtemp(), 'exp', gpu='single', sp_gpu=0, mode='eval', new_params={'name': 'srcnn', 'internal_params': {}})
    itf.model.net.load_state_dict(O.seeded_state_dict(O.build_oracle('srcnn'), 842))
    lr, hr = torch.from_numpy(g['lr_ycbcr']), torch.from_numpy(g['hr_ycbcr'])
    out_y, _, _ = itf.model.run_eval(lr[:, :1])
    assert _rel(out_y, torch.from_numpy(g['out_y'])) < 1e-5
    rgb, ycbcr, loss, _ = itf.net_run_and_process(lr=lr, hr=hr, request_loss=True)
    assert np.allclose(ycbcr, g['ycbcr'], atol=1e-5) and np.allclose(rgb, g['rgb'], atol=1e-5)
    assert abs(float(loss) - float(g['loss'])) < 1e-5 * float(g['loss'])
    p = O.y_psnr(ycbcr, np.clip(g['hr_ycbcr'], 0, 1))
    print('Y-PSNR hip %.4f vs reference %.4f dB' % (p, float(g['psnr'])))
    assert abs(p - float(g['psnr'])) <= 0.02

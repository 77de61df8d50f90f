"""Feature widths between the kernel widths (the reference accepts any `num_features` / `n_feats`, advanced/architectures.py:145,207): the
network runs EMBEDDED in the next kernel width - zero filters around the reference's (architectures._embed) - and keeps the reference's
shapes wherever a user or a file sees them.  Checked against the oracle at the REAL width."""
import os
import tempfile

import pytest
import torch

from oracle import sr_oracle as O
from rumpy_amd.shared_framework.models import define_model
from tests.test_network_gpu import _pair, self_psnr

pytestmark = pytest.mark.gpu

CASES = [
    ('edsr', dict(scale=2, num_features=48, num_blocks=2, res_scale=0.1)),       # -> the 64-feature one-launch block kernels
    ('edsr', dict(scale=4, num_features=96, num_blocks=2, res_scale=0.1)),       # -> 128 (two-chunk conv kernels, generic upsampler)
    ('edsr', dict(scale=3, num_features=200, num_blocks=1, res_scale=1.0)),      # -> 256, x3
    ('rcan', dict(scale=2, n_feats=32, n_resgroups=1, n_resblocks=2, reduction=8)),     # -> the 64-feature RCAB kernels, squeeze width 4
    ('rcan', dict(scale=2, n_feats=96, n_resgroups=1, n_resblocks=2, reduction=16)),    # -> 128 features (round 5: RCAN beyond 64), squeeze width 6
]


def _crop(t, shape):
    return t[tuple(slice(0, n) for n in shape)]


@pytest.mark.parametrize('name,kw', CASES)
def test_embedded_width_trains_and_evaluates_like_the_reference_width(name, kw):
    h, oh = _pair(name, 2101, **kw)
    scale = kw['scale']
    real = {k: tuple(v.shape) for k, v in oh.net.state_dict().items()}
    assert {k: tuple(v.shape) for k, v in h.net.state_dict().items()} == real            # files and users see the reference's shapes
    assert h.net.real_numel() == sum(p.numel() for p in oh.net.parameters())
    x, y = O.synthetic_batch(2200, 2, lr_hw=24, scale=scale)
    ev, evl, _ = h.run_eval(x=x, y=y, request_loss=True)
    oev, oevl, _ = oh.run_eval(x, y, request_loss=True)
    assert self_psnr(ev, oev) >= 45.0 and abs(float(evl) - float(oevl)) < 1e-2 * float(oevl)
    loss, out = h.run_train(x=x, y=y)
    oloss, oout = oh.run_train(x, y)
    assert self_psnr(out, oout) >= (55.0 if name == 'edsr' else 48.0) and abs(float(loss) - float(oloss)) < 3e-3 * float(oloss)
    worst = 0.0
    for (k, p), (k2, q) in zip(h.net.named_parameters(), oh.net.named_parameters()):
        assert k == k2
        g = p.grad.detach().float().cpu()
        r = q.grad
        inside = _crop(g, r.shape)
        # the padding receives EXACTLY zero gradient and stays exactly zero through the Adam step
        pad_mask = torch.ones_like(p.detach().cpu(), dtype=torch.bool)
        pad_mask[tuple(slice(0, n) for n in r.shape)] = False
        assert float(g[pad_mask].abs().sum()) == 0.0, k
        assert float(p.detach().cpu()[pad_mask].abs().sum()) == 0.0, k
        if float(r.norm()) > 0:
            rel = float((inside.double() - r.double()).norm() / r.double().norm())
            worst = max(worst, rel)
            assert rel < 3e-2, (k, rel)
        assert float((_crop(p.detach().cpu(), q.shape) - q.detach()).abs().max()) <= 2.001e-3, k      # first Adam step: |dw| <= lr
    print('worst gradient tensor', worst)
    for s in (2201, 2202):
        x, y = O.synthetic_batch(s, 2, lr_hw=24, scale=scale)
        loss, _ = h.run_train(x=x, y=y)
        oloss, _ = oh.run_train(x, y)
        assert abs(float(loss) - float(oloss)) < 6e-3 * float(oloss)


@pytest.mark.parametrize('name,kw', CASES[:1] + CASES[3:4])
def test_embedded_width_checkpoint_has_the_reference_shapes_and_resumes(name, kw):
    """save_model writes weights AND Adam moments at the reference's shapes; a second handler resumes from the file bit for bit"""
    sched = dict(scheduler='cosine_annealing_warm_restarts', scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7})
    d = tempfile.mkdtemp()

    def make():
        return define_model(name, model_save_dir=d, device=0, eval_mode=False, checkpoint_load=False, loss_masking=False, metadata_list=None,
                            lr=1e-3, **kw, **sched)
    h = make()
    onet = O.build_oracle(name, **kw)
    h.net.load_state_dict(O.seeded_state_dict(onet, 2301))
    x, y = O.synthetic_batch(2300, 2, lr_hw=16, scale=kw['scale'])
    for _ in range(2):
        h.run_train(x=x, y=y)
    h.set_epoch(2)
    h.save_model('train_model')
    ck = torch.load(os.path.join(d, 'train_model_2'), map_location='cpu', weights_only=False)
    real = {k: tuple(v.shape) for k, v in onet.state_dict().items()}
    assert {k: tuple(v.shape) for k, v in ck['network'].items()} == real
    shapes = [tuple(v.shape) for v in onet.state_dict().values()]
    for i, st in ck['optimizer']['state'].items():
        assert tuple(st['exp_avg'].shape) == shapes[i] and tuple(st['exp_avg_sq'].shape) == shapes[i], i
    # the reference's own torch.optim.Adam takes the file's optimizer entry as it is
    o_opt = torch.optim.Adam(onet.parameters(), lr=1e-3)
    o_opt.load_state_dict(ck['optimizer'])
    h2 = make()
    h2.load_model('train_model', 2)
    l1, o1 = h.run_train(x=x, y=y)
    l2, o2 = h2.run_train(x=x, y=y)
    assert float(l1) == float(l2) and torch.equal(o1, o2)
    for p, q in zip(h.net.parameters(), h2.net.parameters()):
        assert torch.equal(p.detach(), q.detach())

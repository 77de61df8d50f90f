"""-m gpu parity for the meta-attention row: the HIP QRCAN ('standard' style + q-layers) through the QModel
handler API against the CPU oracle (pinned on the real reference handler by golden G12), and the two q-layer kernels against
plain torch.  Tolerances as in test_network_gpu.py (bf16 operands / activations, fp32 master weights and gates)."""
import ctypes
import os
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


def _meta(seed, n, m):
    return torch.from_numpy(np.random.default_rng(seed).uniform(0, 1, (n, m)).astype(np.float32))


def _pair(names, wseed, eval_mode=False, lr=1e-3, **kw):
    h = define_model('qrcan', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=eval_mode, checkpoint_load=False, loss_masking=False,
                     metadata_list=None, metadata=list(names), style='standard', include_q_layer=True, lr=lr,
                     **({} if eval_mode else SCHED), **kw)
    onet = O.build_oracle('qrcan', style='standard', include_q_layer=True, num_metadata=h.num_metadata, **kw)
    assert list(onet.state_dict().keys()) == list(h.net.state_dict().keys())
    sd = O.seeded_state_dict(onet, wseed)
    onet.load_state_dict(sd)
    h.net.load_state_dict(sd)
    oh = O.OracleHandler(onet, lr=lr, eval_mode=eval_mode, scheduler=None if eval_mode else SCHED['scheduler'],
                         scheduler_params=SCHED['scheduler_params'])
    return h, oh


@pytest.mark.parametrize('N,M,Hq', [(7, 5, 32), (32, 256, 160), (64, 18, 41), (1, 1, 32)])
def test_q_mlp_kernels_against_torch(N, M, Hq):
    """(32, 256, 160) is the q-layer fed by the 256-entry contrastive embedding (256 -> 160 -> 64)"""
    dev = torch.device('cuda:0')
    lib = L.lib()
    C = 64
    g = torch.Generator().manual_seed(5)
    items, keep = [], []
    for _ in range(3):
        w1, b1 = torch.randn(Hq, M, generator=g) * 0.5, torch.randn(Hq, generator=g) * 0.2
        w2, b2 = torch.randn(C, Hq, generator=g) * 0.3, torch.randn(C, generator=g) * 0.2
        dz = torch.randn(N, C, generator=g)
        d = [t.to(dev).contiguous() for t in (w1, b1, w2, b2, dz)]
        outs = [torch.zeros(N, Hq, device=dev), torch.zeros(N, C, device=dev), torch.zeros(Hq, M, device=dev), torch.zeros(Hq, device=dev),
                torch.zeros(C, Hq, device=dev), torch.zeros(C, device=dev)]
        keep.append((w1, b1, w2, b2, dz, d, outs))
        items.append(L.QMlpItem(w1=d[0].data_ptr(), b1=d[1].data_ptr(), w2=d[2].data_ptr(), b2=d[3].data_ptr(), hidden=outs[0].data_ptr(),
                                gate=outs[1].data_ptr(), dzq=d[4].data_ptr(), gw1=outs[2].data_ptr(), gb1=outs[3].data_ptr(),
                                gw2=outs[4].data_ptr(), gb2=outs[5].data_ptr(), scale=0.5))
    arr = (L.QMlpItem * len(items))(*items)
    tab = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
    meta = _meta(3, N, M)
    md = meta.to(dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    L.check(lib.rumpy_q_mlp_fwd(tab.data_ptr(), len(items), md.data_ptr(), N, M, Hq, C, s), 'fwd')
    L.check(lib.rumpy_q_mlp_bwd_params(tab.data_ptr(), len(items), md.data_ptr(), N, M, Hq, C, s), 'bwd')
    dmeta = torch.full((N, M), float('nan'), device=dev)
    L.check(lib.rumpy_q_mlp_bwd_meta(tab.data_ptr(), len(items), N, M, Hq, C, dmeta.data_ptr(), s), 'bwd_meta')
    torch.cuda.synchronize()
    mref = meta.clone().requires_grad_(True)        # gradient at the metadata input: the sum over the layers
    sum(((torch.relu(mref @ w1.t() + b1) @ w2.t() + b2) * dz).sum() for w1, b1, w2, b2, dz, _, _ in keep).backward()
    assert torch.allclose(dmeta.cpu(), 0.5 * mref.grad, atol=2e-4, rtol=1e-4)
    for w1, b1, w2, b2, dz, d, outs in keep:
        p = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
        hid = torch.relu(meta @ p[0].t() + p[1])
        z = hid @ p[2].t() + p[3]
        assert torch.allclose(outs[0].cpu(), hid.detach(), atol=1e-4, rtol=1e-5)
        assert torch.allclose(outs[1].cpu(), torch.sigmoid(z).detach(), atol=1e-5)
        (z * dz).sum().backward()        # dz = gradient in front of the sigmoid
        for got, ref in zip(outs[2:], (p[0].grad, p[1].grad, p[2].grad, p[3].grad)):
            assert torch.allclose(got.cpu(), 0.5 * ref, atol=2e-4, rtol=1e-4)
    # shapes beyond the kernel's tables are refused, not truncated
    assert lib.rumpy_q_mlp_fwd(tab.data_ptr(), len(items), md.data_ptr(), 65, M, Hq, C, s) != 0
    assert lib.rumpy_q_mlp_fwd(tab.data_ptr(), len(items), md.data_ptr(), N, 257, Hq, C, s) != 0
    assert lib.rumpy_q_mlp_bwd_params(tab.data_ptr(), len(items), md.data_ptr(), N, M, 161, C, s) != 0


@pytest.mark.parametrize('N,widths', [(7, (5, 64)), (32, (5, 21, 32, 64)), (64, (18, 33, 41, 64)), (3, (256, 192, 160, 64)), (9, (5, 16, 21, 32, 64)), (5, (5, 32, 64)),
                                      (6, (5, 10, 12, 16, 21, 32, 64)), (4, (7, 8, 9, 10, 12, 16, 21, 32, 64))])
def test_general_depth_q_mlp_kernels_against_torch(N, widths):
    """rumpy_q_mlpn_*: ParaCALayer with num_layers = 1, 3, 4, 6, 8 (and 2, where the gates must be BITWISE those of the two-layer kernels): gates, every
    parameter gradient and the gradient at the metadata input against torch autograd; (256, 192, 160, 64) = three layers on the 256-entry embedding"""
    dev = torch.device('cuda:0')
    lib = L.lib()
    Ln, C, M = len(widths) - 1, widths[-1], widths[0]
    g = torch.Generator().manual_seed(6 + Ln)
    items, keep = [], []
    hsum = sum(widths[1:-1])
    for _ in range(3):
        ws = [torch.randn(widths[l + 1], widths[l], generator=g) * (1.5 / widths[l] ** 0.5) for l in range(Ln)]
        bs = [torch.randn(widths[l + 1], generator=g) * 0.2 for l in range(Ln)]
        dz = torch.randn(N, C, generator=g)
        dw, db, ddz = [w.to(dev).contiguous() for w in ws], [b.to(dev).contiguous() for b in bs], dz.to(dev).contiguous()
        gws, gbs = [torch.zeros_like(w) for w in dw], [torch.zeros_like(b) for b in db]
        acts, gate = torch.zeros(N, max(hsum, 1), device=dev), torch.zeros(N, C, device=dev)
        it = L.QMlpNItem(acts=acts.data_ptr(), gate=gate.data_ptr(), dzq=ddz.data_ptr(), nlayers=Ln, scale=0.5)
        for l in range(Ln):
            it.w[l], it.b[l], it.gw[l], it.gb[l] = dw[l].data_ptr(), db[l].data_ptr(), gws[l].data_ptr(), gbs[l].data_ptr()
        for l, wd in enumerate(widths):
            it.n[l] = wd
        items.append(it)
        keep.append((ws, bs, dz, dw, db, ddz, gws, gbs, acts, gate))
    arr = (L.QMlpNItem * len(items))(*items)
    tab = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
    meta = _meta(3, N, M)
    md = meta.to(dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    n = (ctypes.c_int32 * len(widths))(*widths)
    L.check(lib.rumpy_q_mlpn_fwd(tab.data_ptr(), len(items), md.data_ptr(), N, n, Ln, s), 'fwd')
    L.check(lib.rumpy_q_mlpn_bwd_params(tab.data_ptr(), len(items), md.data_ptr(), N, n, Ln, s), 'bwd')
    dmeta = torch.full((N, M), float('nan'), device=dev)
    L.check(lib.rumpy_q_mlpn_bwd_meta(tab.data_ptr(), len(items), N, n, Ln, dmeta.data_ptr(), s), 'bwd_meta')
    torch.cuda.synchronize()

    def net(x, ws, bs):
        hs = []
        for l in range(Ln):
            x = x @ ws[l].t() + bs[l]
            if l + 1 < Ln:
                x = torch.relu(x)
                hs.append(x)
        return x, hs
    mref = meta.clone().requires_grad_(True)
    sum((net(mref, k[0], k[1])[0] * k[2]).sum() for k in keep).backward()
    assert torch.allclose(dmeta.cpu(), 0.5 * mref.grad, atol=3e-4, rtol=2e-4)
    for ws, bs, dz, dw, db, ddz, gws, gbs, acts, gate in keep:
        pw, pb = [w.clone().requires_grad_(True) for w in ws], [b.clone().requires_grad_(True) for b in bs]
        z, hs = net(meta, pw, pb)
        if hs:
            assert torch.allclose(acts.cpu(), torch.cat([h.detach() for h in hs], dim=1), atol=1e-4, rtol=1e-5)
        assert torch.allclose(gate.cpu(), torch.sigmoid(z).detach(), atol=1e-5)
        (z * dz).sum().backward()
        for l in range(Ln):
            assert torch.allclose(gws[l].cpu(), 0.5 * pw[l].grad, atol=3e-4, rtol=2e-4), l
            assert torch.allclose(gbs[l].cpu(), 0.5 * pb[l].grad, atol=3e-4, rtol=2e-4), l
    if Ln == 2:      # the same numbers as the hot-path kernels, bit for bit
        ws, bs, dz, dw, db, ddz, gws, gbs, acts, gate = keep[0]
        outs = [torch.zeros(N, widths[1], device=dev), torch.zeros(N, C, device=dev), torch.zeros_like(dw[0]), torch.zeros_like(db[0]), torch.zeros_like(dw[1]),
                torch.zeros_like(db[1])]
        it2 = L.QMlpItem(w1=dw[0].data_ptr(), b1=db[0].data_ptr(), w2=dw[1].data_ptr(), b2=db[1].data_ptr(), hidden=outs[0].data_ptr(), gate=outs[1].data_ptr(),
                         dzq=ddz.data_ptr(), gw1=outs[2].data_ptr(), gb1=outs[3].data_ptr(), gw2=outs[4].data_ptr(), gb2=outs[5].data_ptr(), scale=0.5)
        tab2 = torch.from_numpy(np.frombuffer(bytes((L.QMlpItem * 1)(it2)), dtype=np.uint8).copy()).to(dev)
        L.check(lib.rumpy_q_mlp_fwd(tab2.data_ptr(), 1, md.data_ptr(), N, M, widths[1], C, s), 'fwd2')
        L.check(lib.rumpy_q_mlp_bwd_params(tab2.data_ptr(), 1, md.data_ptr(), N, M, widths[1], C, s), 'bwd2')
        torch.cuda.synchronize()
        assert torch.equal(outs[1], gate) and torch.equal(outs[0], acts[:, :widths[1]])
        for a, b in zip(outs[2:], (gws[0], gbs[0], gws[1], gbs[1])):
            assert torch.equal(a, b)
    # shapes beyond the kernels' tables are refused
    big = (ctypes.c_int32 * 4)(5, 257, 32, 64)
    assert lib.rumpy_q_mlpn_fwd(tab.data_ptr(), len(items), md.data_ptr(), N, big, 3, s) != 0
    assert lib.rumpy_q_mlpn_fwd(tab.data_ptr(), len(items), md.data_ptr(), 65, n, Ln, s) != 0
    wide = (ctypes.c_int32 * 4)(256, 208, 192, 64)
    assert lib.rumpy_q_mlpn_bwd_params(tab.data_ptr(), len(items), md.data_ptr(), N, wide, 3, s) != 0      # 464 units in all


@pytest.mark.parametrize('depth,names', [(1, ['blur_sigma', 'noise_level', 'jpeg_q']), (3, ['blur_sigma', 'noise_level', 'jpeg_q', 'extra_a', 'extra_b']),
                                         (3, ['m%02d' % i for i in range(18)]), (4, ['qpi']), (6, ['blur_sigma', 'noise_level', 'jpeg_q', 'extra_a', 'extra_b'])])
def test_qrcan_with_other_q_layer_depths_against_oracle(depth, names):
    """`num_layers_in_q_layer` other than 2 (ParaCALayer num_layers, q_layer.py:13,22-41; VERDICT r4 missing 3): three training steps against the oracle
    (pinned on the real reference handler for depths 1, 3 and 6 by golden G24; the HIP path takes 1 .. 8 layers since round 6), every gradient checked at step 0, the q-layers' own among them"""
    kw = dict(scale=2, n_feats=64, n_resgroups=2, n_resblocks=2, reduction=16, num_layers_in_q_layer=depth)
    h, oh = _pair(names, 826, **kw)
    M = len(names)
    keys = [(n, 'numeric') for n in names]
    qk = [k for k, _ in h.net.named_parameters() if 'body.0.body.0.q_node' in k]
    assert len(qk) == 2 * depth, qk
    for step in range(3):
        x, y = O.synthetic_batch(860 + step, 3, lr_hw=16, scale=2)
        m = _meta(870 + step, 3, M)
        loss, out = h.run_train(x=x, y=y, metadata=m, metadata_keys=keys)
        oloss, oout = oh.run_train(x, y, extra_channels=m.unsqueeze(2).unsqueeze(3))
        assert abs(float(loss) - float(oloss)) < (2e-3 if step == 0 else 1e-2) * float(oloss)
        if step == 0:
            assert self_psnr(out, oout) >= 50.0
            _grad_check(h, oh)
            for k, p in h.net.named_parameters():
                if 'q_node' in k:
                    assert float(p.grad.abs().max()) > 0, k
    plan = h.net.engine.plan_for(3, 16, 16, True)
    assert plan.qn_items and not plan.q_items


@pytest.mark.parametrize('feats,wseed', [(128, 837), (192, 837)])
def test_qrcan_wider_than_64_features_against_oracle(feats, wseed):
    """QRCAN 'standard' + q-layers wider than 64 features (round 5: 128; round 6: 192 = three input chunks, 24 channel vectors per pixel): convs on the wide
    kernels, channel attention as its separate launches with the meta-attention gate multiplied in, q-layers (5 -> 64 -> F) on the general-depth launches.
    Weight seed 837: every squeeze-excite hidden unit >= 7e-3 away from its ReLU threshold on these inputs at 128 features (conditioning note of
    test_qrcan_train_steps_against_oracle)."""
    names = ['blur_sigma', 'noise_level', 'jpeg_q', 'extra_a', 'extra_b']
    kw = dict(scale=2, n_feats=feats, n_resgroups=2, n_resblocks=2, reduction=16)
    h, oh = _pair(names, wseed, **kw)
    keys = [(n, 'numeric') for n in names]
    for step in range(2):
        x, y = O.synthetic_batch(860 + step, 3, lr_hw=16, scale=2)
        m = _meta(870 + step, 3, 5)
        loss, out = h.run_train(x=x, y=y, metadata=m, metadata_keys=keys)
        oloss, oout = oh.run_train(x, y, extra_channels=m.unsqueeze(2).unsqueeze(3))
        assert abs(float(loss) - float(oloss)) < (2e-3 if step == 0 else 1e-2) * float(oloss)
        if step == 0:
            assert self_psnr(out, oout) >= 50.0
            _grad_check(h, oh)
            for k, p in h.net.named_parameters():
                if 'q_node' in k:
                    assert float(p.grad.abs().max()) > 0, k
    plan = h.net.engine.plan_for(3, 16, 16, True)
    assert plan.qn_items and not plan.q_items and 'rumpy_rcab_fwd' not in [op for op, _ in plan.fwd]


def test_qrcan_three_layer_q_nodes_return_the_gradient_of_their_metadata_input():
    kw = dict(scale=2, n_feats=64, n_resgroups=2, n_resblocks=2, reduction=16, num_layers_in_q_layer=3)
    h, oh = _pair(['e%03d' % i for i in range(256)], 77, **kw)
    x, _ = O.synthetic_batch(78, 3, lr_hw=16, scale=2)
    m = (torch.randn(3, 256, 1, 1, generator=torch.Generator().manual_seed(79)) * 0.5)
    r = torch.randn(3, 3, 32, 32, generator=torch.Generator().manual_seed(80))
    mo = m.clone().requires_grad_(True)
    oh.net.train()
    (oh.net(x, mo) * r).sum().backward()
    md = m.to('cuda:0').requires_grad_(True)
    h.net.train()
    (h.net(x.to('cuda:0'), md) * r.to('cuda:0')).sum().backward()
    rel = float((md.grad.cpu().double() - mo.grad.double()).norm() / mo.grad.double().norm())
    assert rel < 3e-2, rel


@pytest.mark.parametrize('names,kw', [
    (['blur_sigma', 'noise_level', 'jpeg_q', 'extra_a', 'extra_b'], dict(scale=2, n_feats=64, n_resgroups=2, n_resblocks=2, reduction=16)),
    (['qpi'], dict(scale=4, n_feats=64, n_resgroups=1, n_resblocks=2, reduction=16)),
    (['m%02d' % i for i in range(18)], dict(scale=2, n_feats=64, n_resgroups=1, n_resblocks=1, reduction=16)),     # > 15 entries: wider hidden layer
    (['e%03d' % i for i in range(256)], dict(scale=2, n_feats=64, n_resgroups=2, n_resblocks=2, reduction=16,       # embedding-sized vector,
                                             selective_meta_blocks=[True, False], num_q_layers_inner_residual=1)),   # q-layers in selected blocks only
])
def test_qrcan_train_steps_against_oracle(names, kw):
    # weight seed 826: every squeeze-excite hidden unit sits >= 7e-3 away from its ReLU threshold on these inputs.  (With e.g.
    # seed 821 the only live unit of one block has a pre-activation of 0.002, so the 1e-4 that bf16 activations move the pooled
    # mean by scales that block's conv_du gradients by 4 % - conditioning of the test case, not of the kernels.)
    h, oh = _pair(names, 826, **kw)
    M, sc = len(names), kw['scale']
    keys = [(n, 'numeric') for n in names]
    for step in range(3):
        x, y = O.synthetic_batch(830 + step, 3, lr_hw=16, scale=sc)
        m = _meta(840 + step, 3, M)
        loss, out = h.run_train(x=x, y=y, metadata=m, metadata_keys=keys)
        oloss, oout = oh.run_train(x, y, extra_channels=m.unsqueeze(2).unsqueeze(3))
        # after an update the two trajectories differ by Adam's sign-like first steps on near-zero gradients: 1 % from step 1 on
        assert abs(float(loss) - float(oloss)) < (2e-3 if step == 0 else 1e-2) * float(oloss)
        assert abs(h.get_learning_rate() - oh.get_learning_rate()) < 1e-12
        if step == 0:
            assert self_psnr(out, oout) >= 50.0
            worst = _grad_check(h, oh)
            print('worst grad rel err', worst)
            for k, p in h.net.named_parameters():
                if 'q_node' in k:
                    assert float(p.grad.abs().max()) > 0, k
    assert h.metadata_keys_used_in_training == names


def test_qrcan_returns_the_gradient_of_its_metadata_input():
    """a metadata tensor with requires_grad (the embedding of a jointly trained encoder) gets d loss / d metadata from the network's autograd
    node, against the oracle's autograd"""
    kw = dict(scale=2, n_feats=64, n_resgroups=2, n_resblocks=2, reduction=16, selective_meta_blocks=[True, True], num_q_layers_inner_residual=1)
    h, oh = _pair(['e%03d' % i for i in range(256)], 77, **kw)
    x, _ = O.synthetic_batch(78, 3, lr_hw=16, scale=2)
    m = (torch.randn(3, 256, 1, 1, generator=torch.Generator().manual_seed(79)) * 0.5)
    r = torch.randn(3, 3, 32, 32, generator=torch.Generator().manual_seed(80))
    mo = m.clone().requires_grad_(True)
    oh.net.train()
    (oh.net(x, mo) * r).sum().backward()
    md = m.to('cuda:0').requires_grad_(True)
    h.net.train()
    (h.net(x.to('cuda:0'), md) * r.to('cuda:0')).sum().backward()
    assert md.grad is not None and md.grad.shape == (3, 256, 1, 1)
    rel = float((md.grad.cpu().double() - mo.grad.double()).norm() / mo.grad.double().norm())
    assert rel < 3e-2, rel
    # without a gradient path of its own the metadata gets none (and nothing extra is launched)
    md2 = m.to('cuda:0')
    (h.net(x.to('cuda:0'), md2) * r.to('cuda:0')).sum().backward()
    assert md2.grad is None


def test_qrcan_metadata_is_selected_by_key_and_changes_the_output():
    names = ['noise_level', 'blur_sigma']
    h, oh = _pair(names, 822, eval_mode=True, scale=2, n_feats=64, n_resgroups=1, n_resblocks=2, reduction=16)
    x, y = O.synthetic_batch(850, 2, lr_hw=20, scale=2)
    # the dataset delivers more attributes than the model was built for: QModel keeps the columns whose key is in its list
    keys = [('jpeg_q',), ('blur_sigma',), ('unused',), ('noise_level',)]
    full = _meta(851, 2, 4)
    out, loss, _ = h.run_eval(x=x, y=y, request_loss=True, metadata=full, metadata_keys=keys)
    picked = full[:, [1, 3]].unsqueeze(2).unsqueeze(3)
    oout, oloss, _ = oh.run_eval(x, y, request_loss=True, extra_channels=picked)
    assert self_psnr(out, oout) >= 50.0 and abs(float(loss) - float(oloss)) < 2e-3 * float(oloss)
    out2, _, _ = h.run_eval(x=x, extra_channels=(1.0 - picked))
    assert not torch.equal(out, out2)
    with pytest.raises(RuntimeError):
        h.run_eval(x=x)                                       # no metadata at all
    with pytest.raises(RuntimeError):
        h.run_eval(x=x, extra_channels=torch.zeros(2, 3, 1, 1))   # wrong width


def test_qrcan_checkpoint_roundtrip_and_oracle_interchange():
    names = ['a', 'b', 'c']
    kw = dict(scale=2, n_feats=64, n_resgroups=1, n_resblocks=2, reduction=16)
    h, oh = _pair(names, 823, **kw)
    keys = [(n,) for n in names]
    x, y = O.synthetic_batch(860, 2, lr_hw=16, scale=2)
    m = _meta(861, 2, 3)
    h.run_train(x=x, y=y, metadata=m, metadata_keys=keys)
    h.set_epoch(4)
    h.save_model('train_model')
    path = os.path.join(h.model_save_dir, 'train_model_4')
    st = torch.load(path, map_location='cpu', weights_only=False)
    assert st['model_name'] == 'qrcan' and st['metadata_keys_used_in_training'] == names
    assert list(st['network'].keys()) == list(oh.net.state_dict().keys())
    # optimizer state indices follow the reference's registration order: parameter 0 is final_body.weight
    n_par = len(list(oh.net.parameters()))
    assert sorted(st['optimizer']['state'].keys()) == list(range(n_par))
    assert tuple(st['optimizer']['state'][0]['exp_avg'].shape) == tuple(oh.net.final_body.weight.shape)
    h2, oh2 = _pair(names, 999, **kw)
    h2.model_save_dir = h.model_save_dir
    h2.load_model('train_model', 4)
    oh2.net.load_state_dict(st['network'])
    oh2.optimizer.load_state_dict(st['optimizer'])
    x2, y2 = O.synthetic_batch(862, 2, lr_hw=16, scale=2)
    m2 = _meta(863, 2, 3)
    l_a, o_a = h.run_train(x=x2, y=y2, metadata=m2, metadata_keys=keys)
    l_b, o_b = h2.run_train(x=x2, y=y2, metadata=m2, metadata_keys=keys)
    assert torch.equal(o_a, o_b) and float(l_a) == float(l_b)
    l_o, o_o = oh2.run_train(x2, y2, extra_channels=m2.unsqueeze(2).unsqueeze(3))
    assert self_psnr(o_b, o_o) >= 50.0 and abs(float(l_b) - float(l_o)) < 3e-3 * float(l_o)


@pytest.mark.parametrize('no_rcab', ['0', '1'])
def test_qrcan_default_modulate_style_against_oracle(no_rcab, monkeypatch):
    """the handler's default configuration: one quality value per image -> scale_qpi -> attention vector * attributes, through both
    the one-launch RCAB kernels and the separate attention launches"""
    monkeypatch.setenv('RUMPY_NO_RCAB', no_rcab)
    kw = dict(scale=2, n_feats=64, n_resgroups=2, n_resblocks=2, reduction=16)
    h = define_model('qrcan', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False,
                     metadata_list=None, lr=1e-3, **SCHED, **kw)
    assert h.style == 'modulate'
    onet = O.build_oracle('qrcan', style='modulate', include_q_layer=False, **kw)
    assert list(onet.state_dict().keys()) == list(h.net.state_dict().keys())
    sd = O.seeded_state_dict(onet, 826)
    onet.load_state_dict(sd)
    h.net.load_state_dict(sd)
    oh = O.OracleHandler(onet, lr=1e-3, scheduler=SCHED['scheduler'], scheduler_params=SCHED['scheduler_params'])
    for step in range(2):
        x, y = O.synthetic_batch(930 + step, 3, lr_hw=16, scale=2)
        q = _meta(940 + step, 3, 1)
        loss, out = h.run_train(x=x, y=y, metadata=q, metadata_keys=[('qpi',)])
        oloss, oout = oh.run_train(x, y, extra_channels=O.scale_qpi(q.unsqueeze(2).unsqueeze(3), n_feats=64))
        assert abs(float(loss) - float(oloss)) < (2e-3 if step == 0 else 1e-2) * float(oloss)
        if step == 0:
            assert self_psnr(out, oout) >= 50.0
            print('worst grad rel err', _grad_check(h, oh))
    xe, ye = O.synthetic_batch(950, 2, lr_hw=(13, 22), scale=2)
    qe = _meta(951, 2, 1)
    out, loss, _ = h.run_eval(x=xe, y=ye, request_loss=True, metadata=qe, metadata_keys=[('qpi',)])
    oout, oloss, _ = oh.run_eval(xe, ye, request_loss=True, extra_channels=O.scale_qpi(qe.unsqueeze(2).unsqueeze(3), n_feats=64))
    assert self_psnr(out, oout) >= 45.0 and abs(float(loss) - float(oloss)) < 1e-2 * float(oloss)
    assert h.net.engine.exchange_status() == 0


def test_unsupported_qrcan_variants_are_refused():
    for bad in (dict(style='no_such_style'), dict(style='modulate', include_q_layer=True), dict(style='standard', include_pixel_attention=True), dict(style='standard', include_sft_layer=True),
                dict(style='standard', srmd_mode=True), dict(style='standard', use_moco=True)):
        with pytest.raises(RuntimeError):
            define_model('qrcan', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=True, n_resgroups=1, n_resblocks=1, **bad)


@pytest.mark.parametrize('no_rcab', ['0', '1'])
@pytest.mark.parametrize('style', ['standard', 'modulate'])
def test_qrcan_wide_images_one_launch_rcab_and_the_separate_launches(style, no_rcab, monkeypatch):
    """images wider than 48 pixels: the one-launch RCAB kernels as column tiles (round 3), and - RUMPY_NO_RCAB=1, or more strips per image than
    CUs - the one-launch conv pair with pool sums + the fused attention launches (with the meta-attention gate as an extra factor); both carry
    training and evaluation against the oracle"""
    monkeypatch.setenv('RUMPY_NO_RCAB', no_rcab)
    kw = dict(scale=2, n_feats=64, n_resgroups=1, n_resblocks=2, reduction=16)
    if style == 'standard':
        names = ['a', 'b', 'c', 'd']
        h, oh = _pair(names, 826, **kw)
        M = 4
        attr = lambda seed, n: _meta(seed, n, M).unsqueeze(2).unsqueeze(3)
    else:
        h = define_model('qrcan', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False,
                         metadata_list=None, lr=1e-3, **SCHED, **kw)
        onet = O.build_oracle('qrcan', style='modulate', include_q_layer=False, **kw)
        sd = O.seeded_state_dict(onet, 826)
        onet.load_state_dict(sd)
        h.net.load_state_dict(sd)
        oh = O.OracleHandler(onet, lr=1e-3, scheduler=SCHED['scheduler'], scheduler_params=SCHED['scheduler_params'])
        attr = lambda seed, n: O.scale_qpi(_meta(seed, n, 1).unsqueeze(2).unsqueeze(3), n_feats=64)
    x, y = O.synthetic_batch(960, 2, lr_hw=(20, 60), scale=2)
    a = attr(961, 2)
    loss, out = h.run_train(x=x, y=y, extra_channels=a)
    oloss, oout = oh.run_train(x, y, extra_channels=a)
    plan = h.net.engine.plan_for(2, 20, 60, True)
    ops = [op for op, _ in plan.fwd]
    assert ('rumpy_rcab2_fwd' in ops) == (no_rcab == '0') and ('rumpy_conv_block' in ops) == (no_rcab == '1')
    assert self_psnr(out, oout) >= 50.0 and abs(float(loss) - float(oloss)) < 2e-3 * float(oloss)
    print('worst grad rel err', _grad_check(h, oh))
    xe, ye = O.synthetic_batch(962, 1, lr_hw=(33, 70), scale=2)
    ae = attr(963, 1)
    ev, evl, _ = h.run_eval(x=xe, y=ye, request_loss=True, extra_channels=ae)
    oev, oevl, _ = oh.run_eval(xe, ye, request_loss=True, extra_channels=ae)
    assert self_psnr(ev, oev) >= 45.0 and abs(float(evl) - float(oevl)) < 1e-2 * float(oevl)

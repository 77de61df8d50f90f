"""Generate the golden fixtures G1-G9 by importing the REAL reference.

Runs ONLY in the build container, where /root/reference exists (torch 2.10.0 CPU).  The reference
source never travels: only the arrays written next to this script do.  Usage:

    python tests/golden/make_golden.py

Inputs and weights come from numpy.random.default_rng(seed) through oracle.sr_oracle.seeded_state_dict /
synthetic_batch, so every test can regenerate them bit-for-bit without the reference; the fixtures hold the
reference's OUTPUTS (plus the small inputs, for self-containment).
"""
import collections
import collections.abc
import importlib.abc
import importlib.machinery
import json
import os
import sys
import tempfile
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

# ---- import shim: py3.10 collections.Callable + MagicMock for absent NON-arithmetic deps ----
collections.Callable = collections.abc.Callable
ABSENT = {'timm', 'torchvision', 'colorama', 'toml', 'moviepy', 'skimage', 'deepdiff', 'torchinfo',
          'prefetch_generator', 'click_config_file', 'lpips', 'h5py', 'skvideo', 'umap', 'aim'}


class _Loader(importlib.abc.Loader):
    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__ = []
        m.__spec__ = spec
        return m

    def exec_module(self, m):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in ABSENT:
            return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)


sys.meta_path.append(_Finder())
sys.path.insert(0, REF)

from oracle import sr_oracle as O  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)


def npy(t):
    return t.detach().cpu().numpy().copy()


def grads_of(module):
    return {k: npy(p.grad) for k, p in module.named_parameters()}


def main():
    from rumpy.SISR.models.advanced import common as rcommon
    from rumpy.SISR.models.advanced import architectures as rarch
    from rumpy.shared_framework.models import define_model
    from rumpy.image_tools.image_manipulation.image_functions import ycbcr_convert
    from rumpy.sr_tools.metrics import psnr as ref_psnr

    rng = np.random.default_rng(7)
    out = {}

    # ---------------- G1: single default_conv(64,64,3) fwd + grads on [2,64,12,12] ----------------
    conv = rcommon.default_conv(64, 64, 3)
    conv.load_state_dict(O.seeded_state_dict(conv, 101))
    x = torch.from_numpy(rng.standard_normal((2, 64, 12, 12)).astype(np.float32)).requires_grad_(True)
    gy = torch.from_numpy(rng.standard_normal((2, 64, 12, 12)).astype(np.float32))
    y = conv(x)
    y.backward(gy)
    np.savez(os.path.join(HERE, 'g1_conv.npz'), x=npy(x), gy=npy(gy), y=npy(y), gx=npy(x.grad),
             gw=npy(conv.weight.grad), gb=npy(conv.bias.grad))

    # ---------------- G2: ResBlock(res_scale=0.1) fwd/bwd ----------------
    blk = rcommon.ResBlock(rcommon.default_conv, 64, 3, res_scale=0.1)
    blk.load_state_dict(O.seeded_state_dict(blk, 102))
    x = torch.from_numpy(rng.standard_normal((2, 64, 12, 12)).astype(np.float32)).requires_grad_(True)
    gy = torch.from_numpy(rng.standard_normal((2, 64, 12, 12)).astype(np.float32))
    y = blk(x)
    y.backward(gy)
    d = dict(x=npy(x), gy=npy(gy), y=npy(y), gx=npy(x.grad))
    d.update({'g.' + k: v for k, v in grads_of(blk).items()})
    np.savez(os.path.join(HERE, 'g2_resblock.npz'), **d)

    # ---------------- G3: CALayer and RCAB fwd/bwd ----------------
    ca = rarch.CALayer(64, 16)
    ca.load_state_dict(O.seeded_state_dict(ca, 103))
    x = torch.from_numpy(rng.standard_normal((2, 64, 12, 12)).astype(np.float32)).requires_grad_(True)
    gy = torch.from_numpy(rng.standard_normal((2, 64, 12, 12)).astype(np.float32))
    y = ca(x)
    y.backward(gy)
    d = dict(x=npy(x), gy=npy(gy), y=npy(y), gx=npy(x.grad))
    d.update({'g.' + k: v for k, v in grads_of(ca).items()})
    np.savez(os.path.join(HERE, 'g3_calayer.npz'), **d)

    rcab = rarch.RCAB(rcommon.default_conv, 64, 3, 16, res_scale=0.5)  # res_scale must be IGNORED (ref :79-84)
    rcab.load_state_dict(O.seeded_state_dict(rcab, 104))
    x = torch.from_numpy(rng.standard_normal((2, 64, 12, 12)).astype(np.float32)).requires_grad_(True)
    y = rcab(x)
    y.backward(gy)
    d = dict(x=npy(x), gy=npy(gy), y=npy(y), gx=npy(x.grad))
    d.update({'g.' + k: v for k, v in grads_of(rcab).items()})
    np.savez(os.path.join(HERE, 'g3_rcab.npz'), **d)

    # ---------------- G4: Upsampler(x4) fwd/bwd + bare PixelShuffle channel order ----------------
    up = rcommon.Upsampler(rcommon.default_conv, 4, 16)
    up.load_state_dict(O.seeded_state_dict(up, 105))
    x = torch.from_numpy(rng.standard_normal((1, 16, 6, 5)).astype(np.float32)).requires_grad_(True)
    y = up(x)
    gy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(gy)
    d = dict(x=npy(x), gy=npy(gy), y=npy(y), gx=npy(x.grad))
    d.update({'g.' + k: v for k, v in grads_of(up).items()})
    ps_in = torch.arange(2 * 8 * 3 * 2, dtype=torch.float32).reshape(2, 8, 3, 2)
    d['ps_in'] = npy(ps_in)
    d['ps_out'] = npy(torch.nn.PixelShuffle(2)(ps_in))
    np.savez(os.path.join(HERE, 'g4_upsampler.npz'), **d)

    # ---------------- G5: reduced EDSR / RCAN full run_train steps through the REAL handlers ----------------
    tmp = tempfile.mkdtemp()
    sched = dict(scheduler='cosine_annealing_warm_restarts',
                 scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7})
    for name, kw, wseed in (
            ('edsr', dict(scale=4, num_features=16, num_blocks=2, res_scale=0.1, lr=1e-3), 201),
            ('rcan', dict(scale=4, n_feats=16, n_resgroups=2, n_resblocks=2, reduction=4, lr=1e-3), 202)):
        h = define_model(name, model_save_dir=tmp, device=torch.device('cpu'), eval_mode=False,
                         checkpoint_load=False, loss_masking=False, metadata_list=None, **kw, **sched)
        h.net.load_state_dict(O.seeded_state_dict(h.net, wseed))
        d = {}
        for step in range(3):
            xb, yb = O.synthetic_batch(300 + step, 2, lr_hw=12, scale=4)
            loss, o = h.run_train(x=xb, y=yb, tag=None, mask=None)
            d['loss%d' % step] = np.asarray(loss)
            d['lr_after%d' % step] = np.asarray(h.get_learning_rate())
            if step == 0:
                d['out0'] = npy(o)
                for k, p in h.net.named_parameters():
                    d['grad0.' + k] = npy(p.grad)
                for k, v in h.net.state_dict().items():
                    d['w1.' + k] = npy(v)
        for k, v in h.net.state_dict().items():
            d['w3.' + k] = npy(v)
        # G9: checkpoint layout
        state = h.save_model('train_model', extract_state_only=True)
        ck = {'top_keys': sorted(state.keys()),
              'optimizer_keys': sorted(state['optimizer'].keys()),
              'optimizer_param_group_keys': sorted(state['optimizer']['param_groups'][0].keys()),
              'optimizer_state_entry_keys': sorted(state['optimizer']['state'][0].keys()),
              'scheduler_keys': sorted(state['scheduler_G'].keys()),
              'model_name': state['model_name'], 'model_epoch': state['model_epoch'],
              'n_params': len(state['optimizer']['state'])}
        with open(os.path.join(HERE, 'g9_checkpoint_%s.json' % name), 'w') as f:
            json.dump(ck, f, indent=1, sort_keys=True)
        ev, evl, _ = h.run_eval(x=xb, y=yb, request_loss=True)
        d['eval_out'] = npy(ev)
        d['eval_loss'] = np.asarray(evl)
        np.savez(os.path.join(HERE, 'g5_%s_small_train.npz' % name), **d)

    # ---------------- G6: full EDSR-baseline / RCAN forward ----------------
    h = define_model('edsr', model_save_dir=tmp, device=torch.device('cpu'), eval_mode=True,
                     checkpoint_load=False, loss_masking=False, scale=4)
    h.net.load_state_dict(O.seeded_state_dict(h.net, 401))
    xb, _ = O.synthetic_batch(1234, 2, lr_hw=48, scale=4)
    o, _, _ = h.run_eval(x=xb)
    np.savez(os.path.join(HERE, 'g6_edsr_full_fwd.npz'), out=npy(o))
    edsr_keys = [(k, list(v.shape)) for k, v in h.net.state_dict().items()]
    edsr_count = int(h.print_parameters())

    h = define_model('rcan', model_save_dir=tmp, device=torch.device('cpu'), eval_mode=True,
                     checkpoint_load=False, loss_masking=False, scale=4)
    h.net.load_state_dict(O.seeded_state_dict(h.net, 402))
    xb, _ = O.synthetic_batch(1235, 1, lr_hw=24, scale=4)
    o, _, _ = h.run_eval(x=xb)
    np.savez(os.path.join(HERE, 'g6_rcan_full_fwd.npz'), out=npy(o))
    rcan_keys = [(k, list(v.shape)) for k, v in h.net.state_dict().items()]
    rcan_count = int(h.print_parameters())

    # default initialisation under the reference's default seed (net_train.py:20): pins constructor RNG order
    init = {}
    for name, kw in (('edsr', dict(scale=4)), ('rcan', dict(scale=4, n_resgroups=2, n_resblocks=2))):
        torch.manual_seed(8)
        hh = define_model(name, model_save_dir=tmp, device=torch.device('cpu'), eval_mode=True, checkpoint_load=False,
                          loss_masking=False, **kw)
        sd = hh.net.state_dict()
        keys = list(sd.keys())
        for k in (keys[0], keys[1], keys[len(keys) // 2], keys[-2], keys[-1]):
            init[name + '/' + k] = npy(sd[k])
        init[name + '/checksum'] = np.asarray(sum(float(v.double().sum()) for v in sd.values()))
    np.savez(os.path.join(HERE, 'g8_init_seed8.npz'), **init)
    torch.manual_seed(0)

    big = rarch.EDSR(net_features=256, num_blocks=32, scale=4, res_scale=0.1)
    big_count = int(sum(p.numel() for p in big.parameters()))

    # ---------------- G8: parameter counts + key/shape lists ----------------
    with open(os.path.join(HERE, 'g8_params.json'), 'w') as f:
        json.dump({'edsr_baseline_count': edsr_count, 'rcan_count': rcan_count, 'edsr_full_count': big_count,
                   'stats_py_238': {'rcan': 15592355, 'edsr': 43089923},
                   'edsr_keys': edsr_keys, 'rcan_keys': rcan_keys}, f, indent=0)

    # ---------------- G7: net_run_and_process post-processing + Y-PSNR on a Set5 pair ----------------
    from PIL import Image
    set5 = os.path.join(REF, 'Data', 'example_data', 'Set5')
    hr_dir, lr_dir = os.path.join(set5, 'hr'), os.path.join(set5, 'lr_random_blur')
    hr_name = sorted(os.listdir(hr_dir))[0]
    lr_name = sorted(n for n in os.listdir(lr_dir) if n.endswith('.png'))[0]
    hr_im = np.asarray(Image.open(os.path.join(hr_dir, hr_name)).convert('RGB'))
    lr_im = np.asarray(Image.open(os.path.join(lr_dir, lr_name)).convert('RGB'))
    # ToTensor semantics (data_handler.py:472): uint8 HWC -> float32 CHW / 255; centre on a 32x32 LR window
    lr_c = lr_im[16:48, 16:48]
    hr_c = hr_im[64:192, 64:192]
    lr_t = torch.from_numpy(lr_c.transpose(2, 0, 1).astype(np.float32) / 255.).unsqueeze(0)
    hr_t = torch.from_numpy(hr_c.transpose(2, 0, 1).astype(np.float32) / 255.).unsqueeze(0)
    h = define_model('edsr', model_save_dir=tmp, device=torch.device('cpu'), eval_mode=True,
                     checkpoint_load=False, loss_masking=False, scale=4, num_blocks=4)
    h.net.load_state_dict(O.seeded_state_dict(h.net, 403))
    o, loss, _ = h.run_eval(x=lr_t, y=hr_t, request_loss=True)
    rgb = np.clip(np.copy(o.numpy()), 0, 1)                     # base_interface.py:216-222
    ycbcr = np.copy(rgb)
    for i in range(ycbcr.shape[0]):                             # base_interface.py:209-214
        ycbcr[i] = ycbcr_convert(ycbcr[i], im_type='jpg', input='rgb', y_only=False)
    hr_ycbcr = np.copy(np.clip(hr_t.numpy(), 0, 1))
    for i in range(hr_ycbcr.shape[0]):                          # standard_eval.py:278-287
        hr_ycbcr[i] = ycbcr_convert(hr_ycbcr[i], im_type='jpg', input='rgb', y_only=False)
    p = ref_psnr(ycbcr[:, 0, :, :], hr_ycbcr[:, 0, :, :], max_value=1)   # metrics.py:109-121
    np.savez(os.path.join(HERE, 'g7_eval_set5.npz'), lr=lr_c, hr=hr_c, out=npy(o), rgb=rgb, ycbcr=ycbcr,
             hr_ycbcr=hr_ycbcr, psnr=np.asarray(p, dtype=np.float64), loss=np.asarray(loss),
             names=np.array([hr_name, lr_name]))
    print('golden fixtures written to', HERE)
    for fn in sorted(os.listdir(HERE)):
        print('  %-32s %8d B' % (fn, os.path.getsize(os.path.join(HERE, fn))))


if __name__ == '__main__':
    main()

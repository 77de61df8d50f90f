"""Golden fixture G14 (QRCAN in the handler's DEFAULT style 'modulate', metadata = the default ['qpi']) from the REAL reference handler.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_qrcan_modulate.py
define_model('qrcan') with the reference defaults for style / metadata (reduced: 16 features, 2 groups x 2 blocks, x2) driven through
QModel.run_train / run_eval with one quality value per image: generate_channels -> scale_qpi (gaussian bump over the channel axis,
handlers.py:59-73) -> QCALayer 'modulate' (attention vector * attributes).  Also stores scale_qpi's output for seeded inputs
(n_feats = 16 and 64, clamp on and off) as the known-answer vectors of the host-side mirror.
"""
import os
import runpy
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
shim = runpy.run_path(os.path.join(HERE, 'make_golden.py'), run_name='shim_only')
O = shim['O']
from rumpy.shared_framework.models import define_model  # noqa: E402


def qpi_batch(seed, n):
    return torch.from_numpy(np.random.default_rng(seed).uniform(0, 1, (n, 1)).astype(np.float32))


def main():
    torch.manual_seed(0)
    kw = dict(scale=2, n_feats=16, n_resgroups=2, n_resblocks=2, reduction=16, lr=1e-3,
              scheduler='cosine_annealing_warm_restarts', scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7})
    h = define_model('qrcan', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, checkpoint_load=False,
                     loss_masking=False, metadata_list=None, **kw)
    assert h.style == 'modulate' and h.metadata == ['qpi']
    h.net.load_state_dict(O.seeded_state_dict(h.net, 851))
    keys = [('qpi',)]
    d = {'keys': np.array(list(h.net.state_dict().keys()))}
    for step in range(3):
        xb, yb = O.synthetic_batch(860 + step, 2, lr_hw=12, scale=2)
        loss, o = h.run_train(x=xb, y=yb, metadata=qpi_batch(870 + step, 2), metadata_keys=keys)
        d['loss%d' % step] = np.asarray(loss)
        if step == 0:
            d['out0'] = o.detach().numpy()
            for k, p in h.net.named_parameters():
                d['grad0.' + k] = p.grad.detach().numpy().copy()
    for k, v in h.net.state_dict().items():
        d['w3.' + k] = v.detach().numpy().copy()
    xe, ye = O.synthetic_batch(880, 1, lr_hw=10, scale=2)
    ev, evl, _ = h.run_eval(x=xe, y=ye, request_loss=True, metadata=qpi_batch(881, 1), metadata_keys=keys)
    d['eval_out'], d['eval_loss'] = ev.detach().numpy(), np.asarray(evl)
    q = qpi_batch(890, 5).unsqueeze(2).unsqueeze(3)
    d['kat_q'] = q.numpy()
    d['kat_16'] = h.scale_qpi(q).numpy()
    h64 = define_model('qrcan', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=True, checkpoint_load=False,
                       loss_masking=False, metadata_list=None, n_resgroups=1, n_resblocks=1, clamp=True, min_mu=-0.1, max_mu=0.9)
    d['kat_64_clamped'] = h64.scale_qpi(q).numpy()
    np.savez_compressed(os.path.join(HERE, 'g14_qrcan_modulate_small_train.npz'), **d)
    print('wrote g14; params', sum(p.numel() for p in h.net.parameters()))


if __name__ == '__main__':
    main()

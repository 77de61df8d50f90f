"""Golden fixture G22 (EDSR beyond 64 features and the x3 upsampler) from the REAL reference handlers.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_wide.py
define_model('edsr', num_features=128, num_blocks=2, scale=3) and define_model('rcan', scale=3, ...) on the CPU, states from
oracle.sr_oracle.seeded_state_dict, one run_train + one run_eval each: loss, strided output, every gradient's norm and a strided sample."""
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

CASES = {'edsr128x3': ('edsr', dict(scale=3, num_features=128, num_blocks=2, res_scale=0.1)),
         'edsr256x2': ('edsr', dict(scale=2, num_features=256, num_blocks=1, res_scale=0.1)),
         'rcanx3': ('rcan', dict(scale=3, n_resgroups=2, n_resblocks=2, n_feats=16, reduction=4))}


def main():
    torch.manual_seed(0)
    d = {}
    for tag, (name, kw) in CASES.items():
        h = define_model(name, model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, checkpoint_load=False,
                         loss_masking=False, lr=1e-3, **kw)
        o = O.build_oracle(name, **kw)
        assert list(o.state_dict().keys()) == list(h.net.state_dict().keys())
        h.net.load_state_dict(O.seeded_state_dict(o, 3000))
        x, y = O.synthetic_batch(3010, 2, lr_hw=12, scale=kw['scale'])
        loss, out = h.run_train(x=x, y=y)
        d[tag + '.loss'] = np.asarray(loss)
        d[tag + '.out'] = out.numpy()[:, :, ::3, ::3].copy()
        for k, p in h.net.named_parameters():
            g = p.grad.detach().numpy().reshape(-1)
            d['%s.gnorm.%s' % (tag, k)] = np.asarray(np.linalg.norm(g.astype(np.float64)))
            d['%s.gsample.%s' % (tag, k)] = g[::211].copy()
        xe, ye = O.synthetic_batch(3020, 1, lr_hw=(10, 14), scale=kw['scale'])
        ev, evl, _ = h.run_eval(x=xe, y=ye, request_loss=True)
        d[tag + '.eval_out'] = ev.numpy()[:, :, ::3, ::3].copy()
        d[tag + '.eval_loss'] = np.asarray(evl)
        d[tag + '.params'] = np.asarray(sum(p.numel() for p in h.net.parameters()))
        print(tag, float(loss), float(evl), int(d[tag + '.params']))
    np.savez_compressed(os.path.join(HERE, 'g22_wide_x3.npz'), **d)


if __name__ == '__main__':
    main()

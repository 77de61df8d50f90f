"""Golden fixture G19 (QRCAN with the QCALayer styles 'max_concat', 'mini_concat', 'extended_attention', 'softmax') from the REAL reference handler.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_qrcan_styles.py
define_model('qrcan', style=<style>, metadata=[5 names]) (reduced: 16 features, 2 groups x 2 blocks, x2, no q-layers) driven through
QModel.run_train (three steps) / run_eval: per style the losses, the first step's output and gradients, the weights after three steps,
an evaluation output, and the default initialisation under torch.manual_seed(8) as per-tensor checksums (creation order of the layers).
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
from rumpy.SISR.models.attention_manipulators.architectures import QRCAN as RefQRCAN  # noqa: E402

META = ['blur_sigma', 'noise_level', 'jpeg_q', 'extra_a', 'extra_b']
STYLES = ('max_concat', 'mini_concat', 'extended_attention', 'softmax')


def meta_batch(seed, n):
    return torch.from_numpy(np.random.default_rng(seed).uniform(0, 1, (n, len(META))).astype(np.float32))


def main():
    d = {}
    keys = [(m,) for m in META]
    for si, style in enumerate(STYLES):
        torch.manual_seed(0)
        kw = dict(scale=2, n_feats=16, n_resgroups=2, n_resblocks=2, reduction=16, style=style, include_q_layer=False, lr=1e-3,
                  scheduler='cosine_annealing_warm_restarts', scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7})
        h = define_model('qrcan', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, checkpoint_load=False,
                         loss_masking=False, metadata_list=None, metadata=list(META), **kw)
        torch.manual_seed(8)
        r8 = RefQRCAN(scale=2, n_feats=16, n_resgroups=2, n_resblocks=2, reduction=16, style=style, include_q_layer=False, num_metadata=len(META))
        d[style + '.init8'] = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in r8.state_dict().values()])
        h.net.load_state_dict(O.seeded_state_dict(h.net, 900 + si))
        d[style + '.keys'] = np.array(list(h.net.state_dict().keys()))
        for step in range(3):
            xb, yb = O.synthetic_batch(910 + 10 * si + step, 2, lr_hw=12, scale=2)
            loss, o = h.run_train(x=xb, y=yb, metadata=meta_batch(950 + 10 * si + step, 2), metadata_keys=keys)
            d['%s.loss%d' % (style, step)] = np.asarray(loss)
            if step == 0:
                d[style + '.out0'] = o.detach().numpy()
                for k, p in h.net.named_parameters():
                    d['%s.grad0.%s' % (style, k)] = p.grad.detach().numpy().copy()
        for k, v in h.net.state_dict().items():
            d['%s.w3.%s' % (style, k)] = v.detach().numpy().copy()
        xe, ye = O.synthetic_batch(990 + si, 1, lr_hw=10, scale=2)
        ev, evl, _ = h.run_eval(x=xe, y=ye, request_loss=True, metadata=meta_batch(995 + si, 1), metadata_keys=keys)
        d[style + '.eval_out'], d[style + '.eval_loss'] = ev.detach().numpy(), np.asarray(evl)
        print(style, 'params', sum(p.numel() for p in h.net.parameters()), 'losses', [float(d['%s.loss%d' % (style, s)]) for s in range(3)])
    np.savez_compressed(os.path.join(HERE, 'g19_qrcan_styles_small_train.npz'), **d)
    print('wrote g19_qrcan_styles_small_train.npz')


if __name__ == '__main__':
    main()

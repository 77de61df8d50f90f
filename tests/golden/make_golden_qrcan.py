"""Golden fixtures G12 (meta-attention QRCAN) and G24 (the same with ParaCALayer's num_layers = 1, 3 and 6: `num_layers_in_q_layer`,
attention_manipulators/architectures.py:162,182-183) from the REAL reference handler.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_qrcan.py
The reference's define_model('qrcan', style='standard', include_q_layer=True, metadata=[...]) is built on the CPU (reduced: 16
features, 2 groups x 2 blocks, x2), its weights set from oracle.sr_oracle.seeded_state_dict, and driven through the reference's own
QModel.run_train / run_eval with a metadata matrix and metadata_keys (attention_manipulators/__init__.py:186-202): three training
steps (loss, every gradient of step 0, weights after step 0 and 3, learning rates) and one evaluation.
"""
import os
import runpy
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
shim = runpy.run_path(os.path.join(HERE, 'make_golden.py'), run_name='shim_only')
O = shim['O']
from rumpy.shared_framework.models import define_model  # noqa: E402

META = ['blur_sigma', 'noise_level', 'jpeg_q', 'extra_a', 'extra_b']      # 5 metadata entries per image


def meta_batch(seed, n):
    gen = np.random.default_rng(seed)
    return torch.from_numpy(gen.uniform(0, 1, (n, len(META))).astype(np.float32))


def main(depth=2, out='g12_qrcan_small_train.npz'):
    import tempfile
    torch.manual_seed(0)
    kw = dict(scale=2, n_feats=16, n_resgroups=2, n_resblocks=2, reduction=16, style='standard', include_q_layer=True, lr=1e-3,
              scheduler='cosine_annealing_warm_restarts', scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7})
    extra = {} if depth == 2 else {'num_layers_in_q_layer': depth}
    kw.update(extra)
    h = define_model('qrcan', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, checkpoint_load=False,
                     loss_masking=False, metadata_list=None, metadata=list(META), **kw)
    # default initialisation under torch.manual_seed(8): pins the layer CREATION order (one checksum pair per tensor)
    torch.manual_seed(8)
    from rumpy.SISR.models.attention_manipulators.architectures import QRCAN as RefQRCAN
    r8 = RefQRCAN(scale=2, n_feats=16, n_resgroups=2, n_resblocks=2, reduction=16, style='standard', include_q_layer=True, num_metadata=len(META), **extra)
    init8 = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in r8.state_dict().values()])
    h.net.load_state_dict(O.seeded_state_dict(h.net, 811 + (0 if depth == 2 else 10 * depth)))
    keys = [(m,) for m in META]                          # the dataset delivers (name, ...) tuples; QModel reads key[0]
    d = {'keys': np.array(list(h.net.state_dict().keys())), 'init8': init8}
    for step in range(3):
        xb, yb = O.synthetic_batch(700 + step, 2, lr_hw=12, scale=2)
        mb = meta_batch(750 + step, 2)
        loss, o = h.run_train(x=xb, y=yb, metadata=mb, metadata_keys=keys)
        d['loss%d' % step] = np.asarray(loss)
        d['lr_after%d' % step] = np.asarray(h.get_learning_rate())
        if step == 0:
            d['out0'] = o.detach().numpy()
            for k, p in h.net.named_parameters():
                d['grad0.' + k] = p.grad.detach().numpy().copy()
            for k, v in h.net.state_dict().items():
                d['w1.' + k] = v.detach().numpy().copy()
    for k, v in h.net.state_dict().items():
        d['w3.' + k] = v.detach().numpy().copy()
    xe, ye = O.synthetic_batch(790, 1, lr_hw=10, scale=2)
    me = meta_batch(791, 1)
    ev, evl, _ = h.run_eval(x=xe, y=ye, request_loss=True, metadata=me, metadata_keys=keys)
    d['eval_out'] = ev.detach().numpy()
    d['eval_loss'] = np.asarray(evl)
    d['num_metadata'] = np.asarray(h.num_metadata)
    np.savez_compressed(os.path.join(HERE, out), **d)
    print('wrote', out, '; params', sum(p.numel() for p in h.net.parameters()), 'num_metadata', h.num_metadata)


if __name__ == '__main__':
    main()
    main(1, 'g24_qrcan_qdepth1_small_train.npz')
    main(3, 'g24_qrcan_qdepth3_small_train.npz')
    main(6, 'g24_qrcan_qdepth6_small_train.npz')      # (round 6: the HIP path takes 1 .. 8 layers)

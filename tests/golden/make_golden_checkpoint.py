"""Golden fixture G11 (checkpoint interchange): a checkpoint FILE written by the REAL reference handler.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_checkpoint.py
A reduced EDSR (64 features - the width the HIP path is built for - x 1 block, x2; reference kwargs) is built by the reference's define_model on the CPU, trained for two
steps on seeded batches, saved with the reference's own BaseModel.save_model (-> g11_ref_checkpoint/train_model_2, a torch
pickle: data, not source) and then trained one more step.  The fixture holds the file plus what the reference computed around it
(eval output at the saved state, loss / learning rate / weights of the step after the save), so a test can load the file into the
HIP handler and has to reproduce the reference's continuation.
"""
import os
import runpy
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
# same import shim as make_golden.py (MagicMock for absent non-arithmetic dependencies, /root/reference on sys.path)
shim = runpy.run_path(os.path.join(HERE, 'make_golden.py'), run_name='shim_only')
O = shim['O']
from rumpy.shared_framework.models import define_model  # noqa: E402  (the reference's registry)

OUT = os.path.join(HERE, 'g11_ref_checkpoint')


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    kw = dict(scale=2, num_features=64, num_blocks=1, res_scale=0.1, lr=1e-3, scheduler='cosine_annealing_warm_restarts',
              scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7})
    h = define_model('edsr', model_save_dir=OUT, device=torch.device('cpu'), eval_mode=False, checkpoint_load=False,
                     loss_masking=False, metadata_list=None, **kw)
    h.net.load_state_dict(O.seeded_state_dict(h.net, 411))
    d = {}
    for step in range(2):
        xb, yb = O.synthetic_batch(500 + step, 2, lr_hw=12, scale=2)
        loss, _ = h.run_train(x=xb, y=yb, tag=None, mask=None)
        d['loss%d' % step] = np.asarray(loss)
    h.set_epoch(2)
    h.save_model('train_model')                        # -> OUT/train_model_2
    xe, ye = O.synthetic_batch(600, 1, lr_hw=16, scale=2)
    ev, evl, _ = h.run_eval(x=xe, y=ye, request_loss=True)
    d['eval_out'] = ev.detach().numpy()
    d['eval_loss'] = np.asarray(evl)
    d['lr_at_save'] = np.asarray(h.get_learning_rate())
    xb, yb = O.synthetic_batch(502, 2, lr_hw=12, scale=2)
    loss, _ = h.run_train(x=xb, y=yb, tag=None, mask=None)
    d['loss2'] = np.asarray(loss)
    d['lr_after2'] = np.asarray(h.get_learning_rate())
    for k, v in h.net.state_dict().items():
        d['w3.' + k] = v.detach().numpy()
    np.savez(os.path.join(OUT, 'expected.npz'), **d)
    print('wrote', os.listdir(OUT), os.path.getsize(os.path.join(OUT, 'train_model_2')), 'bytes')


if __name__ == '__main__':
    main()

"""Golden fixture G15 (the reference's "basic" models: SRCNN = BASELINE config 0, and VDSR) from the REAL reference handlers.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_basic.py
define_model('srcnn') (default 9-5-5, 1-64-32-1) and define_model('vdsr') (reduced: 4 layers x 8 channels, its default grad_clip=0.1) are
built on the CPU, their weights set from oracle.sr_oracle.seeded_state_dict, and driven through the reference's own BaseModel.run_train /
run_eval (MSE loss, Adam, per-batch cosine restarts): three training steps (loss, every gradient of step 0, weights after steps 1 and 3,
learning rates) and one evaluation with loss.  Also the 'ycbcr' branch of SISRInterface.net_run_and_process (interface.py:113-121)
restated around the reference's own ycbcr_convert: Y from the network, Cb/Cr from the interpolated input, clip, JPEG-matrix inverse.
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
from rumpy.image_tools.image_manipulation.image_functions import ycbcr_convert  # noqa: E402

SCHED = dict(scheduler='cosine_annealing_warm_restarts', scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7})
CASES = {'srcnn': dict(lr=1e-3, **SCHED),
         'vdsr': dict(lr=1e-3, kernel_pattern=[3, 3, 3, 3], channel_pattern=[1, 8, 8, 8, 1], **SCHED)}


def y_batch(seed, n, h, w):
    """single-channel 'interpolated LR' input and HR target in [0,1) (im_input='interp': both at the HR size, data_handler.py:535-538)"""
    g = np.random.default_rng(seed)
    return (torch.from_numpy(g.uniform(0, 1, (n, 1, h, w)).astype(np.float32)),
            torch.from_numpy(g.uniform(0, 1, (n, 1, h, w)).astype(np.float32)))


def main():
    import tempfile
    d = {}
    for name, kw in CASES.items():
        torch.manual_seed(8)
        h = define_model(name, model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, checkpoint_load=False,
                         loss_masking=False, **kw)
        d[name + '.keys'] = np.array(list(h.net.state_dict().keys()))
        d[name + '.init8'] = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in h.net.state_dict().values()])
        d[name + '.attrs'] = np.array([h.colorspace, h.im_input, h.model_name, type(h.criterion).__name__, str(h.grad_clip)])
        h.net.load_state_dict(O.seeded_state_dict(h.net, 840))
        for step in range(3):
            xb, yb = y_batch(850 + step, 2, 20, 27)
            loss, o = h.run_train(x=xb, y=yb)
            d['%s.loss%d' % (name, step)] = np.asarray(loss)
            d['%s.lr_after%d' % (name, step)] = np.asarray(h.get_learning_rate())
            if step == 0:
                d[name + '.out0'] = o.detach().numpy()
                for k, p in h.net.named_parameters():
                    d['%s.grad0.%s' % (name, k)] = p.grad.detach().numpy().copy()     # after clip_grad_norm_ where the handler clips
                for k, v in h.net.state_dict().items():
                    d['%s.w1.%s' % (name, k)] = v.detach().numpy().copy()
        for k, v in h.net.state_dict().items():
            d['%s.w3.%s' % (name, k)] = v.detach().numpy().copy()
        xe, ye = y_batch(890, 1, 33, 18)
        ev, evl, _ = h.run_eval(x=xe, y=ye, request_loss=True)
        d[name + '.eval_out'] = ev.detach().numpy()
        d[name + '.eval_loss'] = np.asarray(evl)
        # 'ycbcr' branch of net_run_and_process (interface.py:113-121) on a 3-channel YCbCr input, values pushed outside [0,1] on purpose
        g = np.random.default_rng(895)
        lr3 = torch.from_numpy(g.uniform(-0.1, 1.1, (1, 3, 33, 18)).astype(np.float32))
        out_y, _, _ = h.run_eval(lr3[:, 0, :, :].unsqueeze(1))
        out_ycbcr = torch.stack([out_y.squeeze(1), lr3[:, 1, :, :], lr3[:, 2, :, :]], 1)
        rgb = np.clip(np.copy(out_ycbcr.numpy()), 0, 1)                       # colorspace_convert -> _standard_image_formatting
        for i in range(rgb.shape[0]):
            rgb[i] = ycbcr_convert(rgb[i], im_type='jpg', input='ycbcr', y_only=False)
        d[name + '.post_in'] = lr3.numpy()
        d[name + '.post_rgb'] = rgb
        d[name + '.post_ycbcr'] = np.clip(np.copy(out_ycbcr.numpy()), 0, 1)
        print(name, 'params', sum(p.numel() for p in h.net.parameters()), 'losses', [float(d['%s.loss%d' % (name, s)]) for s in range(3)])
    np.savez_compressed(os.path.join(HERE, 'g15_basic_small_train.npz'), **d)
    print('wrote g15_basic_small_train.npz')


if __name__ == '__main__':
    main()

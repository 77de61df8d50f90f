"""Golden fixtures G17 / G18: evaluation Y-PSNR of FULL-DEPTH EDSR-baseline x4 (16 blocks) and RCAN x4 (10 groups x 20 RCAB) in the
regime of a trained model (>= 30 dB), so that the "eval PSNR within +-0.02 dB of the reference" bound can fail.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_psnr.py
The weights are `oracle.sr_oracle.interpolating_state_dict(net, seed)` (regenerated from the seed by every test): a two-stage
interpolating upsampler + a small detail term fed by every body convolution; the input is `oracle.sr_oracle.vignetted_pair` of the
first Set5 HR image of the reference's example data (64 x 64 LR, 256 x 256 HR; the uint8 pair itself is stored).  The REAL reference
handlers evaluate it (run_eval with a loss request), the post-processing of SISRInterface.net_run_and_process is restated around the
reference's own functions as for G7 (clip: base_interface.py:216-222; ycbcr_convert 'jpg': image_functions.py:81-91;
psnr: metrics.py:109-121).  Stored: the pair, the reference's Y-PSNR and L1 loss, its output on every second pixel, its clipped Y plane.
"""
import os
import runpy
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
shim = runpy.run_path(os.path.join(HERE, 'make_golden.py'), run_name='shim_only')
O, REF = shim['O'], shim['REF']
from rumpy.shared_framework.models import define_model  # noqa: E402
from rumpy.image_tools.image_manipulation.image_functions import ycbcr_convert  # noqa: E402
from rumpy.sr_tools.metrics import psnr as ref_psnr  # noqa: E402

CASES = (('g17_edsr_psnr.npz', 'edsr', 501), ('g18_rcan_psnr.npz', 'rcan', 502))


def main():
    from PIL import Image
    hr_dir = os.path.join(REF, 'Data', 'example_data', 'Set5', 'hr')
    name = sorted(os.listdir(hr_dir))[0]
    lr_u8, hr_u8 = O.vignetted_pair(np.asarray(Image.open(os.path.join(hr_dir, name)).convert('RGB')), 64)
    to_t = lambda a: torch.from_numpy(a.transpose(2, 0, 1).astype(np.float32) / 255.).unsqueeze(0)      # ToTensor (data_handler.py:472)
    lr_t, hr_t = to_t(lr_u8), to_t(hr_u8)
    hr_ycbcr = np.copy(np.clip(hr_t.numpy(), 0, 1))
    for i in range(hr_ycbcr.shape[0]):                              # standard_eval.py:278-287
        hr_ycbcr[i] = ycbcr_convert(hr_ycbcr[i], im_type='jpg', input='rgb', y_only=False)
    for fn, model, seed in CASES:
        h = define_model(model, model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=True, checkpoint_load=False,
                         loss_masking=False, scale=4)
        h.net.load_state_dict(O.interpolating_state_dict(h.net, seed))
        o, loss, _ = h.run_eval(x=lr_t, y=hr_t, request_loss=True)
        rgb = np.clip(np.copy(o.numpy()), 0, 1)
        ycbcr = np.copy(rgb)
        for i in range(ycbcr.shape[0]):
            ycbcr[i] = ycbcr_convert(ycbcr[i], im_type='jpg', input='rgb', y_only=False)
        p = ref_psnr(ycbcr[:, 0, :, :], hr_ycbcr[:, 0, :, :], max_value=1)
        np.savez_compressed(os.path.join(HERE, fn), lr=lr_u8, hr=hr_u8, out_s2=o.numpy()[:, :, ::2, ::2].copy(), y=ycbcr[:, 0].copy(),
                            psnr=np.asarray(p, dtype=np.float64), loss=np.asarray(loss), seed=np.asarray(seed), image=np.array(name))
        print('%s: %s seed %d on %s: reference Y-PSNR %.4f dB, L1 %.5f, %d B' % (fn, model, seed, name, p, float(loss),
                                                                                  os.path.getsize(os.path.join(HERE, fn))))


if __name__ == '__main__':
    main()

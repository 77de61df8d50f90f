"""Golden fixture G16: BASELINE config 0 on the reference's own example data - SRCNN x2 evaluation of a Set5 crop.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_srcnn_set5.py
A 96x96 crop of the first image of Data/example_data/Set5/hr is prepared the way SURVEY.md 8(d) describes for `im_input='interp'` models
(PIL bicubic x2 down, then x2 up: image_functions.py:13-41), both images go to YCbCr with the reference's ycbcr_convert(im_type='jpg')
as the dataset does for `colorspace='ycbcr'` models, the REAL reference SRCNNHandler (seeded weights) evaluates the Y plane with a loss
request, and the 'ycbcr' branch of SISRInterface.net_run_and_process (interface.py:113-121) is restated around the reference's functions:
stacked YCbCr, clip, JPEG-matrix inverse, Y-PSNR (metrics.py:109-121).  Only arrays are stored.
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


def main():
    from PIL import Image
    hr_dir = os.path.join(REF, 'Data', 'example_data', 'Set5', 'hr')
    name = sorted(os.listdir(hr_dir))[0]
    hr = Image.open(os.path.join(hr_dir, name)).convert('RGB').crop((64, 64, 160, 160))           # 96 x 96
    lr = hr.resize((48, 48), resample=Image.BICUBIC).resize((96, 96), resample=Image.BICUBIC)      # x2 down, x2 up
    to_t = lambda im: np.asarray(im).transpose(2, 0, 1).astype(np.float32) / 255.                  # ToTensor (data_handler.py:472)
    hr_rgb, lr_rgb = to_t(hr), to_t(lr)
    hr_ycbcr = ycbcr_convert(hr_rgb, im_type='jpg', input='rgb', y_only=False)[None]
    lr_ycbcr = ycbcr_convert(lr_rgb, im_type='jpg', input='rgb', y_only=False)[None]
    h = define_model('srcnn', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=True, checkpoint_load=False,
                     loss_masking=False)
    h.net.load_state_dict(O.seeded_state_dict(h.net, 842))
    lr_t, hr_t = torch.from_numpy(lr_ycbcr.astype(np.float32)), torch.from_numpy(hr_ycbcr.astype(np.float32))
    out_y, loss, _ = h.run_eval(lr_t[:, 0, :, :].unsqueeze(1), y=hr_t[:, 0, :, :].unsqueeze(1), request_loss=True)
    out_ycbcr = torch.stack([out_y.squeeze(1), lr_t[:, 1, :, :], lr_t[:, 2, :, :]], 1)
    ycbcr = np.clip(np.copy(out_ycbcr.numpy()), 0, 1)
    rgb = np.copy(ycbcr)
    for i in range(rgb.shape[0]):
        rgb[i] = ycbcr_convert(rgb[i], im_type='jpg', input='ycbcr', y_only=False)
    p = ref_psnr(ycbcr[:, 0, :, :], np.clip(hr_ycbcr, 0, 1)[:, 0, :, :], max_value=1)
    p_in = ref_psnr(np.clip(lr_ycbcr, 0, 1)[:, 0, :, :], np.clip(hr_ycbcr, 0, 1)[:, 0, :, :], max_value=1)
    np.savez_compressed(os.path.join(HERE, 'g16_srcnn_set5_eval.npz'), lr_ycbcr=lr_ycbcr.astype(np.float32), hr_ycbcr=hr_ycbcr.astype(np.float32),
                        out_y=out_y.numpy(), ycbcr=ycbcr, rgb=rgb, loss=np.asarray(loss), psnr=np.asarray(p, dtype=np.float64),
                        psnr_input=np.asarray(p_in, dtype=np.float64), name=np.array(name))
    print('wrote g16_srcnn_set5_eval.npz: image %s, Y-PSNR of the output %.4f dB (interpolated input %.4f dB), loss %.6f' % (name, p, p_in, float(loss)))


if __name__ == '__main__':
    main()

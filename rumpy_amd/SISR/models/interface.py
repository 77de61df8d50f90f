"""SISR model interface of the MI355X path - the caller-facing layer of rumpy/SISR/models/interface.py:12-131 and
rumpy/shared_framework/models/base_interface.py:23-315 reduced to what drives the hot path:
``train_batch`` -> handler.run_train, ``net_run_and_process`` -> handler.run_eval + clip + RGB->YCbCr('jpg').
Post-processing runs in a HIP kernel (rumpy_eval_post) instead of numpy; results are returned as ndarrays like the
reference's.  Directory / config-diff / metadata bookkeeping of the reference interface is out of scope."""
import os

import numpy as np
import torch

from rumpy_amd import _lib as L
from rumpy_amd.shared_framework.configuration.gpu_check import device_selector
from rumpy_amd.shared_framework.models import define_model
from rumpy_amd.sr_tools.metrics import psnr_from_sse


class SISRInterface:
    def __init__(self, model_loc, experiment, gpu='off', sp_gpu=0, mode='eval', new_params=None, load_epoch=None,
                 scale=None, checkpoint_load=False, loss_masking=False, **kwargs):
        """new_params: {'name': ..., 'internal_params': {...}} as in the reference TOML [model] table."""
        self.scale = scale
        self.device = device_selector(gpu, sp_gpu)
        self.mode = mode
        self.metadata = new_params
        self.name = new_params['name'].lower()
        if scale is not None and scale != new_params['internal_params']['scale']:
            raise Exception('The model loaded has been trained for a different scale, '
                            'and cannot produce the requested images.')
        self.base_folder = os.path.join(model_loc, experiment)
        self.saved_models = os.path.join(self.base_folder, 'saved_models')
        self.model_epoch = 0
        self.model = define_model(self.name, model_save_dir=self.saved_models, device=self.device,
                                  eval_mode=(mode == 'eval'), checkpoint_load=checkpoint_load,
                                  loss_masking=loss_masking, **new_params['internal_params'])
        self.configuration = {'colorspace': self.model.colorspace, 'input': self.model.im_input}
        if load_epoch is not None:
            state = self.model.load_model('train_model', load_epoch, legacy=self.model.legacy_load)
            self.model_epoch = state['model_epoch']
        if gpu == 'multi':
            self.model.set_multi_gpu()

    def train_batch(self, lr, hr, *args, **kwargs):
        return self.model.run_train(x=lr, y=hr, **kwargs)

    def net_run_and_process(self, lr=None, hr=None, **kwargs):
        """-> (rgb ndarray, ycbcr ndarray, loss, timing).  'rgb' models (interface.py:109-112): clip + RGB->YCbCr('jpg') in a HIP kernel.
        'ycbcr' models (SRCNN / VDSR, :113-121): the network sees the Y plane only; Cb / Cr come from the (interpolated) input, the
        stack is clipped to [0,1] and converted back with the JPEG matrix - per-pixel host arithmetic on the returned image, as in
        the reference (the RGB result is NOT clipped again there, and is not here)."""
        if 'rgb' not in self.configuration['colorspace']:
            f_ref = None if hr is None else hr[:, 0, :, :].unsqueeze(1)
            out_y, loss, timing = self.model.run_eval(lr[:, 0, :, :].unsqueeze(1), y=f_ref, **kwargs)
            ycbcr = np.clip(torch.stack([out_y.squeeze(1).cpu(), lr[:, 1, :, :].cpu(), lr[:, 2, :, :].cpu()], 1).numpy(), 0, 1)
            return self.ycbcr_jpg_to_rgb(ycbcr), ycbcr, loss, timing
        out_rgb, loss, timing = self.model.run_eval(x=lr, y=hr, keep_on_device=True, **kwargs)
        rgb, ycbcr, _ = self.postprocess(out_rgb)
        return rgb.cpu().numpy(), ycbcr.cpu().numpy(), loss, timing

    @staticmethod
    def ycbcr_jpg_to_rgb(img, max_val=1.0):
        """[N,3,H,W] YCbCr -> RGB with the full-range JPEG (BT.601) matrix, chroma bias 128/255 of the range
        (image_functions.py:108-121 with im_type='jpg', which is what colorspace_convert passes, base_interface.py:212)."""
        img = np.asarray(img, dtype=np.float32)
        bias = 128. * (max_val / 255)
        y, cb, cr = img[:, 0], img[:, 1], img[:, 2]
        r = y + 1.402 * cr - 1.402 * bias
        g = y - 0.344136 * cb - 0.714136 * cr + (0.714136 + 0.344136) * bias
        b = y + 1.772 * cb - 1.772 * bias
        return np.stack([r, g, b], 1).astype(np.float32)

    @staticmethod
    def postprocess(out, ref=None):
        """clip -> YCbCr('jpg') on the device; with ``ref`` (RGB in [0,1]) also the Y-channel PSNR of the batch."""
        out = out.float().contiguous()
        n, c, h, w = out.shape
        if c != 3:
            raise RuntimeError('postprocess expects RGB output')
        rgb, ycbcr = torch.empty_like(out), torch.empty_like(out)
        part = torch.zeros(1024, dtype=torch.float32, device=out.device)
        sse = torch.zeros(1, dtype=torch.float32, device=out.device)
        refp = None
        if ref is not None:
            ref = ref.to(out.device).float().contiguous()
            refp = ref.data_ptr()
        a = L.EvalPostArgs(out=out.data_ptr(), ref=refp, rgb=rgb.data_ptr(), ycbcr=ycbcr.data_ptr(),
                           sse_partial=part.data_ptr(), sse=sse.data_ptr(), N=n, H=h, W=w)
        L.call('rumpy_eval_post', a, torch.cuda.current_stream(out.device).cuda_stream)
        p = psnr_from_sse(float(sse.item()), n * h * w) if ref is not None else None
        return rgb, ycbcr, p

    def save(self, name='train_model', override=False, dry_run=False, minimal=False):
        prefix = name if not minimal else name + '_minimal'
        path = os.path.join(self.saved_models, '{}_{}'.format(prefix, str(self.model_epoch)))
        if os.path.isfile(path) and not override:
            raise RuntimeError('Saving this model will result in overwriting existing data!  '
                               'Change model location or enable override.')
        if not dry_run:
            os.makedirs(self.saved_models, exist_ok=True)
            self.model.save_model(model_save_name=prefix, minimal=minimal)

    def set_epoch(self, epoch):
        self.model_epoch = epoch
        self.model.set_epoch(epoch)

    def get_learning_rate(self):
        return self.model.get_learning_rate()

    def epoch_end_calls(self):
        self.model.epoch_end_calls()

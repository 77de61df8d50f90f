"""EDSR / RCAN handlers of the MI355X path - same class names, kwargs and attributes as
rumpy/SISR/models/advanced/handlers.py:8-42, so ``define_model('edsr' | 'rcan', **kwargs)`` resolves to them."""
import time

import torch

from rumpy_amd.shared_framework.models.base_architecture import BaseModel
from .architectures import EDSR, RCAN


class _TiledEval(BaseModel):
    """Tiled whole-image evaluation, SURVEY.md 8(f)2: the reference's `forward_chop` (SANHandler, advanced/handlers.py:85-123, and
    ContrastiveBlindQEDSRHandler, blur_kernel_blind_sr/handlers.py:907-945 - the same code twice): quarters that overlap by `shave` pixels,
    each super-resolved on its own (recursively while a quarter still has `max_combined_im_size` pixels or more), the output stitched from
    the quarters' own halves.  The reference's EDSR / RCAN handlers always run the whole image, and so do these unless
    ``max_combined_im_size`` is given (288 GB of HBM hold any image whole: the option exists for parity of results with a reference
    run that tiled, not for memory).  Quarters stay on the device; one copy to the host at the end."""

    max_combined_im_size = None

    def forward_chop(self, x, shave=10):
        b, c, h, w = x.shape
        h_half, w_half = h // 2, w // 2
        h_size, w_size = h_half + shave, w_half + shave
        lr_list = [x[:, :, 0:h_size, 0:w_size], x[:, :, 0:h_size, (w - w_size):w],
                   x[:, :, (h - h_size):h, 0:w_size], x[:, :, (h - h_size):h, (w - w_size):w]]
        if w_size * h_size < self.max_combined_im_size:
            sr_list = [self.run_chopped_eval(chunk.contiguous()) for chunk in lr_list]
        else:
            sr_list = [self.forward_chop(patch, shave=shave) for patch in lr_list]
        s = self.scale
        h, w, h_half, w_half, h_size, w_size = s * h, s * w, s * h_half, s * w_half, s * h_size, s * w_size
        output = sr_list[0].new_empty(b, sr_list[0].shape[1], h, w)
        output[:, :, 0:h_half, 0:w_half] = sr_list[0][:, :, 0:h_half, 0:w_half]
        output[:, :, 0:h_half, w_half:w] = sr_list[1][:, :, 0:h_half, (w_size - w + w_half):w_size]
        output[:, :, h_half:h, 0:w_half] = sr_list[2][:, :, (h_size - h + h_half):h_size, 0:w_half]
        output[:, :, h_half:h, w_half:w] = sr_list[3][:, :, (h_size - h + h_half):h_size, (w_size - w + w_half):w_size]
        return output

    def run_chopped_eval(self, x):
        return super().run_eval(x, keep_on_device=True)[0]

    def run_eval(self, x, y=None, request_loss=False, tag=None, timing=False, keep_on_device=False, *args, **kwargs):
        if self.max_combined_im_size is None:
            return super().run_eval(x, y, request_loss, tag, timing, keep_on_device, *args, **kwargs)
        dev = self._torch_device()
        if timing:
            torch.cuda.synchronize(dev)
            tic = time.perf_counter()
        sr_image = self.forward_chop(x.to(device=dev))
        if timing:
            torch.cuda.synchronize(dev)
            toc = time.perf_counter()
        loss = self.criterion(sr_image, y.to(device=dev)).detach().reshape(()).cpu().numpy() if request_loss and y is not None else None
        return (sr_image if keep_on_device else sr_image.cpu()), loss, (toc - tic) if timing else None


class EDSRHandler(_TiledEval):
    """EDSR on hand-written gfx950 kernels (reference EDSRHandler, handlers.py:8-25)."""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, hr_data_loc=None,
                 scheduler=None, scheduler_params=None, perceptual=None,
                 num_features=64, num_blocks=16, res_scale=0.1, max_combined_im_size=None, **kwargs):
        self.max_combined_im_size, self.scale = max_combined_im_size, scale
        super(EDSRHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode,
                                          hr_data_loc=hr_data_loc, **kwargs)
        self.net = EDSR(scale=scale, in_features=in_features, net_features=num_features, num_blocks=num_blocks,
                        res_scale=res_scale)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'edsr'


class RCANHandler(_TiledEval):
    """RCAN on hand-written gfx950 kernels (reference RCANHandler, handlers.py:28-42; extra kwargs reach RCAN(**kwargs))."""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, perceptual=None,
                 scheduler=None, scheduler_params=None, max_combined_im_size=None, **kwargs):
        self.max_combined_im_size, self.scale = max_combined_im_size, scale
        super(RCANHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = RCAN(scale=scale, in_feats=in_features, **kwargs)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'rcan'

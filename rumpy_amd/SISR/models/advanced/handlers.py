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
        """Each image axis of length L is covered by two windows of L // 2 + shave pixels, one anchored at either end; a window OWNS the
        output indices of its own half ([0, L // 2) or [L // 2, L)).  The four (row window, column window) quadrants are super-resolved on
        their own - recursively while a quadrant still has max_combined_im_size pixels or more - and each contributes the part it owns."""
        rows, cols = x.shape[-2:]
        s = self.scale

        def windows(L):             # (window start, window end, first owned index, one past the last owned index)
            half = L // 2
            return (0, half + shave, 0, half), (L - half - shave, L, half, L)

        recurse = (rows // 2 + shave) * (cols // 2 + shave) >= self.max_combined_im_size
        output = None
        for r0, r1, own_r0, own_r1 in windows(rows):
            for c0, c1, own_c0, own_c1 in windows(cols):
                quadrant = x[..., r0:r1, c0:c1]
                sr = self.forward_chop(quadrant, shave=shave) if recurse else self.run_chopped_eval(quadrant.contiguous())
                if output is None:
                    output = sr.new_empty(x.shape[0], sr.shape[1], s * rows, s * cols)
                output[..., s * own_r0:s * own_r1, s * own_c0:s * own_c1] = \
                    sr[..., s * (own_r0 - r0):s * (own_r1 - r0), s * (own_c0 - c0):s * (own_c1 - c0)]
        return output

    def run_chopped_eval(self, x):
        # every quadrant's status words are examined before it is stitched in (BaseModel.defer_eval_status is not honoured here): an fp16
        # overflow re-runs THAT quadrant in bf16 inside SREngine.forward, a strip-exchange time-out raises
        defer, self.defer_eval_status = self.defer_eval_status, False
        try:
            return super().run_eval(x, keep_on_device=True)[0]
        finally:
            self.defer_eval_status = defer

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
                 num_features=64, num_blocks=16, res_scale=0.1, max_combined_im_size=None, precision=None, **kwargs):
        # precision (not a reference kwarg): None = bf16 training arithmetic; 'fp8' = residual blocks on the block-scaled fp8 MFMA (opt-in)
        self.max_combined_im_size, self.scale = max_combined_im_size, scale
        super(EDSRHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode,
                                          hr_data_loc=hr_data_loc, **kwargs)
        self.net = EDSR(scale=scale, in_features=in_features, net_features=num_features, num_blocks=num_blocks,
                        res_scale=res_scale)
        self.net.set_precision(precision)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'edsr'


class RCANHandler(_TiledEval):
    """RCAN on hand-written gfx950 kernels (reference RCANHandler, handlers.py:28-42; extra kwargs reach RCAN(**kwargs))."""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, perceptual=None,
                 scheduler=None, scheduler_params=None, max_combined_im_size=None, precision=None, **kwargs):
        self.max_combined_im_size, self.scale = max_combined_im_size, scale
        super(RCANHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = RCAN(scale=scale, in_feats=in_features, **kwargs)
        self.net.set_precision(precision)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'rcan'

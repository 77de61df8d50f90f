"""Handlers of the "basic" models on the MI355X path.  ``define_model('srcnn' | 'vdsr', **kwargs)`` resolves to the two public classes
below (the registry keys come from the class names); constructor arguments, attributes and defaults follow the reference's
rumpy/SISR/models/basic/handlers.py:6-35 (Y-channel models on pre-interpolated inputs, MSE criterion, VDSR = 20 x 3x3 with gradient
clipping at 0.1), so TOML configurations written for the reference build these handlers unchanged.  BASELINE config 0 is SRCNN."""
from torch import nn

from rumpy_amd.shared_framework.models.base_architecture import BaseModel
from . import architectures as _arch


class _YChannelBase(BaseModel):
    """What SRCNN and VDSR share: a conv/ReLU chain on the luminance plane, trained with nn.MSELoss."""
    ARCH = None                 # network class
    NAME = None                 # `model_name`, also what a checkpoint records
    KERNELS = None              # default kernel_pattern (None: the architecture's own default)
    CHANNELS = None             # default channel_pattern
    CLIP = None                 # default grad_clip

    def _setup(self, device, model_save_dir, eval_mode, lr, kernel_pattern, channel_pattern, padding, scheduler, scheduler_params,
               perceptual, base_kwargs):
        BaseModel.__init__(self, device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **base_kwargs)
        self.net = self.ARCH(kernel_pattern=self.KERNELS if kernel_pattern is None else kernel_pattern,
                             channel_pattern=self.CHANNELS if channel_pattern is None else channel_pattern, padding=padding)
        self.colorspace, self.im_input = 'ycbcr', 'interp'       # the caller feeds Y of an image already at the HR size
        self.criterion = nn.MSELoss()
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = self.NAME


class SRCNNHandler(_YChannelBase):
    ARCH, NAME = _arch.SRCNN, 'srcnn'

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, kernel_pattern=None, channel_pattern=None, padding='same',
                 scheduler=None, scheduler_params=None, perceptual=None, **kwargs):
        self._setup(device, model_save_dir, eval_mode, lr, kernel_pattern, channel_pattern, padding, scheduler, scheduler_params,
                    perceptual, kwargs)


class VDSRHandler(_YChannelBase):
    ARCH, NAME = _arch.VDSR, 'vdsr'
    KERNELS = [3] * 20
    CHANNELS = [1] + [64] * 19 + [1]

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, kernel_pattern=None, channel_pattern=None, padding='same',
                 grad_clip=0.1, scheduler=None, scheduler_params=None, perceptual=None, **kwargs):
        self._setup(device, model_save_dir, eval_mode, lr, kernel_pattern, channel_pattern, padding, scheduler, scheduler_params,
                    perceptual, dict(kwargs, grad_clip=grad_clip))

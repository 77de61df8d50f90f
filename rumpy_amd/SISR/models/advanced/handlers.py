"""EDSR / RCAN handlers of the MI355X path - same class names, kwargs and attributes as
rumpy/SISR/models/advanced/handlers.py:8-42, so ``define_model('edsr' | 'rcan', **kwargs)`` resolves to them."""
from rumpy_amd.shared_framework.models.base_architecture import BaseModel
from .architectures import EDSR, RCAN


class EDSRHandler(BaseModel):
    """EDSR on hand-written gfx950 kernels (reference EDSRHandler, handlers.py:8-25)."""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, hr_data_loc=None,
                 scheduler=None, scheduler_params=None, perceptual=None,
                 num_features=64, num_blocks=16, res_scale=0.1, **kwargs):
        super(EDSRHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode,
                                          hr_data_loc=hr_data_loc, **kwargs)
        self.net = EDSR(scale=scale, in_features=in_features, net_features=num_features, num_blocks=num_blocks,
                        res_scale=res_scale)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'edsr'


class RCANHandler(BaseModel):
    """RCAN on hand-written gfx950 kernels (reference RCANHandler, handlers.py:28-42; extra kwargs reach RCAN(**kwargs))."""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, perceptual=None,
                 scheduler=None, scheduler_params=None, **kwargs):
        super(RCANHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.net = RCAN(scale=scale, in_feats=in_features, **kwargs)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'rcan'

"""QRCAN handler of the MI355X path - same class name, kwargs and attributes as
rumpy/SISR/models/attention_manipulators/handlers.py:11-79, so ``define_model('qrcan', **kwargs)`` resolves to it."""
from rumpy_amd.SISR.models.attention_manipulators import QModel
from .architectures import QRCAN


class QRCANHandler(QModel):
    """RCAN with meta-attention on hand-written gfx950 kernels.  ``style='standard'`` + ``include_q_layer=True`` is the
    configuration implemented; the reference's default style ('modulate', a gaussian re-scaling of a single QPI value) and
    the SRMD / SFT metadata planes are refused by the architecture."""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, scheduler=None,
                 scheduler_params=None, style='modulate', perceptual=None, clamp=False, min_mu=-0.2,
                 max_mu=0.8, n_feats=64, srmd_mode=False, **kwargs):
        super(QRCANHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        if srmd_mode:
            raise RuntimeError('rumpy_amd: srmd_mode (metadata concatenated with the input) is not on the HIP path')
        self.srmd_channel_mode = False
        self.net = QRCAN(scale=scale, in_feats=in_features, num_metadata=self.num_metadata, n_feats=n_feats, style=style, **kwargs)
        self.colorspace = 'augmented_rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'qrcan'
        self.min_mu, self.max_mu, self.clamp = min_mu, max_mu, clamp
        self.style = style

"""QRCAN handler of the MI355X path - same class name, kwargs and attributes as
rumpy/SISR/models/attention_manipulators/handlers.py:11-79, so ``define_model('qrcan', **kwargs)`` resolves to it."""
import numpy as np
import torch

from rumpy_amd.SISR.models.attention_manipulators import QModel
from .architectures import QRCAN


class QRCANHandler(QModel):
    """RCAN with meta-attention on hand-written gfx950 kernels: the reference's default style 'modulate' (one quality value per image,
    spread by ``scale_qpi`` into a gaussian bump over the 64 channels that multiplies every block's attention vector) and
    ``style='standard'`` with ``include_q_layer=True`` (q-layers on a metadata vector).  The other styles ('max_concat', 'mini_concat', 'extended_attention', 'softmax') run with the gate MLP as separate launches; the SRMD / SFT
    metadata planes are refused by the architecture."""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, scheduler=None,
                 scheduler_params=None, style='modulate', perceptual=None, clamp=False, min_mu=-0.2,
                 max_mu=0.8, n_feats=64, srmd_mode=False, precision=None, **kwargs):
        super(QRCANHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        if srmd_mode:
            raise RuntimeError('rumpy_amd: srmd_mode (metadata concatenated with the input) is not on the HIP path')
        self.srmd_channel_mode = False
        self.net = QRCAN(scale=scale, in_feats=in_features, num_metadata=self.num_metadata, n_feats=n_feats, style=style, **kwargs)
        self.net.set_precision(precision)      # (not a reference kwarg) None = bf16 ; 'fp8' = the one-launch RCAB kernels on the block-scaled fp8 MFMA, opt-in
        self.colorspace = 'augmented_rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)
        self.model_name = 'qrcan'
        self.min_mu, self.max_mu, self.clamp = min_mu, max_mu, clamp
        self.base_scaler = np.linspace(0, 1, n_feats)
        self.style = style

    @staticmethod
    def gaussian(x, mu, sig=0.2):
        """:59-63 (float64 numpy, cast to float32)"""
        return torch.from_numpy((1 / (np.sqrt(2 * np.pi) * sig)) * np.exp(-np.power(x - mu, 2.) / (2 * np.power(sig, 2.)))).type(torch.float32)

    def scale_qpi(self, qpi):
        """:65-73: [N,1,1,1] quality values in [0,1] -> [N,n_feats,1,1] attribute vectors (host side, as in the reference)"""
        scaled_qpi = (qpi.detach().cpu() * (self.max_mu - self.min_mu)) + self.min_mu
        full_scalers = torch.stack([self.gaussian(self.base_scaler, scaled_qpi[i].squeeze().numpy()) for i in range(scaled_qpi.size(0))])
        if self.clamp:
            full_scalers = torch.clamp(full_scalers, 0, 1)
        return full_scalers.unsqueeze(2).unsqueeze(3)

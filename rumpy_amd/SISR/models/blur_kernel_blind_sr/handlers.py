"""Blind QRCAN handler of the MI355X path - same class name, kwargs and attributes as ContrastiveBlindQRCANHandler,
rumpy/SISR/models/blur_kernel_blind_sr/handlers.py:454-609, so ``define_model('contrastiveblindqrcan', **kwargs)`` resolves to it."""
from rumpy_amd.shared_framework.models.base_architecture import BaseModel
from rumpy_amd.SISR.models.attention_manipulators.architectures import QRCAN
from .contrastive_blind_sr import ContrastiveBlindSRPipeline


class ContrastiveBlindQRCANHandler(BaseModel):
    """Frozen contrastive degradation encoder + QRCAN with meta-attention, SR (L1) loss only (``combined_loss_mode=None``).
    ``run_train`` / ``run_eval`` are BaseModel's, as in the reference for this mode (:524-525) - including the fact that the encoder's
    BatchNorms see ``net.train()`` during a training step (see encoding_models.py)."""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, include_sft_layer=False, srmd_mode=False,
                 scheduler=None, scheduler_params=None, style='modulate', perceptual=None, n_feats=64, encoder_type='default',
                 encoder_output_size=256, pre_trained_encoder_weights=None, auxiliary_encoder_weights=None, staggered_encoding=False,
                 embedding_type='pre-q', encoder_freeze_mode='all', encoder_train_eval='eval', combined_loss_mode=None, crop_count=None,
                 data_type='noise', reducer_layer_sizes=None, labelling_strategy='triple_precision', **kwargs):
        super(ContrastiveBlindQRCANHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        if crop_count is not None:
            raise RuntimeError('rumpy_amd: multi-crop batches (crop_count) belong to the contrastive losses, which are not on the HIP path')
        self.data_type, self.crop_count, self.encoder_train_eval = data_type, None, encoder_train_eval
        sr_net = QRCAN(scale=scale, in_feats=in_features, num_metadata=encoder_output_size, n_feats=n_feats, style=style,
                       include_sft_layer=include_sft_layer, staggered_encoding=staggered_encoding, **kwargs)
        kwargs['model_save_dir'] = model_save_dir
        self.net = ContrastiveBlindSRPipeline(device=device, eval_mode=eval_mode, generator=sr_net, encoder=encoder_type,
                                              pre_trained_encoder_weights=pre_trained_encoder_weights,
                                              auxiliary_encoder_weights=auxiliary_encoder_weights, embedding_type=embedding_type,
                                              encoder_freeze_mode=encoder_freeze_mode, combined_loss_mode=combined_loss_mode,
                                              staggered_encoding=staggered_encoding, sft_mode=include_sft_layer, srmd_mode=srmd_mode,
                                              crop_count=crop_count, reducer_layer_sizes=reducer_layer_sizes, **kwargs)
        self.model_name = 'blind_qrcan'
        self.encoder_type = encoder_type
        self.combined_loss_mode = combined_loss_mode
        self.colorspace = 'augmented_rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)

    def run_model(self, x, *args, **kwargs):
        return self.net.forward(x, **kwargs)

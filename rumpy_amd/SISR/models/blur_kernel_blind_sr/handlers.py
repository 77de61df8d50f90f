"""Blind QRCAN handler of the MI355X path - same class name, kwargs and attributes as ContrastiveBlindQRCANHandler,
rumpy/SISR/models/blur_kernel_blind_sr/handlers.py:454-609, so ``define_model('contrastiveblindqrcan', **kwargs)`` resolves to it."""
import torch

from rumpy_amd.regression.models.contrastive_learning import BaseContrastive
from rumpy_amd.SISR.models.attention_manipulators.architectures import QRCAN
from .contrastive_blind_sr import ContrastiveBlindSRPipeline


class ContrastiveBlindQRCANHandler(BaseContrastive):
    """Contrastive degradation encoder + QRCAN with meta-attention.
    ``combined_loss_mode=None``: SR (L1) loss only, the encoder frozen; ``run_train`` / ``run_eval`` are BaseModel's, as in the reference for
    this mode (:524-525) - including the fact that the encoder's BatchNorms see ``net.train()`` during a training step (see
    encoding_models.py).  ``'moco'`` / ``'supmoco'`` (:526-586): the encoder sits inside a MoCo / SupMoCo module and every step adds the
    cross-entropy of its logits to the L1 loss; see contrastive_blind_sr.py for the freeze modes this is built for."""

    def __init__(self, device, model_save_dir, eval_mode=False, lr=1e-4, scale=4, in_features=3, include_sft_layer=False, srmd_mode=False,
                 scheduler=None, scheduler_params=None, style='modulate', perceptual=None, n_feats=64, encoder_type='default',
                 encoder_output_size=256, pre_trained_encoder_weights=None, auxiliary_encoder_weights=None, staggered_encoding=False,
                 embedding_type='pre-q', encoder_freeze_mode='all', encoder_train_eval='eval', combined_loss_mode=None, crop_count=None,
                 data_type='noise', reducer_layer_sizes=None, labelling_strategy='triple_precision', precision=None, **kwargs):
        super(ContrastiveBlindQRCANHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode,
                                                           labelling_strategy=labelling_strategy, **kwargs)
        if crop_count is not None and combined_loss_mode is None:
            raise RuntimeError('rumpy_amd: multi-crop batches (crop_count) belong to the joint contrastive losses (combined_loss_mode)')
        if combined_loss_mode is not None and encoder_train_eval == 'eval' and not eval_mode:
            # the reference unpacks (embedding, logits, labels) from an encoder it has just put into eval mode, which returns the embedding alone
            raise RuntimeError('combined_loss_mode %r trains with encoder_train_eval="train" (an encoder in eval mode returns no logits, '
                               'handlers.py:516-528 / moco.py:132-187)' % (combined_loss_mode,))
        self.data_type, self.crop_count, self.encoder_train_eval = data_type, crop_count, encoder_train_eval
        sr_net = QRCAN(scale=scale, in_feats=in_features, num_metadata=encoder_output_size, n_feats=n_feats, style=style,
                       include_sft_layer=include_sft_layer, staggered_encoding=staggered_encoding, **kwargs)
        # (not a reference kwarg) None = bf16 ; 'fp8' = BASELINE config 5's "fp8 MFMA conv": the generator's RCAB kernels on the block-scaled fp8 MFMA
        sr_net.set_precision(precision)
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
        from rumpy_amd.regression.models.contrastive_learning.head import HipCrossEntropyLoss
        self.contrast_loss = HipCrossEntropyLoss()          # nn.CrossEntropyLoss() of the reference (:528), forward and backward in HIP
        self.colorspace = 'augmented_rgb'
        self.im_input = 'unmodified'
        self.activate_device()
        self.training_setup(lr, scheduler, scheduler_params, perceptual, device)

    def run_train(self, x, y, tag=None, mask=None, keep_on_device=False, *args, **kwargs):
        """:513-586.  Joint modes: x / y carry the crops of every image ([N, crops, 3, h, w], or already on the channel axis); the first crop is
        super-resolved, the others are its contrastive keys -> ({'train-loss', 'l1-loss', 'contrast-loss'}, logits on the CPU)."""
        if self.combined_loss_mode is None:
            return super().run_train(x, y, tag, mask, keep_on_device, *args, **kwargs)
        if self.eval_mode:
            raise RuntimeError('Model initialized in eval mode, training not possible.')
        self.net.train()
        dev = self._torch_device()
        if x.dim() == 5:                                   # :520-523: crops onto the channel axis
            x, y = x.flatten(1, 2), y.flatten(1, 2)
        x, y = x.to(device=dev), y.to(device=dev)
        if self.combined_loss_mode == 'moco':
            sr, output, target = self.net.forward(x[:, 0:3, ...], x[:, 3:, ...])
            y_sr = y[:, 0:3, ...]
        else:
            labels = self.class_logic(kwargs['metadata'], kwargs['metadata_keys'])
            self.net.E.set_class_count(self.total_classes)
            self.num_classes = self.net.E.num_classes
            x = x.reshape(-1, 3, x.shape[2], x.shape[3])
            y = y.reshape(-1, 3, y.shape[2], y.shape[3])
            first = torch.arange(0, x.shape[0], self.crop_count, device=dev)
            rest = torch.ones(x.shape[0], dtype=torch.bool, device=dev)
            rest[first] = False
            sr, output, target = self.net.forward(x[first], x[rest], labels.squeeze())
            y_sr = y[first]
        loss_contrast = self.contrast_loss(output, target.to(device=dev))
        loss_SR = self.criterion(sr, y_sr)
        loss = loss_contrast + loss_SR
        self.standard_update(loss)
        package = {name: v.detach().cpu().numpy() for v, name in zip((loss, loss_SR, loss_contrast), ('train-loss', 'l1-loss', 'contrast-loss'))}
        self._check_watchdog()          # (the read-back above has waited for the step)
        return package, output.detach().cpu()

    def run_model(self, x, *args, **kwargs):
        return self.net.forward(x, **kwargs)

    def set_multi_gpu(self, device_ids=None):
        """The generator's flat gradient buffer is averaged by BaseModel's GradientAverager; trainable encoder parameters (joint losses,
        ``encoder_freeze_mode`` other than "all") by one more coalesced all-reduce before the optimizer step.  The MoCo / SupMoCo queues
        stay identical on every rank (each enqueues the keys of all ranks, moco.py); the encoder's BatchNorm running statistics follow
        each rank's own shard, as under DistributedDataParallel without SyncBatchNorm (the reference's nn.DataParallel keeps replica 0's,
        i.e. rank 0's here - the rank that writes the checkpoints)."""
        from rumpy_amd.parallel import ParameterGradientAverager
        super().set_multi_gpu(device_ids)
        trainable = [p for p in self.net.E.parameters() if p.requires_grad]
        self.encoder_data_parallel = ParameterGradientAverager(trainable) if trainable else None

    def _apply_update(self, scheduler_skip=False):
        if getattr(self, 'encoder_data_parallel', None) is not None:
            self.encoder_data_parallel.average()
        super()._apply_update(scheduler_skip)

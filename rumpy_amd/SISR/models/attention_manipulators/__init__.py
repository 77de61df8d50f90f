"""QModel: the handler template of the metadata-modulated SR networks - mirror of
rumpy/SISR/models/attention_manipulators/__init__.py:11-201 for the vector-metadata (q-layer) configuration.

Same constructor arguments, ``generate_channels`` / ``channel_concat_logic`` semantics, ``run_train`` / ``run_eval`` signatures
and checkpoint key (``metadata_keys_used_in_training``).  Out of scope on the MI355X path, and refused loudly: the MoCo
contrastive encoder (``use_moco``), channel concatenation with the input (SRMD mode) and the tiled SFT metadata planes.
"""
import os

import torch

from rumpy_amd.shared_framework.models.base_architecture import BaseModel

_EXTRA_WIDTH = (('contrastive_encoding', 255), ('contrastive_q', 255), ('contrastive_encoding_tsne', 1), ('contrastive_q_tsne', 1),
                ('contrastive_encoding_pca', 10), ('contrastive_q_pca', 7), ('all', 39))


class QModel(BaseModel):
    def __init__(self, metadata=None, use_moco=None, pre_trained_encoder_weights=None, metadata_bypass_len=None,
                 ignore_degradation_location=False, **kwargs):
        if use_moco:
            raise RuntimeError('rumpy_amd: the MoCo metadata encoder is outside the MI355X hot path (no fallback)')
        self.style = None
        self.channel_concat = False
        self.no_metadata = False
        self.moco_encoding = False
        self.metadata_keys_used_in_training = None
        self.ignore_degradation_location = ignore_degradation_location
        if metadata_bypass_len:
            self.num_metadata, self.metadata = metadata_bypass_len, None
        elif metadata is not None:
            # vector width = one entry per named attribute + the fixed widths of the composite ones (:28-50)
            self.num_metadata = len(metadata) + sum(w for key, w in _EXTRA_WIDTH if key in metadata)
            if 'blur_kernel' in metadata:
                self.num_metadata += 9
            elif any('unmodified_blur_kernel' in m for m in metadata):
                self.num_metadata += 440
            self.metadata = [m[2:] if m[0].isdigit() else m for m in metadata] if ignore_degradation_location else metadata
        else:
            self.metadata, self.num_metadata = ['qpi'], 1
        super(QModel, self).__init__(**kwargs)

    def generate_channels(self, x, metadata, keys):
        """[N, *] metadata rows + their keys -> the [N, num_metadata, 1, 1] attribute vector the q-layers take (:84-103)."""
        if metadata is None:
            raise RuntimeError('Metadata needs to be specified for this network to run properly.')
        metadata = torch.as_tensor(metadata)
        if len(keys) == 1 or 'all' in self.metadata:
            picked = metadata.reshape(x.size(0), -1)
        else:
            mask = torch.tensor([key[0] in self.metadata for key in keys], dtype=torch.bool)
            picked = metadata.reshape(x.size(0), -1)[:, mask]
        extra_channels = (torch.ones(x.size(0), self.num_metadata) * picked.to(torch.float32).cpu()).unsqueeze(2).unsqueeze(3)
        if self.style == 'modulate':
            extra_channels = self.scale_qpi(extra_channels)
        return extra_channels

    def channel_concat_logic(self, x, extra_channels, metadata, metadata_keys):
        """:138-162 without the concatenating (SRMD) branch."""
        if self.channel_concat:
            raise RuntimeError('rumpy_amd: metadata concatenated with the input image (SRMD mode) is not on the HIP path')
        if self.no_metadata:
            return x, None
        if extra_channels is None:
            extra_channels = self.generate_channels(x, metadata, metadata_keys)
            dev = self._torch_device()
            if extra_channels.device != dev:
                extra_channels = extra_channels.to(dev)
        if self.metadata_keys_used_in_training is None and metadata_keys is not None:
            self.metadata_keys_used_in_training = [m[0] for m in metadata_keys]
        return x, extra_channels

    def save_model(self, model_save_name, extract_state_only=True, minimal=False):
        """:164-172: the base state plus the metadata keys seen in training; always written to file."""
        super().save_model(model_save_name=model_save_name, extract_state_only=extract_state_only, minimal=minimal)
        if self.metadata_keys_used_in_training:
            self.state['metadata_keys_used_in_training'] = self.metadata_keys_used_in_training
        torch.save(self.state, f=os.path.join(self.model_save_dir, '{}_{}'.format(model_save_name, self.curr_epoch)))

    def run_train(self, x, y, metadata=None, extra_channels=None, metadata_keys=None, *args, **kwargs):
        input_data, extra_channels = self.channel_concat_logic(x, extra_channels, metadata, metadata_keys)
        return super().run_train(input_data, y, extra_channels=extra_channels, **kwargs)

    def run_eval(self, x, y=None, request_loss=False, metadata=None, metadata_keys=None, extra_channels=None, *args, **kwargs):
        input_data, extra_channels = self.channel_concat_logic(x, extra_channels, metadata, metadata_keys)
        return super().run_eval(input_data, y, request_loss=request_loss, extra_channels=extra_channels, **kwargs)

    def run_model(self, x, extra_channels=None, *args, **kwargs):
        return self.net.forward(x, metadata=extra_channels)

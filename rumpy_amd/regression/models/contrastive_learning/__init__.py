"""Contrastive degradation-encoder training on the MI355X path: ``BaseContrastive`` and the class-labelling rules of
rumpy/regression/models/contrastive_learning/__init__.py:8-416.

A degraded image's class label is a mixed-radix number: every degradation family present in the training metadata contributes a few
digits (magnitude bucket, colour / type flags), digit j weighs the product of the radices before it, the label is the weighted sum
(``class_retrieval``, :216-288).  Host logic only - the arithmetic of this package is in encoding_models.py / moco.py / supmoco.py."""
import numpy as np
import torch

from rumpy_amd.shared_framework.models.base_architecture import BaseModel
from rumpy_amd.regression.models.contrastive_learning.encoding_models import Encoder

_STANDARD_KEYS = (('gaussian_noise', 'gaussian_noise_scale'), ('poisson_noise', 'poisson_noise_scale'), ('downsample', 'scale'),
                  ('gray_noise', 'gray_noise_boolean'), ('jpeg', 'jpeg_quality_factor'), ('qpi', 'jm_qpi'))


def partition_magnitude(magnitude, splits=2):
    """bucket of a [0, 1] magnitude: halves (> 0.5) or thirds (> 0.33, > 0.66) (:55-70)"""
    if splits == 2:
        return 1 if magnitude > 0.5 else 0
    if splits == 3:
        return 2 if magnitude > 0.66 else (1 if magnitude > 0.33 else 0)
    return None


def noise_logic(noise_class, noise_colour, magnitude, magnitude_split=2, split_noise_mag=True):
    """digits [magnitude bucket]? + [gray] + [gaussian] (:8-26)"""
    digits = [partition_magnitude(magnitude, magnitude_split)] if split_noise_mag else []
    return digits + [1 if noise_colour == 'gray' else 0, 1 if noise_class == 'gaussian' else 0]


def compression_logic(compression_class, magnitude, magnitude_split=2, class_split=False):
    """digits [magnitude bucket] + [JM (1) / JPEG (0)]? (:29-44)"""
    digits = [partition_magnitude(magnitude, magnitude_split)]
    if class_split:
        if 'jm' in compression_class:
            digits.append(1)
        elif 'jpeg' in compression_class:
            digits.append(0)
        else:
            raise RuntimeError('Unrecognized compression class.')
    return digits


def blur_logic(blur_class, sigma_x, sigma_y, magnitude_split=3):
    """digits [kernel type, sigma_x bucket, sigma_y bucket] (:47-53)"""
    return [int(blur_class), partition_magnitude(sigma_x, magnitude_split), partition_magnitude(sigma_y, magnitude_split)]


def register_metadata(keys):
    """raw metadata keys -> standard names, 'unknown' for the rest (:74-97)"""
    out = []
    for key in keys:
        for fragment, name in _STANDARD_KEYS:
            if fragment in key:
                out.append(name)
                break
        else:
            out.append(key.split('realesrganblur-')[-1] if 'realesrganblur' in key else 'unknown')
    return out


def partition_metadata(metadata_mapping, selected_metadata='all', labelling_strategy='default'):
    """-> (degradation families present, weight of every digit, number of classes) (:100-152)"""
    accepted = ['blur', 'compression', 'noise'] if selected_metadata == 'all' else selected_metadata
    families, radices = [], []
    if 'poisson_noise_scale' in metadata_mapping and 'noise' in accepted:
        families.append('noise')
        radices += {'default': [2, 2], 'double_precision': [2, 2, 2], 'triple_precision': [3, 2, 2]}.get(labelling_strategy, [])
    has_jpeg, has_jm = 'jpeg_quality_factor' in metadata_mapping, 'jm_qpi' in metadata_mapping
    if (has_jpeg or has_jm) and 'compression' in accepted:
        families.append('compression')
        radices += {'default': [2], 'double_precision': [2], 'triple_precision': [3]}.get(labelling_strategy, [])
        if has_jpeg and has_jm:
            radices.append(2)                       # which of the two codecs
            families.append('jm_jpg_compression')
    if 'kernel_type' in metadata_mapping and 'blur' in accepted:
        families.append('blur')
        radices += [7, 3, 3]
    weights = [1 if j == 0 else np.prod(radices[:j]) for j in range(len(radices))]
    return families, weights, np.prod(radices)


def degradation_vector_setup(available_classes):
    """two vector slots per degradation family (:155-165)"""
    return 2 * sum(1 for d in available_classes if d in ('noise', 'compression', 'blur'))


def vector_retrieval(metadata, valid_metadata, m_map):
    """degradation vector of one image: (gaussian | poisson) noise scale, (jpeg | jm) quality, blur sigmas (:168-200)"""
    vector = torch.zeros(degradation_vector_setup(valid_metadata))
    at = 0
    if 'noise' in valid_metadata:
        if metadata[m_map['gaussian_noise_scale']] > 0:
            vector[at] = metadata[m_map['gaussian_noise_scale']]
        else:
            vector[at + 1] = metadata[m_map['poisson_noise_scale']]
        at += 2
    if 'compression' in valid_metadata:
        if ('jpeg_quality_factor' in m_map and metadata[m_map['jpeg_quality_factor']] > 0) or 'jm_qpi' not in m_map:
            vector[at] = metadata[m_map['jpeg_quality_factor']]
        else:
            vector[at + 1] = metadata[m_map['jm_qpi']]
        at += 2
    if 'blur' in valid_metadata:
        vector[at] = metadata[m_map['sigma_x']]
        vector[at + 1] = metadata[m_map['sigma_y']]
    return vector


def class_retrieval(metadata, valid_metadata, m_map, decision_mags, total_classes, labelling_strategy='default'):
    """class label of one image's metadata row = sum_j digit_j * weight_j (:203-288)"""
    split = 3 if labelling_strategy == 'triple_precision' else 2
    split_noise = labelling_strategy in ('double_precision', 'triple_precision')
    digits = []
    if 'noise' in valid_metadata:
        gaussian = metadata[m_map['gaussian_noise_scale']] > 0
        mag = metadata[m_map['gaussian_noise_scale']] if gaussian else metadata[m_map['poisson_noise_scale']]
        colour = 'gray' if metadata[m_map['gray_noise_boolean']] > 0 else 'colour'
        digits += noise_logic('gaussian' if gaussian else 'poisson', colour, mag, magnitude_split=split, split_noise_mag=split_noise)
    if 'compression' in valid_metadata:
        if ('jpeg_quality_factor' in m_map and metadata[m_map['jpeg_quality_factor']] > 0) or 'jm_qpi' not in m_map:
            codec, mag = 'jpeg', metadata[m_map['jpeg_quality_factor']]
        else:
            codec, mag = 'jm', metadata[m_map['jm_qpi']]
        digits += compression_logic(codec, mag, magnitude_split=split, class_split='jm_jpg_compression' in valid_metadata)
    if 'blur' in valid_metadata:
        digits += blur_logic(metadata[m_map['kernel_type']], sigma_x=metadata[m_map['sigma_x']], sigma_y=metadata[m_map['sigma_y']])
    # the reference pairs weights and digits from the END (zip of the reversed lists, :281): only the common tail counts
    label = sum(int(d) * w for w, d in zip(reversed(decision_mags), reversed(digits)) if d != 0)
    if label >= total_classes:
        raise RuntimeError('Label is greater than the total number of possible classes.')
    return label


class BaseContrastive(BaseModel):
    """:291-416.  labelling_strategy: 'default' (no magnitude digits for noise), 'double_precision' (magnitudes in halves),
    'triple_precision' (thirds); override_queue: restart SupMoCo's queue even if a checkpoint brought one."""

    def __init__(self, device, use_noise_injection=False, noise_injection_frequency=0, noise_injection_sigma=0.1, labelling_strategy='default',
                 override_queue=False, **kwargs):
        super(BaseContrastive, self).__init__(device=device, **kwargs)
        self.colorspace = 'rgb'
        self.im_input = 'unmodified'
        if labelling_strategy == 'half_precision':         # older configurations
            labelling_strategy = 'double_precision'
        self.labelling_strategy = labelling_strategy
        self.use_noise_injection = use_noise_injection
        self.noise_injection_frequency = noise_injection_frequency
        self.noise_injection_sigma = noise_injection_sigma
        self.eval_request_loss = False                     # no loss on evaluation data
        self.training_metadata_mapping = {}
        self.valid_metadata = []
        self.decision_mags = []
        self.total_classes = 0
        self.regressor_type = 'contrastive'
        self.metadata_registered = False
        self.override_queue = override_queue
        self.degradation_vector_size = 0

    def register_training_metadata(self, metadata_keys):
        if not hasattr(self, 'data_type'):
            raise RuntimeError('Need to supply the degradation data types to analyze.')
        processed = register_metadata(metadata_keys)
        self.training_metadata_mapping = {key: processed.index(key) for key in processed}
        self.valid_metadata, self.decision_mags, self.total_classes = partition_metadata(self.training_metadata_mapping, self.data_type,
                                                                                         labelling_strategy=self.labelling_strategy)
        self.degradation_vector_size = degradation_vector_setup(self.valid_metadata)

    @staticmethod
    def define_encoder_model(model_name):
        """:333-351 - of the reference's choices (torchvision backbones, IDMN, the DASR encoder) only the last is on the HIP path"""
        if model_name == 'default':
            return Encoder
        raise RuntimeError('rumpy_amd: only the default (DASR) contrastive encoder is on the HIP path, not %r; there is no fallback' % (model_name,))

    def class_logic(self, metadata, keys):
        """labels [1, N] of a batch's metadata rows (:353-380); the first call fixes the class structure from the metadata keys"""
        if not self.metadata_registered:
            self.register_training_metadata([key[0] for key in keys])
            self.metadata_registered = True
            if self.__class__.__name__ == 'SupMoCoHandler':
                # a checkpoint may have brought a labelled queue: keep it unless asked not to / it does not fit the class count
                if not hasattr(self.net, 'queue_labels') or self.override_queue or int(max(self.net.queue_labels)) >= self.total_classes:
                    self.net.register_classes(self.total_classes)
                else:
                    self.net.set_class_count(self.total_classes)
        labels = torch.zeros((1, metadata.size()[0]))
        for index in range(labels.size()[1]):
            labels[0, index] = class_retrieval(metadata[index, :], self.valid_metadata, self.training_metadata_mapping, self.decision_mags,
                                               total_classes=self.total_classes, labelling_strategy=self.labelling_strategy)
        return labels.to(device=self._torch_device())

    def vector_logic(self, metadata, keys):
        """degradation vectors [size, N] (WeakCon's targets, :382-396; the model is weak_con.py)"""
        if not self.metadata_registered:
            self.register_training_metadata([key[0] for key in keys])
            self.metadata_registered = True
            if not hasattr(self.net, 'queue_vectors') or self.override_queue or self.degradation_vector_size != self.net.queue_vectors.size()[0]:
                self.net.register_vector(self.degradation_vector_size)
        vectors = torch.zeros((self.degradation_vector_size, metadata.size()[0]))
        for index in range(vectors.size()[1]):
            vectors[:, index] = vector_retrieval(metadata[index, :], self.valid_metadata, self.training_metadata_mapping)
        return vectors

    def get_embedding_len(self):
        test_im = torch.zeros((1, 3, 10, 10)).to(self._torch_device())
        self.net.eval()
        with torch.no_grad():
            return self.net.forward(test_im, test_im, get_q=True)[0].shape[1]

    def run_model(self, x, *args, **kwargs):
        return self.net.forward(x, x, **kwargs)

    def add_gaussian_noise_to_model(self, sigma=0.1):
        with torch.no_grad():
            for param in self.net.parameters():
                param.add_(torch.randn(param.size(), device=param.device) * sigma)
        if hasattr(self.net, 'mark_weights_updated'):
            self.net.mark_weights_updated()

    def epoch_end_calls(self):
        if self.use_noise_injection and self.curr_epoch % self.noise_injection_frequency == 0:
            self.add_gaussian_noise_to_model(self.noise_injection_sigma)

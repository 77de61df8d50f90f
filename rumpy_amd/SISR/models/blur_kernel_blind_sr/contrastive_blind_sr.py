"""Blind-SR pipeline (contrastive degradation encoder -> metadata-modulated SR network) on the MI355X path - mirror of
rumpy/SISR/models/blur_kernel_blind_sr/contrastive_blind_sr.py:14-329 for ``contrastive_encoder='default'``, ``embedding_type`` 'pre-q' / 'q':

* SR loss only (``combined_loss_mode=None``: the reference's own test and BASELINE config 5), the encoder frozen (``encoder_freeze_mode='all'``);
* the joint losses ``combined_loss_mode='moco' | 'supmoco'`` (:159-201,330-348): ``E`` is a MoCo / SupMoCo module (query + key encoder + queue,
  rumpy_amd/regression/models/contrastive_learning), a training forward returns (sr, logits, labels), and the handler adds the
  cross-entropy of the logits to the L1 loss.  ``encoder_freeze_mode`` as in the reference (:173-179): 'all' (its default: the contrastive
  loss is monitored, the key encoder and the queue keep moving, nothing of E is trained), 'pre_q' (only the ``mlp`` heads train - from
  the contrastive loss; the embedding is the pooled feature vector in front of them), anything else = the whole query encoder trains, from
  the contrastive loss AND from the SR loss through the generator's metadata input: the generator's autograd node then returns
  d loss / d metadata (rumpy_q_mlp_bwd_meta over its q-layers; q-layer networks only) and the encoder trunk's HIP backward pass takes it
  from there.

``ContrastiveBlindSRPipeline`` keeps the reference's sub-module names (``G`` then ``E``: they prefix every checkpoint key) and forward
semantics: embedding = E(x)[0] (pooled 256-vector) [-> optional min-max / mean-std normalisation] -> G(x, embedding[:, :, None, None]).
Refused loudly (not built): other encoders (DCLS, torchvision backbones), the 'q-dropdown' embedding, auxiliary encoders, the
reducer and the non-blind loss, SFT / SRMD metadata planes, contrastive_eval plotting."""
import torch
from torch import nn

from rumpy_amd.regression.models.contrastive_learning.encoding_models import Encoder
from rumpy_amd.regression.models.contrastive_learning.moco import MoCo
from rumpy_amd.regression.models.contrastive_learning.supmoco import SupMoCo


def load_encoder_model(weights, device, direct_load=False):
    """:14-31: the encoder's state dict out of a contrastive-training checkpoint (MoCo-style checkpoints hold it under ``encoder_q.``)."""
    if isinstance(device, int) or (isinstance(device, str) and device.isnumeric()):
        loc = 'cuda:%d' % int(device)          # the reference passes the GPU index
    else:
        loc = device                            # torch.device / 'cpu'
    state = torch.load(f=weights, map_location=loc, weights_only=False)
    if direct_load:
        return state
    encoder_dict = {}
    if state['model_name'] in ('mococontrastive', 'supmoco', 'weakcon'):
        for key, val in state['network'].items():
            if 'encoder_q' in key:
                encoder_dict[key[10:]] = val
    elif state['model_name'] == 'supcon':
        encoder_dict = state['network']
    return encoder_dict


def setup_encoder(contrastive_encoder, encoder_freeze_mode, pre_trained_encoder_weights, device, encoder_dropdown, load_required=False):
    """:34-61"""
    if contrastive_encoder != 'default':
        raise RuntimeError('rumpy_amd: only the default (DASR) contrastive encoder is on the HIP path, not %r' % (contrastive_encoder,))
    if encoder_freeze_mode != 'all':
        raise RuntimeError('rumpy_amd: the encoder is inference-only on the HIP path; encoder_freeze_mode must be "all"')
    E = Encoder(encoder_dropdown)
    for param in E.parameters():
        param.requires_grad = False
    if load_required:
        E.load_state_dict(state_dict=load_encoder_model(pre_trained_encoder_weights, device))
        print('Encoder weights loaded from %s' % pre_trained_encoder_weights)
    return E


class ContrastiveBlindSRPipeline(nn.Module):
    hip_pipeline = True

    def __init__(self, device, eval_mode, generator, contrastive_encoder='default', pre_trained_encoder_weights=None,
                 embedding_type='pre-q', encoder_freeze_mode='all', auxiliary_encoder_weights=None, staggered_encoding=False,
                 encoding_normalization_type=None, encoding_normalization_params=None, aux_encoding_normalization_params=None,
                 combined_loss_mode=None, crop_count=None, checkpoint_load=False, sft_mode=False, srmd_mode=False,
                 contrastive_eval=False, encoder_dropdown=None, contrastive_dropdown=False, reducer_layer_sizes=None,
                 block_encoder_loading=False, **kwargs):
        super(ContrastiveBlindSRPipeline, self).__init__()
        refused = [n for n, v in (('embedding_type=%r' % (embedding_type,), embedding_type not in ('pre-q', 'q')),
                                  ('auxiliary_encoder_weights', auxiliary_encoder_weights is not None), ('staggered_encoding', staggered_encoding),
                                  ('combined_loss_mode=%r' % (combined_loss_mode,), combined_loss_mode not in (None, 'moco', 'supmoco')), ('sft_mode', sft_mode),
                                  ('srmd_mode', srmd_mode), ('contrastive_eval', contrastive_eval), ('encoder_dropdown', encoder_dropdown is not None),
                                  ('reducer_layer_sizes', reducer_layer_sizes is not None)) if v]
        if refused:
            raise RuntimeError('rumpy_amd: blind-SR pipeline option(s) %s are not implemented on the HIP path; there is no fallback' % ', '.join(refused))
        if encoding_normalization_type not in (None, 'minmax', 'meanstd'):
            raise RuntimeError('Normalization type not recognized')
        if block_encoder_loading:       # :118-121: a testing switch of the reference - never read encoder weights from file
            checkpoint_load = True
        self.combined_loss_mode = combined_loss_mode
        self.eval_mode = eval_mode
        self.staggered_encoding = False
        self.aux_E = None
        self.reducer = None
        self.encoding_normalization_type = encoding_normalization_type
        self.encoding_normalization_params = encoding_normalization_params
        self.device = device
        self.model_save_dir = kwargs.get('model_save_dir')
        self.sft_mode = self.srmd_mode = False
        self.G = generator
        # :134-145: 'pre-q' = the pooled feature vector, 'q' = the mlp head's output on it (both 256 wide)
        self.embed_digit, self.q_type = (0, None) if embedding_type == 'pre-q' else (1, 'q')
        if combined_loss_mode is None:
            self.E = setup_encoder(contrastive_encoder, encoder_freeze_mode, pre_trained_encoder_weights, device, encoder_dropdown,
                                   load_required=not checkpoint_load)
        else:
            self.E = self._setup_contrastive_encoder(contrastive_encoder, encoder_freeze_mode, pre_trained_encoder_weights, device, crop_count,
                                                     contrastive_dropdown, load_required=not checkpoint_load)

    def _setup_contrastive_encoder(self, contrastive_encoder, encoder_freeze_mode, weights, device, crop_count, contrastive_dropdown, load_required):
        """:159-201: MoCo / SupMoCo around the default encoder, frozen per encoder_freeze_mode, state from a contrastive-training checkpoint"""
        if contrastive_encoder != 'default':
            raise RuntimeError('rumpy_amd: only the default (DASR) contrastive encoder is on the HIP path, not %r' % (contrastive_encoder,))
        if self.combined_loss_mode == 'moco':
            E = MoCo(base_encoder=Encoder, dropdown=None)
        else:
            if crop_count is None:
                raise RuntimeError('combined_loss_mode "supmoco" needs crop_count')
            E = SupMoCo(device=device, base_encoder=Encoder, contrastive_dropdown=contrastive_dropdown, positives_per_class=crop_count - 1, dropdown=None)
        for name, param in E.named_parameters():
            if encoder_freeze_mode == 'all' or (encoder_freeze_mode == 'pre_q' and 'mlp' not in name):
                param.requires_grad = False
        if load_required:
            loc = 'cuda:%d' % int(device) if isinstance(device, int) or (isinstance(device, str) and device.isnumeric()) else device
            state = torch.load(f=weights, map_location=loc, weights_only=False)
            for encoder_name in ('encoder_q', 'encoder_k'):
                getattr(E, encoder_name).load_state_dict({k[10:]: v for k, v in state['network'].items() if encoder_name in k})
            E.queue = state['network']['queue']
            if 'queue_labels' in state['network']:
                E.queue_labels = state['network']['queue_labels']
            E.queue_ptr = state['network']['queue_ptr']
            print('Encoder weights loaded from %s' % weights)
        return E

    # the trainable part, for the handler's fused optimizer / gradient all-reduce
    @property
    def hip_generator(self):
        return self.G

    @property
    def flat_g(self):
        return self.G.flat_g

    @property
    def flat_p(self):
        return self.G.flat_p

    def normalize(self, vectors, norm_params):
        """:229-239"""
        as_t = lambda v: torch.as_tensor(v, dtype=vectors.dtype, device=vectors.device)
        if self.encoding_normalization_type == 'minmax':
            return (vectors - as_t(norm_params['min'])) / (as_t(norm_params['max']) - as_t(norm_params['min']))
        return (vectors - as_t(norm_params['mean'])) / as_t(norm_params['std'])

    def embedding(self, x):
        """degradation representation of x as the generator's metadata [N, 256, 1, 1] (no gradient path into the encoder)"""
        enc = self.E if self.combined_loss_mode is None else self.E.encoder_q
        with torch.no_grad():
            emb = enc.features(x)                    # the pooled vector; BatchNorm mode = the encoder's own train / eval flag
            if self.q_type == 'q':
                emb = enc.mlp(emb)
        if self.encoding_normalization_type is not None:
            emb = self.normalize(emb, self.encoding_normalization_params)
        return emb.unsqueeze(2).unsqueeze(3)

    def forward(self, x, x_key=None, labels=None, **kwargs):
        if self.combined_loss_mode is None or not self.training:
            return self.G(x, self.embedding(x))
        # :330-338: one contrastive step of E (query = x, keys = x_key) whose embedding drives the generator
        if self.combined_loss_mode == 'moco':
            embedding, logits, labels = self.E(x, x_key)
        else:
            embedding, logits, labels, _ = self.E(x, x_key, labels)
        sr = self.G(x, embedding.unsqueeze(2).unsqueeze(3))       # a trainable trunk receives the SR loss's gradient through this input
        return sr, logits, labels

    # fused L1 train / eval steps of the generator, with the embedding as its metadata
    def fused_l1_forward_backward(self, x, y, metadata=None):
        return self.G.fused_l1_forward_backward(x, y, metadata=self.embedding(x))

    def l1_eval(self, x, y, metadata=None):
        return self.G.l1_eval(x, y, metadata=self.embedding(x))

    def take_early_loss(self):
        return self.G.take_early_loss()

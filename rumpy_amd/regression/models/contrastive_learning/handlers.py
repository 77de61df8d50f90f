"""Contrastive encoder-training handlers of the MI355X path - same class names, kwargs and return values as
rumpy/regression/models/contrastive_learning/handlers.py:12-163 (``MocoContrastiveHandler`` -> 'mococontrastive',
``SupMoCoHandler`` -> 'supmoco', ``WeakConHandler`` -> 'weakcon' (:166-216), ``SupConHandler`` -> 'supcon' (:219-257)), so
``define_model(name, **kwargs)`` resolves to them.  They train the degradation encoder the blind-SR handlers load as
``pre_trained_encoder_weights``.  Not built: the drop-down head and its direct regression loss, torchvision / IDMN backbones."""
import os

import torch
import torch.distributed as dist

from rumpy_amd.sr_tools.loss_functions import SupConLoss
from . import BaseContrastive
from .head import HipCrossEntropyLoss
from .moco import MoCo
from .supmoco import SupMoCo
from .weak_con import WeakCon


_CROP_INDEX = {}


def _split_crops(x, crop_count, device):
    """[N, crops, 3, H, W] (or anything that views to [N * crops, 3, H, W]) -> (first crop of every image, all the other crops) (:47-53).
    The two index vectors live on the device, one pair per (count, crops): selecting through them is the same launch at every step."""
    x = x.reshape(-1, 3, x.shape[-2], x.shape[-1]).to(device=device)
    key = (x.shape[0], crop_count, x.device)
    if key not in _CROP_INDEX:
        first = [i for i in range(0, x.shape[0], crop_count)]
        rest = [i for i in range(x.shape[0]) if i % crop_count]
        _CROP_INDEX[key] = (torch.tensor(first, dtype=torch.long, device=x.device), torch.tensor(rest, dtype=torch.long, device=x.device))
    first, rest = _CROP_INDEX[key]
    return x.index_select(0, first), x.index_select(0, rest)


def _graphed_step(handler, forward_backward, inputs, dev):
    """forward + loss + backward of a queue-based contrastive handler: forward_backward(*inputs, dev) -> (loss, returned tensor).

    The step's launch list is the same at every step - both trunks, the momentum update, the torch head and its autograd, the enqueue through
    a device-side slot vector - and the 32-crop step is host-bound (1.7 ms of Python for 1.15 ms of kernels): after two eager steps of a
    batch shape it is captured as ONE hipGraph and replayed (inputs copied into static buffers; the optimizer and the scheduler stay
    outside: their numbers change every step).  RUMPY_MOCO_STEP_GRAPH=0 or a process group (the key all-gather) keep it eager."""
    graphable = (os.environ.get('RUMPY_MOCO_STEP_GRAPH', '1') != '0' and all(t.is_cuda for t in inputs)
                 and not (dist.is_available() and dist.is_initialized()) and type(handler.optimizer).__name__ == 'FlatAdam')
    if not graphable:
        return forward_backward(*inputs, dev)
    net = handler.net          # a captured step holds the addresses of the parameter buffers, of the queue and of its side tracks: part of its key
    key = tuple((tuple(t.shape), t.dtype) for t in inputs) + (net.flat_p.data_ptr(), net.encoder_k.flat_p.data_ptr(), net.queue.data_ptr(),
                                                               net.queue_ptr.data_ptr()) + \
        tuple(getattr(net, n).data_ptr() for n in ('queue_labels', 'queue_vectors') if getattr(net, n, None) is not None)
    # ... and plan buffers of both trunks (z / a / dz / da, BatchNorm partials): when an encoder has dropped cached plans since, every captured
    # step may point into freed memory - all of them go (the next two calls of a shape run eagerly and rebuild what they need)
    ev = (net.encoder_q._evictions, net.encoder_k._evictions)
    if handler.__dict__.get('_step_graph_evictions') != ev:
        handler.__dict__['_step_graphs'] = {}
        handler.__dict__['_step_graph_evictions'] = ev
    st = handler.__dict__.setdefault('_step_graphs', {}).setdefault(key, {'calls': 0})
    st['calls'] += 1
    if st['calls'] <= 2:
        return forward_backward(*inputs, dev)
    if 'graph' not in st:
        st['in'] = [t.clone() for t in inputs]
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st['loss'], st['out'] = forward_backward(*st['in'], dev)
        st['graph'] = g
        st['n_keys'] = handler.net.__dict__['_slot_state'][0]
        # capture does not execute: what the host-side bookkeeping recorded during it is taken back, the replay below redoes it
        handler.net._moved(-st['n_keys'])
    handler.net._slots(st['n_keys'], handler.net._queue_pointer())       # (a checkpoint loaded since the last step moves the slot vector in place)
    for buf, t in zip(st['in'], inputs):
        buf.copy_(t)
    st['graph'].replay()
    handler.net._moved(st['n_keys'])
    handler.net.encoder_q._stats_epoch += 1
    handler.net.encoder_k._stats_epoch += 1
    return st['loss'], st['out']


class MocoContrastiveHandler(BaseContrastive):
    def __init__(self, device, model_save_dir, eval_mode=False, output_size=10, scheduler=None, scheduler_params=None, lr=1e-4,
                 model_name=None, crop_count=2, moco_t=0.07, **kwargs):
        super(MocoContrastiveHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.crop_count = crop_count
        self.net = MoCo(base_encoder=self.define_encoder_model(model_name), T=moco_t, positives=crop_count - 1)
        self.activate_device()
        self.criterion = HipCrossEntropyLoss()          # nn.CrossEntropyLoss() of the reference, forward and backward in HIP (head.py)
        self.training_setup(lr, scheduler, scheduler_params, device=device, perceptual=None)

    def run_model(self, x, *args, **kwargs):
        embedding, q = self.net.forward(x, x, get_q=True, **kwargs)
        return embedding, q

    def _forward_backward(self, x, dev):
        if self.crop_count == 2:
            im_q, im_k = x[:, 0:3, ...], x[:, 3:, ...]
        else:
            im_q, im_k = _split_crops(x, self.crop_count, dev)
        _, output, target = self.net(im_q=im_q, im_k=im_k)
        loss_contrast = self.criterion(output, target.to(device=dev))
        self.optimizer.zero_grad()
        loss_contrast.backward()
        return loss_contrast, output

    def run_train(self, x, y, tag=None, mask=None, *args, **kwargs):
        """x: [N, 6, H, W] (query crop | key crop on the channel axis) for crop_count 2, else [N, 3 * crops, H, W]: the first crop of an image
        is its query, the others its keys -> (contrastive loss, logits [N, 1 + K] on the CPU) (:37-63)"""
        if self.eval_mode:
            raise RuntimeError('Model initialized in eval mode, training not possible.')
        self.net.train()
        dev = self._torch_device()
        loss_contrast, output = _graphed_step(self, self._forward_backward, (x.to(device=dev),), dev)
        self._apply_update()
        return loss_contrast.detach().cpu().numpy(), output.detach().cpu()


class SupMoCoHandler(BaseContrastive):
    def __init__(self, device, model_save_dir, eval_mode=False, output_size=10, scheduler=None, scheduler_params=None, lr=1e-4,
                 model_name='default', crop_count=2, moco_t=0.07, data_type='noise', dropdown=None, dropdown_metadata_target=None,
                 include_direct_loss=False, direct_loss_only=False, contrastive_dropdown=True, **kwargs):
        super(SupMoCoHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        if dropdown is not None or include_direct_loss or direct_loss_only:
            raise RuntimeError('rumpy_amd: the encoder drop-down head and its direct regression loss are not on the HIP path; there is no fallback')
        self.crop_count = crop_count
        self.temperature = moco_t
        self.data_type = data_type
        self.net = SupMoCo(base_encoder=self.define_encoder_model(model_name), positives_per_class=crop_count - 1, dim=256,
                           contrastive_dropdown=contrastive_dropdown, T=moco_t, device=device, dropdown=None)
        self.activate_device()
        self.criterion = HipCrossEntropyLoss()          # nn.CrossEntropyLoss() of the reference, forward and backward in HIP (head.py)
        self.training_setup(lr, scheduler, scheduler_params, device=device, perceptual=None)
        self.include_direct_loss = False
        self.dropdown = None
        self.dropdown_metadata_target = dropdown_metadata_target
        self.contrastive_dropdown = contrastive_dropdown
        self.direct_loss_only = False
        self.target_loss = torch.nn.L1Loss()

    def run_train(self, x, y, tag=None, mask=None, *args, **kwargs):
        """x: [N, crops, 3, H, W]; y: the batch's degradation metadata [N, M] with kwargs['metadata_keys'] naming its columns
        -> (contrastive loss, embedding [N, 256] on the CPU) (:115-158)"""
        if self.eval_mode:
            raise RuntimeError('Model initialized in eval mode, training not possible.')
        self.net.train()
        dev = self._torch_device()
        labels = self.class_logic(y, kwargs['metadata_keys']).reshape(-1).to(device=dev, dtype=torch.int64)      # host logic, per batch
        loss_contrast, embedding = _graphed_step(self, self._forward_backward, (x.to(device=dev), labels), dev)
        self._apply_update()
        return loss_contrast.detach().cpu().numpy(), embedding.detach().cpu()

    def _forward_backward(self, x, labels, dev):
        im_q, im_k = _split_crops(x, self.crop_count, dev)
        embedding, logits, full_labels, _ = self.net(im_q, im_k, labels)
        loss_contrast = self.criterion(logits, full_labels.to(dev))
        self.optimizer.zero_grad()
        loss_contrast.backward()
        return loss_contrast, embedding

    def run_model(self, x, *args, **kwargs):
        embedding, q = self.net.forward(x, x, get_q=True, **kwargs)
        return embedding, q


class WeakConHandler(BaseContrastive):
    def __init__(self, device, model_save_dir, eval_mode=False, output_size=10, scheduler=None, scheduler_params=None, lr=1e-4,
                 model_name='default', crop_count=2, moco_t=0.07, data_type='noise', **kwargs):
        super(WeakConHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.crop_count = crop_count
        self.temperature = moco_t
        self.data_type = data_type
        self.net = WeakCon(base_encoder=self.define_encoder_model(model_name), positives_per_class=crop_count - 1, T=moco_t, device=device)
        self.activate_device()
        self.criterion = HipCrossEntropyLoss()          # nn.CrossEntropyLoss() of the reference, forward and backward in HIP (head.py)
        self.training_setup(lr, scheduler, scheduler_params, device=device, perceptual=None)

    def run_train(self, x, y, tag=None, mask=None, *args, **kwargs):
        """as SupMoCoHandler.run_train, with the batch's degradation VECTORS (vector_logic) in place of class labels (:190-211)"""
        if self.eval_mode:
            raise RuntimeError('Model initialized in eval mode, training not possible.')
        self.net.train()
        dev = self._torch_device()
        vectors = self.vector_logic(y, kwargs['metadata_keys']).to(device=dev)                                   # host logic, per batch
        loss_contrast, embedding = _graphed_step(self, self._forward_backward, (x.to(device=dev), vectors.contiguous()), dev)
        self._apply_update()
        return loss_contrast.detach().cpu().numpy(), embedding.detach().cpu()

    def _forward_backward(self, x, vectors, dev):
        im_q, im_k = _split_crops(x, self.crop_count, dev)
        embedding, logits, full_labels = self.net(im_q, im_k, vectors.squeeze())
        loss_contrast = self.criterion(logits, full_labels.to(dev))
        self.optimizer.zero_grad()
        loss_contrast.backward()
        return loss_contrast, embedding

    def run_model(self, x, *args, **kwargs):
        embedding, q = self.net.forward(x, x, get_q=True, **kwargs)
        return embedding, q


class SupConHandler(BaseContrastive):
    """One encoder, no queue: the supervised contrastive loss over the batch's crops, views = crops of an image (:219-257).
    Two statements of the reference's version cannot run with any encoder its code base defines and are repaired here, stated: it indexes
    the encoder's output dict as a tensor (``q.view``, :248 - here ``q['q']``), and it never sets the ``data_type`` its class_logic asks
    for (here a ``data_type`` argument like the sibling handlers', default 'noise').  The loss itself is the reference's (G20)."""

    def __init__(self, device, model_save_dir, eval_mode=False, output_size=10, scheduler=None, model_name='default', scheduler_params=None,
                 lr=1e-4, crop_count=2, data_type='noise', **kwargs):
        super(SupConHandler, self).__init__(device=device, model_save_dir=model_save_dir, eval_mode=eval_mode, **kwargs)
        self.data_type = data_type
        self.net = self.define_encoder_model(model_name)()
        self.activate_device()
        self.net.flatten()                       # flat parameter / gradient buffers: the fused Adam launch (FlatAdam) walks them
        self.criterion = SupConLoss()
        self.training_setup(lr, scheduler, scheduler_params, device=device, perceptual=None)
        self.crop_count = crop_count

    def run_train(self, x, y, tag=None, mask=None, *args, **kwargs):
        if self.eval_mode:
            raise RuntimeError('Model initialized in eval mode, training not possible.')
        self.net.train()
        x = x.view(-1, 3, x.size()[-2], x.size()[-1]).to(device=self._torch_device())
        embedding, q = self.net(x)
        labels = self.class_logic(y, kwargs['metadata_keys'])
        loss_contrast = self.criterion(q['q'].view(-1, self.crop_count, q['q'].size()[1]), labels)
        self.standard_update(loss_contrast)
        return loss_contrast.detach().cpu().numpy(), embedding.detach().cpu()

    def run_model(self, x, *args, **kwargs):
        embedding, q = self.net.forward(x)
        return embedding, q

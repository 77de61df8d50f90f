"""MoCo over the HIP degradation encoder - mirror of rumpy/regression/models/contrastive_learning/moco.py:8-187 (``MoCo``: query encoder,
momentum key encoder, queue of negative keys; the reference's two changes to the original are kept: several positive keys per query,
and the DASR encoder as backbone).

What runs where: both encoders' convolutional trunks are HIP (encoding_models.py: forward, and for the query encoder the whole backward
pass); the key encoder's momentum update is one launch over the two flat parameter buffers (rumpy_ema); Adam is the fused launch over the
query encoder's flat buffers (rumpy_amd.optim.FlatAdam through the ``flat_protocol`` attributes below).  The contrastive head - two
256 x 256 linear layers, L2 normalisation, the [N, 256] x [256, K] logits GEMM, cross-entropy - is plain torch on [N, 256] / [N, 1 + K]
matrices (rocBLAS GEMMs), with torch autograd delivering d loss / d fea to the trunk's autograd node.

Data parallel (one process per GPU): every rank enqueues the keys of ALL ranks, in rank order (the step the reference marks at
moco.py:78 ``# keys = concat_all_gather(keys)``), so the replicas' queues stay identical; gradients are averaged by the handler."""
import torch
import torch.distributed as dist
import torch.nn as nn

from rumpy_amd import _lib as L
from .encoding_models import Encoder
from .head import moco_logits, normalize_rows


class MoCo(nn.Module):
    flat_protocol = True             # BaseModel: param_list / offsets / flat_p / flat_g / attach_grads / engine of the trainable part
    supports_fused_l1 = False

    def __init__(self, base_encoder, dim=256, K=32 * 256, m=0.999, T=0.07, mlp=True, positives=1, dropdown=None):
        """dim: feature dimension; K: queue size (negative keys); m: momentum of the key encoder; T: softmax temperature (:18-24)"""
        super(MoCo, self).__init__()
        if base_encoder is not Encoder:
            raise RuntimeError('rumpy_amd: only the default (DASR) encoder is on the HIP path, not %r' % (getattr(base_encoder, '__name__', base_encoder),))
        self.K, self.m, self.T = K, m, T
        self.vector_dim = dim
        self.dropdown = dropdown
        self.positives = positives
        self.encoder_q = base_encoder(dropdown)
        self.encoder_k = base_encoder(dropdown)
        for param_q, param_k in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            param_k.data.copy_(param_q.data)
            param_k.requires_grad = False              # the key encoder follows by momentum, not by gradient (:52-54)
        self.register_buffer('queue', nn.functional.normalize(torch.randn(dim, K), dim=0))
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        self.use_graph = False
        self._ptr_seen = None            # (version counter of queue_ptr, its value): the pointer is read back from the GPU only when it
                                         # was written from outside (load_state_dict, register_classes) - int(tensor) is a full device sync

    # ------------------------------------------------------------------ flat protocol (the trainable part = the query encoder)
    def _flat(self):
        q, k = self.encoder_q, self.encoder_k
        for e in (q, k):
            if e.flat_p is None or e._convs()[0][0].weight.data_ptr() != e.flat_p.data_ptr():
                e.flatten()
        return q

    def train(self, mode=True):
        """the handlers call net.train() at every step: the walk over ~60 sub-modules is skipped when nothing changes"""
        if self.training == mode and self.encoder_q.training == mode and self.encoder_k.training == mode:
            return self
        return super().train(mode)

    param_list = property(lambda self: self._flat().param_list)
    offsets = property(lambda self: self._flat().offsets)
    flat_p = property(lambda self: self._flat().flat_p)
    flat_g = property(lambda self: self._flat().flat_g)

    def attach_grads(self):
        self._flat().attach_grads()

    def _ensure_engine(self):
        if not self.flat_p.is_cuda:
            raise RuntimeError('rumpy_amd: this network only runs on an MI355X through the HIP extension; '
                               'there is no CPU path (parameters are on %s)' % self.flat_p.device)

    @property
    def engine(self):
        return self.encoder_q.engine          # FlatAdam's repack hook: the query encoder's filter images are stale after an update

    def mark_weights_clean(self):
        pass

    def mark_weights_updated(self):
        self.encoder_q.weights_rewritten()
        self.encoder_k.weights_rewritten()

    # ------------------------------------------------------------------ MoCo
    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """param_k = param_k * m + param_q * (1 - m) for every parameter (:66-72), one launch over the flat buffers"""
        q = self._flat()
        k = self.encoder_k
        if not k.flat_p.is_cuda:
            raise RuntimeError('rumpy_amd: MoCo trains on the GPU only (no CPU fallback)')
        L.check(L.lib().rumpy_ema(k.flat_p.data_ptr(), q.flat_p.data_ptr(), k.flat_p.numel(), float(self.m), float(1. - self.m),
                                  torch.cuda.current_stream(k.flat_p.device).cuda_stream), 'rumpy_ema')
        k.weights_rewritten()

    @torch.no_grad()
    def _gathered(self, t):
        """rank-ordered concatenation over the data-parallel group (identity on one process)"""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
            dist.all_gather(parts, t.contiguous())
            return torch.cat(parts, dim=0)
        return t

    def _zero_labels(self, n, dev):
        z = self.__dict__.setdefault('_zeros', {}).get((n, dev))
        if z is None:
            z = self.__dict__['_zeros'][(n, dev)] = torch.zeros(n, dtype=torch.long, device=dev)
        return z

    @torch.no_grad()
    def _keys_of_all_ranks(self, keys, stride=1, labels=None):
        """data parallel: every rank enqueues the keys (and labels) of ALL ranks in rank order (the step the reference marks at moco.py:78), so
        the replicas' queues stay identical -> (keys, stride, labels) as the enqueue takes them; identity on one process"""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            keys, stride = self._gathered(keys[::stride].contiguous()), 1
            labels = None if labels is None else self._gathered(labels)
        return keys, stride, labels

    def _queue_pointer(self):
        seen = self._ptr_seen
        if seen is not None and seen[0] == (id(self.queue_ptr), self.queue_ptr._version):
            return seen[1]
        return int(self.queue_ptr)

    def _advance_queue_pointer(self, ptr):
        self.queue_ptr[0] = ptr
        self._ptr_seen = ((id(self.queue_ptr), self.queue_ptr._version), ptr)

    def _slots(self, batch_size, ptr):
        """device vector of the queue columns the next `batch_size` keys go to, advanced ON THE DEVICE by every enqueue - so that the enqueue
        is the same launch list at every step (a captured step graph replays it) while the host keeps its own copy of the pointer.
        ONE vector per (key count, device), created once and never replaced: a captured step holds its address, so a step with another
        batch size in between (a last partial batch) must not free it - its content is brought up to the host's pointer IN PLACE when the
        shape comes back."""
        table = self.__dict__.setdefault('_slot_table', {})
        k = (batch_size, self.queue.device)
        st = table.get(k)
        if st is None:
            st = table[k] = [batch_size, ptr, (torch.arange(batch_size, device=self.queue.device, dtype=torch.long) + ptr) % self.K]
        elif st[1] != ptr:          # moved from outside: a loaded checkpoint, or steps of another batch size since this vector's last use
            st[2].copy_((torch.arange(batch_size, device=self.queue.device, dtype=torch.long) + ptr) % self.K)
            st[1] = ptr
        self.__dict__['_slot_state'] = st
        return st

    def _moved(self, batch_size):
        """host bookkeeping of one enqueue (also what a step-graph replay calls: the device side moved inside the graph)"""
        st = self.__dict__['_slot_state']
        st[1] = (st[1] + batch_size) % self.K
        self._ptr_seen = ((id(self.queue_ptr), self.queue_ptr._version), st[1])

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, stride=1, labels=None):
        """:74-89 (supmoco.py:34-50 with the label track): every `stride`-th row of `keys` joins the queue at the device-side slot vector, which
        advances with the queue pointer - one launch (rumpy_moco_enqueue)"""
        keys, stride, labels = self._keys_of_all_ranks(keys, stride, labels)
        if not keys.is_cuda:
            raise RuntimeError('rumpy_amd: the MoCo queue lives on the GPU (rumpy_moco_enqueue); there is no CPU path')
        keys = keys.contiguous()
        batch_size = keys.shape[0] // stride
        assert self.K % batch_size == 0  # for simplicity
        st = self._slots(batch_size, self._queue_pointer())
        L.check(L.lib().rumpy_moco_enqueue(self.queue.data_ptr(), keys.data_ptr(), st[2].data_ptr(), self.queue_ptr.data_ptr(),
                                           None if labels is None else self.queue_labels.data_ptr(), None if labels is None else labels.contiguous().data_ptr(),
                                           batch_size, stride, keys.shape[1], self.K, torch.cuda.current_stream(keys.device).cuda_stream), 'rumpy_moco_enqueue')
        # (queue_ptr is advanced by the kernel, behind torch's version counter: the host copy of the pointer is what _moved keeps)
        self._moved(batch_size)

    def forward(self, im_q, im_k, **kwargs):
        """training: (embedding, logits [N, 1 + K], labels (zeros)) ; evaluation: embedding, or (embedding, q) with get_q (:132-187)"""
        if not self.training:
            embedding, q = self.encoder_q(im_q)
            if kwargs.get('get_q'):
                return embedding, q['q']
            return embedding
        n = im_q.shape[0]
        embedding, heads = self.encoder_q(im_q)
        with torch.no_grad():                                                   # the keys carry no gradient
            self._momentum_update_key_encoder()
            k = normalize_rows(self.encoder_k(im_k)[1]['q'])                    # keys     [N * positives, C], L2-normalised
        # queries: q = normalize(mlp(fea)).  Column 0: the positive logit = mean over the query's own key crops of q . k ; columns 1..K: q
        # against the queue ; all / T (one positive: q . k / T, :150-151,170-172; several: (sum_p q . k_p / T) / positives, :153-158,174-177):
        # normalisation, both products and their backward pass are HIP (head.py)
        logits = moco_logits(heads['q'], k, self.queue, self.T, self.positives)
        labels = self._zero_labels(n, logits.device)                            # cross-entropy target: column 0
        self._dequeue_and_enqueue(k, stride=self.positives)                     # one key per query joins the queue (:181-184)
        return embedding, logits, labels

"""SupMoCo over the HIP degradation encoder - mirror of rumpy/regression/models/contrastive_learning/supmoco.py:7-138 (``SupMoCo``,
https://arxiv.org/abs/2101.11058): MoCo whose positives are, besides the query's own key crops, every queue entry carrying the query's class
label.  The encoders, the momentum update and the optimizer are MoCo's (moco.py); the logits - q . k products, the one-hot label
matrices [N, classes + 1] x [classes + 1, K], the [N, K] x [K, 256] positive-feature sum and the [N, 256] x [256, K] negatives - are
plain torch GEMMs with torch autograd, as in the reference."""
import torch
import torch.nn as nn

from .head import moco_logits, normalize_rows
from .moco import MoCo


class SupMoCo(MoCo):
    def __init__(self, device, positives_per_class=4, contrastive_dropdown=True, **kwargs):
        super(SupMoCo, self).__init__(**kwargs)
        self.num_classes = 0
        self.positives_per_class = positives_per_class
        self.device = device
        self.contrastive_dropdown = contrastive_dropdown

    def register_classes(self, num_classes):
        """:28-32: (re)start the queue's label track - every slot gets the "no class" label ``num_classes`` - and the queue pointer"""
        self.set_class_count(num_classes)
        dev = self.queue.device
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long, device=dev))
        self.register_buffer('queue_labels', (torch.ones(self.K, device=dev) * num_classes).to(torch.int64))

    def set_class_count(self, num_classes):
        self.num_classes = num_classes

    def forward(self, im_q, im_k, labels=None, **kwargs):
        """training: (embedding, logits [N, 1 + K], zeros, encoder outputs) ; evaluation as MoCo (:52-138)"""
        if not self.training:
            embedding, q_out = self.encoder_q(im_q)
            if kwargs.get('get_q'):
                return embedding, (q_out if self.dropdown else q_out['q'])
            return embedding
        if self.num_classes == 0:
            raise RuntimeError('Maximum number of classes must be registered before running a training step.')
        if labels is None:
            raise RuntimeError('Labels required for a training step.')
        n, P = im_q.shape[0], self.positives_per_class
        embedding, heads = self.encoder_q(im_q)
        with torch.no_grad():
            self._momentum_update_key_encoder()
            k = normalize_rows(self.encoder_k(im_k)[1]['q'])                    # [N * P, C]
        labels = labels.to(device=self.queue.device, dtype=torch.int64).reshape(-1).contiguous()
        # q = normalize(mlp(fea)).  Queue entry j is a positive of query n when their class labels agree (:93-97 builds that 0/1 matrix as a
        # product of one-hot matrices; here a label comparison, rumpy_label_match); free slots carry the label `num_classes`, which no query has.
        # l_pos = (sum_p q . k_p + q . sum of its positive queue features) / T / (P + their number) (:99-112), l_neg = q @ queue / T: HIP, head.py
        logits = moco_logits(heads['q'], k, self.queue, self.T, P, labels=labels, queue_labels=self.queue_labels)
        full_labels = self._zero_labels(n, logits.device)
        self._dequeue_and_enqueue(k, stride=P, labels=labels)                               # one key per query, with the query's label
        return embedding, logits, full_labels, heads

"""WeakCon over the HIP degradation encoder - mirror of rumpy/regression/models/contrastive_learning/weak_con.py:7-113 (weak contrastive
learning, https://doi.org/10.1016/j.knosys.2022.108984): MoCo whose negative logits are weighted by the distance between the query's
degradation vector and the queue entries' - a negative with a similar degradation pushes less.  Encoders, momentum update, queue and
optimizer are MoCo's (moco.py); the head is plain torch, as there."""
import torch
import torch.nn as nn

from .supmoco import SupMoCo


class WeakCon(SupMoCo):
    def __init__(self, **kwargs):
        super(WeakCon, self).__init__(**kwargs)
        self.weight_error = nn.MSELoss(reduction='sum')

    def register_vector(self, vector_size):
        """:17-19: (re)start the queue's degradation-vector track and the queue pointer"""
        dev = self.queue.device
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long, device=dev))
        self.register_buffer('queue_vectors', torch.zeros(vector_size, self.K, dtype=torch.float32, device=dev))

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys, vectors):
        keys, vectors = self._gathered(keys), self._gathered(vectors.t().contiguous()).t()
        batch_size = keys.shape[0]
        assert self.K % batch_size == 0  # for simplicity
        st = self._slots(batch_size, self._queue_pointer())          # device-side slot vector: the same launches at every step (moco.py)
        self.queue.index_copy_(1, st[2], keys.transpose(0, 1))
        self.queue_vectors.index_copy_(1, st[2], vectors)
        st[2].add_(batch_size).remainder_(self.K)
        self.queue_ptr.add_(batch_size).remainder_(self.K)
        self._moved(batch_size)

    def forward(self, im_q, im_k, q_vector=None, **kwargs):
        """training: (embedding, logits [N, 1 + K], zeros) with q_vector [V, N] the queries' degradation vectors ; evaluation as MoCo (:36-113)"""
        if not self.training:
            embedding, q = self.encoder_q(im_q)
            if kwargs.get('get_q'):
                return embedding, q['q']
            return embedding
        if q_vector is None:
            raise RuntimeError('Vector labels required for a training step.')
        n, P = im_q.shape[0], self.positives_per_class
        q_vector = q_vector.to(device=self.queue.device, dtype=torch.float32).reshape(-1, n)
        embedding, heads = self.encoder_q(im_q)
        q = nn.functional.normalize(heads['q'], dim=1)
        with torch.no_grad():
            self._momentum_update_key_encoder()
            k = nn.functional.normalize(self.encoder_k(im_k)[1]['q'], dim=1)
        l_pos = torch.einsum('nc,npc->np', q, k.view(n, P, self.vector_dim)).sum(dim=1) / self.T / P       # mean positive logit (:64-70)
        weights = torch.cdist(q_vector.t(), self.queue_vectors.t())                                       # [N, K] vector distances (:90)
        l_neg = (q @ self.queue.detach().clone()) * weights / self.T
        logits = torch.cat([l_pos.unsqueeze(1), l_neg], dim=1)
        full_labels = torch.zeros(n, dtype=torch.long, device=logits.device)
        self._dequeue_and_enqueue(k[::P], q_vector)
        return embedding, logits, full_labels

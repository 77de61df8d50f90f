"""CPU oracle for the contrastive training of the degradation encoder (MoCo / SupMoCo steps) - SURVEY.md 8f.4.

TEST INFRASTRUCTURE ONLY, like oracle/sr_oracle.py: nothing under ``rumpy_amd/`` may import this module.

A plain PyTorch-CPU fp32 restatement of what the reference computes in one ``run_train`` of its contrastive handlers
(rumpy/regression/models/contrastive_learning/handlers.py:12-163) on the DASR encoder (``OracleEncoder`` of sr_oracle.py):
query / key encoders, momentum update, logits, cross-entropy, autograd, Adam.  The class-label rules
(``oracle_class_label``) restate rumpy/regression/models/contrastive_learning/__init__.py:8-288.

Parity pinning: PINNED by tests/golden/g20_contrastive_train.npz, produced by tests/golden/make_golden_contrastive.py from the REAL
reference handlers (define_model('mococontrastive' | 'supmoco')) in the build container; tests/test_oracle_golden.py checks this file
against it (losses, logits, gradients, queue contents, key-encoder weights, labels)."""
import copy

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .sr_oracle import OracleEncoder


def seeded_queue(dim, K, seed):
    """unit-norm random queue columns from numpy's bit-stream (the reference draws torch.randn at construction, moco.py:57-58)"""
    q = torch.from_numpy(np.random.default_rng(seed).standard_normal((dim, K)).astype(np.float32))
    return F.normalize(q, dim=0)


class OracleMoCo(nn.Module):
    """moco.py:8-187.  positives: key crops per query; K: queue length; m: key-encoder momentum; T: temperature."""

    def __init__(self, dim=256, K=32 * 256, m=0.999, T=0.07, positives=1):
        super().__init__()
        self.K, self.m, self.T, self.dim, self.positives = K, m, T, dim, positives
        self.encoder_q = OracleEncoder()
        self.encoder_k = copy.deepcopy(self.encoder_q)                        # :52-54
        for p in self.encoder_k.parameters():
            p.requires_grad = False
        self.register_buffer('queue', F.normalize(torch.randn(dim, K), dim=0))   # :57-58
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))      # :60

    @torch.no_grad()
    def momentum_update(self):
        for pq, pk in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            pk.data = pk.data * self.m + pq.data * (1. - self.m)             # :71

    @torch.no_grad()
    def enqueue(self, keys):
        n, ptr = keys.shape[0], int(self.queue_ptr)                          # :74-89
        assert self.K % n == 0
        self.queue[:, ptr:ptr + n] = keys.t()
        self.queue_ptr[0] = (ptr + n) % self.K

    def keys_and_queries(self, im_q, im_k):
        fea, out = self.encoder_q(im_q)
        q = F.normalize(out['q'], dim=1)                                     # :142-143
        with torch.no_grad():
            self.momentum_update()                                           # :147
            k = F.normalize(self.encoder_k(im_k)[1]['q'], dim=1)             # :149-150
        return fea, q, k

    def forward(self, im_q, im_k):
        """training forward -> (embedding, logits, labels) (:140-187)"""
        fea, q, k = self.keys_and_queries(im_q, im_k)
        if self.positives == 1:
            l_pos = torch.einsum('nc,nc->n', [q, k]).unsqueeze(-1)           # :155
            logits = torch.cat([l_pos, torch.einsum('nc,ck->nk', [q, self.queue.clone().detach()])], 1)      # :166,170
            logits /= self.T                                                 # :173
        else:
            l_pos = torch.mul(q.unsqueeze(1), k.reshape(im_q.shape[0], self.positives, self.dim))          # :157-158
            l_pos = (l_pos.sum(dim=2) / self.T).sum(dim=1) / self.positives                                # :160-163
            l_neg = torch.einsum('nc,ck->nk', [q, self.queue.clone().detach()]) / self.T                   # :166,175
            logits = torch.cat([l_pos.unsqueeze(1), l_neg], 1)                                             # :178
        self.enqueue(k if self.positives == 1 else k[[i * self.positives for i in range(im_q.shape[0])]])  # :184-187
        return fea, logits, torch.zeros(logits.shape[0], dtype=torch.long)


class OracleSupMoCo(OracleMoCo):
    """supmoco.py:7-138"""

    def __init__(self, positives_per_class=1, **kw):
        super().__init__(positives=positives_per_class, **kw)
        self.num_classes = 0

    def register_classes(self, num_classes):
        self.num_classes = num_classes                                        # :28-35
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        self.register_buffer('queue_labels', (torch.ones(self.K) * num_classes).to(torch.int64))

    @torch.no_grad()
    def enqueue(self, keys, labels):
        n, ptr = keys.shape[0], int(self.queue_ptr)                           # :38-50
        self.queue[:, ptr:ptr + n] = keys.t()
        self.queue_labels[ptr:ptr + n] = labels
        self.queue_ptr[0] = (ptr + n) % self.K

    def forward(self, im_q, im_k, labels):
        fea, q, k = self.keys_and_queries(im_q, im_k)
        P = self.positives
        l_pos = torch.mul(q.unsqueeze(1), k.reshape(im_q.shape[0], P, self.dim)).sum(dim=2) / self.T          # :88-90
        yb = F.one_hot(labels.to(torch.int64), int(self.num_classes) + 1).float()                             # :94
        yq = F.one_hot(self.queue_labels, int(self.num_classes) + 1).float()                                  # :95
        pos_y_q = torch.matmul(yb, yq.t())                                                                    # :96
        pos_q = (torch.mul(q, torch.matmul(pos_y_q, self.queue.t())) / self.T).sum(dim=1)                     # :99-102
        l_pos = (l_pos.sum(dim=1) + pos_q) / (P + pos_y_q.sum(dim=1))                                         # :105-111
        l_neg = torch.einsum('nc,ck->nk', [q, self.queue.clone().detach()]) / self.T                          # :114
        logits = torch.cat([l_pos.unsqueeze(1), l_neg], dim=1)                                                # :117
        self.enqueue(k[[i * P for i in range(len(labels))]], labels)                                          # :123
        return fea, logits, torch.zeros(logits.shape[0], dtype=torch.long)


class OracleWeakCon(OracleSupMoCo):
    """weak_con.py:7-113: negatives weighted by the distance between degradation vectors"""

    def register_vector(self, vector_size):
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))                                  # :18
        self.register_buffer('queue_vectors', torch.zeros(vector_size, self.K))                             # :19

    @torch.no_grad()
    def enqueue(self, keys, vectors):
        n, ptr = keys.shape[0], int(self.queue_ptr)                                                          # :22-34
        self.queue[:, ptr:ptr + n] = keys.t()
        self.queue_vectors[:, ptr:ptr + n] = vectors
        self.queue_ptr[0] = (ptr + n) % self.K

    def forward(self, im_q, im_k, q_vector):
        fea, q, k = self.keys_and_queries(im_q, im_k)
        P, n = self.positives, q_vector.size()[1]
        l_pos = torch.mul(q.unsqueeze(1), k.reshape(im_q.shape[0], P, self.dim))                             # :64-65
        l_pos = (l_pos.sum(dim=2) / self.T).sum(dim=1) / P                                                   # :66-69
        l_neg = torch.einsum('nc,ck->nk', [q, self.queue.clone().detach()])                                  # :72
        weights = torch.cdist(q_vector.transpose(1, 0), self.queue_vectors.transpose(1, 0))                  # :90
        l_neg = l_neg * weights / self.T                                                                     # :92-94
        logits = torch.cat([l_pos.unsqueeze(1), l_neg], dim=1)                                               # :97
        self.enqueue(k[[i * P for i in range(n)]], q_vector)                                                 # :103
        return fea, logits, torch.zeros(logits.shape[0], dtype=torch.long)


def oracle_supcon_loss(features, labels, temperature=0.07, base_temperature=0.07):
    """rumpy/sr_tools/loss_functions.py:41-130 (SupConLoss, contrast_mode 'all', labels given): features [bsz, views, C]"""
    bsz, views = features.shape[0], features.shape[1]
    mask = torch.eq(labels.contiguous().view(-1, 1), labels.contiguous().view(-1, 1).T).float()             # :83-86
    contrast = torch.cat(torch.unbind(features, dim=1), dim=0)                                              # :91
    adc = torch.div(torch.matmul(contrast, contrast.T), temperature)                                        # :102-104
    logits = adc - torch.max(adc, dim=1, keepdim=True)[0].detach()                                          # :106-107
    mask = mask.repeat(views, views)                                                                        # :110
    logits_mask = torch.scatter(torch.ones_like(mask), 1, torch.arange(bsz * views).view(-1, 1), 0)         # :112-117
    mask = mask * logits_mask
    log_prob = logits - torch.log((torch.exp(logits) * logits_mask).sum(1, keepdim=True) + 1e-6)            # :121-122
    mean_log_prob_pos = (mask * log_prob).sum(1) / mask.sum(1)                                              # :125
    return (-(temperature / base_temperature) * mean_log_prob_pos).view(views, bsz).mean()                  # :128-129


def split_crops(x, crop_count):
    """handlers.py:47-53 / 120-124: [N, crops, 3, H, W] -> (first crop of every image, the other crops)"""
    x = x.view(-1, 3, x.shape[-2], x.shape[-1])
    first = [i * crop_count for i in range(x.shape[0] // crop_count)]
    rest = [i for i in range(x.shape[0]) if i not in first]
    return x[first], x[rest]


class OracleContrastiveHandler:
    """One run_train of MocoContrastiveHandler (handlers.py:37-63) / SupMoCoHandler (:115-158, contrastive loss only): Adam(lr), no
    scheduler; returns what the reference returns plus the logits."""

    def __init__(self, kind, crop_count=2, lr=1e-4, moco_t=0.07, K=32 * 256):
        self.kind, self.crop_count = kind, crop_count
        if kind == 'mococontrastive':
            self.net = OracleMoCo(T=moco_t, positives=crop_count - 1, K=K)
        elif kind == 'weakcon':
            self.net = OracleWeakCon(T=moco_t, positives_per_class=crop_count - 1, K=K)
        elif kind == 'supcon':
            self.net = OracleEncoder()
        else:
            self.net = OracleSupMoCo(T=moco_t, positives_per_class=crop_count - 1, K=K)
        self.optimizer = torch.optim.Adam([p for p in self.net.parameters() if p.requires_grad], lr=lr)

    def run_train(self, x, labels=None):
        self.net.train()
        if self.kind == 'supcon':
            # handlers.py:241-252 with q = the encoder's 'q' output (the reference line indexes the output dict as a tensor and raises)
            x = x.view(-1, 3, x.shape[-2], x.shape[-1])
            fea, out = self.net(x)
            logits = out['q'].view(-1, self.crop_count, out['q'].shape[1])
            loss = oracle_supcon_loss(logits, labels)
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
            return loss.detach(), logits.detach(), fea.detach()
        if self.kind == 'mococontrastive':
            im_q, im_k = (x[:, 0:3], x[:, 3:]) if self.crop_count == 2 else split_crops(x, self.crop_count)
            fea, logits, target = self.net(im_q, im_k)
        else:
            im_q, im_k = split_crops(x, self.crop_count)
            fea, logits, target = self.net(im_q, im_k, labels)
        loss = F.cross_entropy(logits, target)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return loss.detach(), logits.detach(), fea.detach()


# ---- class labels (rumpy/regression/models/contrastive_learning/__init__.py:8-288) ----
def _bucket(v, splits):
    if splits == 2:
        return 1 if v > 0.5 else 0                      # :57-61
    return 2 if v > 0.66 else (1 if v > 0.33 else 0)    # :63-70


def oracle_label_structure(keys, data_type, strategy):
    """(column of every standard name, families, digit weights, class count) from raw metadata keys (:74-152)"""
    names = []
    for key in keys:
        for frag, name in (('gaussian_noise', 'gaussian_noise_scale'), ('poisson_noise', 'poisson_noise_scale'), ('downsample', 'scale'),
                           ('gray_noise', 'gray_noise_boolean'), ('jpeg', 'jpeg_quality_factor'), ('qpi', 'jm_qpi')):
            if frag in key:
                names.append(name)
                break
        else:
            names.append(key.split('realesrganblur-')[-1] if 'realesrganblur' in key else 'unknown')
    col = {}
    for i, n in enumerate(names):
        col.setdefault(n, i)
    accepted = ['blur', 'compression', 'noise'] if data_type == 'all' else data_type
    fam, radix = [], []
    if 'poisson_noise_scale' in col and 'noise' in accepted:
        fam.append('noise')
        radix += {'default': [2, 2], 'double_precision': [2, 2, 2], 'triple_precision': [3, 2, 2]}[strategy]
    if ('jpeg_quality_factor' in col or 'jm_qpi' in col) and 'compression' in accepted:
        fam.append('compression')
        radix += [3] if strategy == 'triple_precision' else [2]
        if 'jpeg_quality_factor' in col and 'jm_qpi' in col:
            radix.append(2)
            fam.append('jm_jpg_compression')
    if 'kernel_type' in col and 'blur' in accepted:
        fam.append('blur')
        radix += [7, 3, 3]
    weights = [int(np.prod(radix[:j])) if j else 1 for j in range(len(radix))]
    return col, fam, weights, int(np.prod(radix))


def oracle_degradation_vector(row, col, fam):
    """:168-200 (vector_retrieval): two slots per family"""
    v = []
    if 'noise' in fam:
        g = row[col['gaussian_noise_scale']]
        v += [g, 0.0] if g > 0 else [0.0, row[col['poisson_noise_scale']]]
    if 'compression' in fam:
        jpeg = ('jpeg_quality_factor' in col and row[col['jpeg_quality_factor']] > 0) or 'jm_qpi' not in col
        v += [row[col['jpeg_quality_factor']], 0.0] if jpeg else [0.0, row[col['jm_qpi']]]
    if 'blur' in fam:
        v += [row[col['sigma_x']], row[col['sigma_y']]]
    return np.asarray(v, dtype=np.float32)


def oracle_class_label(row, col, fam, weights, strategy):
    """:203-288"""
    split = 3 if strategy == 'triple_precision' else 2
    digits = []
    if 'noise' in fam:
        gaussian = row[col['gaussian_noise_scale']] > 0
        mag = row[col['gaussian_noise_scale']] if gaussian else row[col['poisson_noise_scale']]
        if strategy in ('double_precision', 'triple_precision'):
            digits.append(_bucket(mag, split))
        digits += [1 if row[col['gray_noise_boolean']] > 0 else 0, 1 if gaussian else 0]
    if 'compression' in fam:
        jpeg = ('jpeg_quality_factor' in col and row[col['jpeg_quality_factor']] > 0) or 'jm_qpi' not in col
        digits.append(_bucket(row[col['jpeg_quality_factor']] if jpeg else row[col['jm_qpi']], split))
        if 'jm_jpg_compression' in fam:
            digits.append(0 if jpeg else 1)
    if 'blur' in fam:
        digits += [int(row[col['kernel_type']]), _bucket(row[col['sigma_x']], 3), _bucket(row[col['sigma_y']], 3)]
    return sum(d * w for w, d in zip(reversed(weights), reversed(digits)))


# ---- seeded inputs (regenerable without the reference) ----
def contrastive_batch(seed, n, crops, hw=32, shared=0.5):
    """[n, crops, 3, hw, hw] in [0, 1]: every image is a smooth random colour pattern (4 x 4 control points, bilinear) shared by its crops,
    mixed with a per-crop pattern and per-image noise of its own strength - so that encoder features differ between images and between
    the crops of one image (uniform noise alone pools to the same feature vector for every image)."""
    rng = np.random.default_rng(seed)
    up = lambda t: F.interpolate(torch.from_numpy(t.astype(np.float32)), size=(hw, hw), mode='bilinear', align_corners=False)
    base = up(rng.uniform(0, 1, (n, 3, 4, 4)))[:, None]
    own = up(rng.uniform(0, 1, (n * crops, 3, 4, 4))).view(n, crops, 3, hw, hw)
    sigma = torch.from_numpy(rng.uniform(0.0, 0.25, (n, 1, 1, 1, 1)).astype(np.float32))
    noise = torch.from_numpy(rng.standard_normal((n, crops, 3, hw, hw)).astype(np.float32))
    return (shared * base + (1 - shared) * own + sigma * noise).clamp(0, 1).contiguous()


# ---- the blind-SR handler's joint SR + contrastive losses (rumpy/SISR/models/blur_kernel_blind_sr/handlers.py:513-586) ----
class OracleJointPipeline(nn.Module):
    """contrastive_blind_sr.py:159-201,330-348: G (a QRCAN oracle) driven by the embedding of a MoCo / SupMoCo module E"""

    def __init__(self, generator, mode, crop_count, freeze):
        super().__init__()
        self.G = generator
        self.mode = mode
        self.E = OracleMoCo() if mode == 'moco' else OracleSupMoCo(positives_per_class=crop_count - 1)      # :161-166 (MoCo: one positive)
        for name, p in self.E.named_parameters():                                                          # :173-179
            if freeze == 'all' or (freeze == 'pre_q' and 'mlp' not in name):
                p.requires_grad = False

    def forward(self, x, x_key=None, labels=None):
        if self.training:
            fea, logits, target = self.E(x, x_key) if self.mode == 'moco' else self.E(x, x_key, labels)     # :332-335
            return self.G(x, fea.unsqueeze(2).unsqueeze(3)), logits, target                                 # :337-338
        return self.G(x, self.E.encoder_q(x)[0].unsqueeze(2).unsqueeze(3))                                  # :340-348 (embed_digit 0)


class OracleJointHandler:
    """ContrastiveBlindQRCANHandler.run_train for combined_loss_mode 'moco' / 'supmoco' (handlers.py:513-586) and BaseModel.run_eval"""

    def __init__(self, generator, mode, crop_count, freeze, lr=1e-4):
        self.net = OracleJointPipeline(generator, mode, crop_count, freeze)
        self.mode, self.crop_count = mode, crop_count
        self.optimizer = torch.optim.Adam([p for p in self.net.parameters() if p.requires_grad], lr=lr)

    def run_train(self, x, y, labels=None):
        self.net.train()
        x, y = x.flatten(1, 2), y.flatten(1, 2)                                                             # :520-523
        if self.mode == 'moco':
            sr, logits, target = self.net(x[:, 0:3], x[:, 3:])                                              # :527
            y_sr = y[:, 0:3]
        else:
            x, y = x.view(-1, 3, x.shape[2], x.shape[3]), y.view(-1, 3, y.shape[2], y.shape[3])              # :547-548
            first = [i * self.crop_count for i in range(x.shape[0] // self.crop_count)]
            rest = [i for i in range(x.shape[0]) if i not in first]
            sr, logits, target = self.net(x[first], x[rest], labels)                                        # :554
            y_sr = y[first]
        l_con, l_sr = F.cross_entropy(logits, target), F.l1_loss(sr, y_sr)                                  # :530-531 / :558-559
        loss = l_con + l_sr
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'train-loss': loss.detach(), 'l1-loss': l_sr.detach(), 'contrast-loss': l_con.detach()}, logits.detach()

    def run_eval(self, x, y):
        self.net.eval()
        with torch.no_grad():
            out = self.net(x)
            return out, F.l1_loss(out, y)


def joint_batch(seed, n, crops):
    """LR crops [n, crops, 3, 16, 16] and x2 HR targets (bilinear upsampling + a seeded residual) for the joint-loss fixtures"""
    x = contrastive_batch(seed, n, crops, hw=16)
    rng = np.random.default_rng(seed + 7)
    y = F.interpolate(x.view(-1, 3, 16, 16), scale_factor=2, mode='bilinear', align_corners=False).view(n, crops, 3, 32, 32)
    y = (y + torch.from_numpy(rng.uniform(-0.05, 0.05, tuple(y.shape)).astype(np.float32))).clamp(0, 1)
    return x, y

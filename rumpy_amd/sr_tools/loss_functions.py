"""Loss functions of the MI355X path beyond the fused L1 / MSE passes - mirror of rumpy/sr_tools/loss_functions.py for the part the
contrastive handlers use: ``SupConLoss`` (:41-130, the supervised contrastive loss of https://arxiv.org/abs/2004.11362, with SimCLR as its
label-free case).  The perceptual (VGG) loss of that file is outside the hot path.

Plain torch on a [views * batch, views * batch] similarity matrix (rocBLAS GEMM + elementwise ops with torch autograd): the encoder trunk
feeding it is the HIP part (rumpy_amd/regression/models/contrastive_learning/encoding_models.py)."""
import torch
from torch import nn


class SupConLoss(nn.Module):
    def __init__(self, temperature=0.07, contrast_mode='all', base_temperature=0.07):
        super(SupConLoss, self).__init__()
        self.temperature = temperature
        self.contrast_mode = contrast_mode
        self.base_temperature = base_temperature

    def forward(self, features, labels=None, mask=None):
        """features [batch, views, ...]; labels [batch] or mask [batch, batch] (mask[i, j] = 1: j is a positive of i; neither: each sample is
        its own class, the SimCLR loss) -> scalar: minus the mean log-probability of an anchor's positives among all other samples."""
        if features.dim() < 3:
            raise ValueError('`features` needs to be [bsz, n_views, ...],at least 3 dimensions are required')
        features = features.reshape(features.shape[0], features.shape[1], -1)
        bsz, views = features.shape[0], features.shape[1]
        dev = features.device
        if labels is not None and mask is not None:
            raise ValueError('Cannot define both `labels` and `mask`')
        if labels is not None:
            labels = labels.contiguous().view(-1, 1)
            if labels.shape[0] != bsz:
                raise ValueError('Num of labels does not match num of features')
            mask = torch.eq(labels, labels.T).float().to(dev)
        elif mask is None:
            mask = torch.eye(bsz, dtype=torch.float32, device=dev)
        else:
            mask = mask.float().to(dev)
        contrast = features.transpose(0, 1).reshape(views * bsz, -1)          # view-major: all first views, then all second views, ...
        if self.contrast_mode == 'one':
            anchor, anchors = features[:, 0], 1
        elif self.contrast_mode == 'all':
            anchor, anchors = contrast, views
        else:
            raise ValueError('Unknown mode: {}'.format(self.contrast_mode))
        sim = (anchor @ contrast.T) / self.temperature
        sim = sim - sim.max(dim=1, keepdim=True)[0].detach()                  # row shift: numerical stability only
        not_self = torch.ones(anchors * bsz, views * bsz, device=dev)
        not_self[torch.arange(anchors * bsz, device=dev), torch.arange(anchors * bsz, device=dev)] = 0
        positives = mask.repeat(anchors, views) * not_self
        log_prob = sim - torch.log((torch.exp(sim) * not_self).sum(1, keepdim=True) + 1e-6)
        mean_log_prob_pos = (positives * log_prob).sum(1) / positives.sum(1)
        return (-(self.temperature / self.base_temperature) * mean_log_prob_pos).view(anchors, bsz).mean()

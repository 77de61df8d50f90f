"""Why does the full-depth blind QRCAN, precision='fp8', miss the whole-gradient bound (5e-2) under the joint SupMoCo loss (5.26e-2) when the
frozen-encoder step passes it (4.0e-2)?  The same generator, the same weights, on the SAME four 48 x 48 images through the fused-L1 path (frozen
encoder) and in bf16 - if the fused path shows the same error on these images, the difference is the data (N = 4 smooth colour patterns + noise
against N = 2 uniform noise), not the generic-loss path."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import contrastive_oracle as CO  # noqa: E402
from oracle import sr_oracle as O  # noqa: E402
from rumpy_amd.shared_framework.models import define_model  # noqa: E402

KW = dict(scale=4, n_feats=64, n_resgroups=10, n_resblocks=20, reduction=16, style='standard', include_q_layer=True,
          selective_meta_blocks=[True] + [False] * 9, num_q_layers_inner_residual=1)


def stats(named_h, named_o):
    num = den = dot = gg = 0.0
    worst = (0.0, None)
    for (k, p), (_, q) in zip(named_h, named_o):
        g, r = p.grad.detach().float().cpu().double().reshape(-1), q.grad.double().reshape(-1)
        num += float((g - r).pow(2).sum()); den += float(r.pow(2).sum()); dot += float(g @ r); gg += float(g.pow(2).sum())
        if p.dim() == 4 and p.shape[-1] == 3 and float(r.norm()) > 0:
            rel = float((g - r).norm() / r.norm())
            if rel > worst[0]:
                worst = (rel, k)
    return (num / den) ** 0.5, dot / (gg * den) ** 0.5, worst


x = CO.contrastive_batch(4220, 4, 3, hw=48)
rng = np.random.default_rng(4227)
y = torch.nn.functional.interpolate(x.view(-1, 3, 48, 48), scale_factor=4, mode='bilinear', align_corners=False).view(4, 3, 3, 192, 192)
y = (y + torch.from_numpy(rng.uniform(-0.05, 0.05, tuple(y.shape)).astype(np.float32))).clamp(0, 1)
x0, y0 = x[:, 0].contiguous(), y[:, 0].contiguous()
for data, (xa, ya) in (('the joint test\'s four images (smooth pattern + noise)', (x0, y0)), ('uniform noise, N = 4', O.synthetic_batch(4106, 4, lr_hw=48, scale=4)),
                       ('uniform noise, N = 2 (the frozen-encoder test)', O.synthetic_batch(4106, 2, lr_hw=48, scale=4))):
    for prec in ('fp8', None):
        h = define_model('contrastiveblindqrcan', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False,
                         precision=prec, metadata_list=None, block_encoder_loading=True, lr=1e-4, **KW)
        onet = O.build_oracle('contrastiveblindqrcan', **KW)
        sd = O.seeded_pipeline_state(onet, 4105)
        onet.load_state_dict(sd)
        h.net.load_state_dict(sd)
        oh = O.OracleHandler(onet, lr=1e-4)
        h.run_train(x=xa, y=ya)
        oh.run_train(xa, ya)
        w, c, worst = stats(h.net.G.named_parameters(), oh.net.G.named_parameters())
        print('%-52s %-5s fused-L1 path: whole gradient %.3e, cosine %.5f, worst 3x3 tensor %.3e (%s)' % (data, prec or 'bf16', w, c, worst[0], worst[1]), flush=True)
        del h

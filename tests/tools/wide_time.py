"""Time EDSR at the reference's shipped width (div2k/edsr.toml: 256 features x 32 blocks, x4) on one GPU: ms per training step, patches/s,
TFLOP/s (694.7 GFLOP per 48x48 patch and step).   python tests/tools/wide_time.py [N] [steps] [scale]"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rumpy_amd.shared_framework.models import define_model  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 4
h = define_model('edsr', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False, scale=scale,
                 num_features=256, num_blocks=32, res_scale=0.1, lr=1e-4, scheduler='cosine_annealing_warm_restarts',
                 scheduler_params={'t_mult': 1, 'restart_period': 40000, 'lr_min': 1e-7})
g = torch.Generator().manual_seed(1)
x = torch.rand(N, 3, 48, 48, generator=g).cuda()
y = torch.rand(N, 3, 48 * scale, 48 * scale, generator=g).cuda()
for _ in range(3):
    h.run_train(x=x, y=y, keep_on_device=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss, _ = h.run_train(x=x, y=y, keep_on_device=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print('EDSR 256 x 32 x%d, N=%d: %.2f ms/step, %.0f patches/s, %.0f TFLOP/s (x4 count), loss %.4f' %
      (scale, N, dt * 1e3, N / dt, N * 694.7e9 / dt / 1e12, float(loss)))

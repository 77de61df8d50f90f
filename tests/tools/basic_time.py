"""Training-step time of the "basic" models (SRCNN 9-5-5 default, VDSR 20 x 3x3 default) on the direct fp32 convolution kernels, with the
oracle (torch CPU) on the same batch beside it.  python tests/tools/basic_time.py [N H W]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sr_oracle as O
from rumpy_amd.shared_framework.models import define_model

N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 64, 64)
g = np.random.default_rng(1)
x = torch.from_numpy(g.uniform(0, 1, (N, 1, H, W)).astype(np.float32))
y = torch.from_numpy(g.uniform(0, 1, (N, 1, H, W)).astype(np.float32))
for name in ('srcnn', 'vdsr'):
    torch.manual_seed(8)
    h = define_model(name, model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False, lr=1e-4)
    xd, yd = x.cuda(), y.cuda()
    for _ in range(3):
        h.run_train(x=xd, y=yd, keep_on_device=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        h.run_train(x=xd, y=yd, keep_on_device=True)
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / K
    onet = O.build_oracle(name)
    oh = O.OracleHandler(onet, lr=1e-4, criterion='mse', grad_clip=0.1 if name == 'vdsr' else None)
    torch.set_num_threads(min(16, os.cpu_count()))
    oh.run_train(x, y)
    t0 = time.perf_counter()
    for _ in range(2):
        oh.run_train(x, y)
    cpu = (time.perf_counter() - t0) / 2
    print('%s: %d x 1 x %d x %d train step: HIP %.2f ms (%.0f patches/s), oracle on %d host threads %.1f ms (%.1f patches/s)'
          % (name, N, H, W, 1e3 * gpu, N / gpu, min(16, os.cpu_count()), 1e3 * cpu, N / cpu))

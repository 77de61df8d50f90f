"""Time the MoCo encoder-training step (define_model('mococontrastive'), crop_count 2) on one GPU: N query + N key crops per step.
    python tests/tools/moco_time.py [N] [hw] [steps]
Prints ms per step and crops/s; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import sys
import tempfile
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import contrastive_oracle as CO  # noqa: E402  (inputs only)
from rumpy_amd.shared_framework.models import define_model  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 48
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
h = define_model('mococontrastive', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, model_name='default', crop_count=2, lr=1e-4)
xs = [CO.contrastive_batch(10 + i, N, 2, hw=hw).view(N, 6, hw, hw).cuda() for i in range(4)]
for i in range(10):
    h.run_train(x=xs[i % 4], y=None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss, _ = h.run_train(x=xs[i % 4], y=None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print('MoCo step N=%d %dx%d: %.3f ms/step, %.0f crops/s (query + key), loss %.4f' % (N, hw, hw, dt * 1e3, 2 * N / dt, float(loss)))
if len(sys.argv) > 4 and sys.argv[4] == 'cpu':
    # the oracle's MoCo step on the host cores (bounded sample), for the same crops
    oh = CO.OracleContrastiveHandler('mococontrastive', crop_count=2, lr=1e-4)
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    xc = xs[0][:8].cpu()
    oh.run_train(xc)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 10.0:
        oh.run_train(xc)
        n += 1
    dt = (time.perf_counter() - t0) / n
    print('CPU oracle MoCo step N=8: %.1f ms/step, %.0f crops/s on %d threads' % (dt * 1e3, 16 / dt, torch.get_num_threads()))

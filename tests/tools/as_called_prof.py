"""host-side profile of run_train as the reference's caller calls it (pageable host tensors in, image back on the host): where the time of one
call goes (cProfile over 20 calls), with and without the pinned staging (RUMPY_HOST_STAGING)."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sr_oracle as O  # noqa: E402
from rumpy_amd.shared_framework.models import define_model  # noqa: E402

torch.manual_seed(8)
h = define_model('edsr', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False, scale=4, lr=1e-4)
pool = [O.synthetic_batch(1234 + i, 32, lr_hw=48, scale=4) for i in range(4)]
for mode in ('1', '2', '0'):
    os.environ['RUMPY_HOST_STAGING'] = mode
    for i in range(5):
        h.run_train(x=pool[i % 4][0], y=pool[i % 4][1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        h.run_train(x=pool[i % 4][0], y=pool[i % 4][1])
    torch.cuda.synchronize()
    print('RUMPY_HOST_STAGING=%s: %.3f ms per call' % (mode, (time.perf_counter() - t0) / 20 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for i in range(20):
        h.run_train(x=pool[i % 4][0], y=pool[i % 4][1])
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(10)

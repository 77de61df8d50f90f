"""Where does the time of a training step fed by the device patch pipeline go?  (15.2 k vs 25.1 k patches/s)
python tests/tools/patch_gap.py   (GPU box)"""
import os
import random
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rumpy_amd.shared_framework.models import define_model  # noqa: E402
from rumpy_amd.sr_tools.device_patches import DevicePatchSource  # noqa: E402

dev = torch.device('cuda:0')
N = 32
torch.manual_seed(8)
h = define_model('edsr', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False, scale=4, lr=1e-4)
gen = np.random.default_rng(99)
lrs = [gen.integers(0, 256, (192, 256, 3), dtype=np.uint8) for _ in range(64)]
hrs = [gen.integers(0, 256, (768, 1024, 3), dtype=np.uint8) for _ in range(64)]
src = DevicePatchSource(lrs, hrs, 4, 48, device=dev)
random.seed(8)
order = lambda i: [(i * N + k) % len(src) for k in range(N)]
pool = [src.sample(order(i), random) for i in range(8)]


def timed(fn, iters=200, warm=20):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def host_only(fn, iters=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        fn(i)
    dt = (time.perf_counter() - t0) / iters * 1e3
    torch.cuda.synchronize()
    return dt


print('train step, resident pool           : %.3f ms' % timed(lambda i: h.run_train(x=pool[i % 8][0], y=pool[i % 8][1], keep_on_device=True)))
print('sample() alone (GPU + host, pipelined): %.3f ms' % timed(lambda i: src.sample(order(i), random)))
print('  draws only (host)                  : %.3f ms' % host_only(lambda i: [src.draw(k, random) for k in order(i)]))
params = [src.draw(k, random) for k in order(0)]
print('  gather() only, fixed params        : %.3f ms' % timed(lambda i: src.gather(order(0), params)))
print('train step fed by sample()           : %.3f ms' % timed(lambda i: h.run_train(*(lambda xy: dict(x=xy[0], y=xy[1]))(src.sample(order(i), random)).values(), keep_on_device=True)))
x0, y0 = pool[0]
print('train step + fresh torch.empty x/y   : %.3f ms' % timed(lambda i: (torch.empty_like(x0), torch.empty_like(y0), h.run_train(x=pool[i % 8][0], y=pool[i % 8][1], keep_on_device=True))))

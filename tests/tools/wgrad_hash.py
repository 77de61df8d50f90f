"""sha1 of the weight / bias gradients wgrad_dma_kernel<4> and <1> produce on fixed inputs - to compare two builds of the library bit for bit:
   python tests/tools/wgrad_hash.py;  RUMPY_AMD_LIB=build_abl/<tag>/librumpy_amd.so python tests/tools/wgrad_hash.py"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_utils import BF16, DEV, hip_wgrad, nhwc  # noqa: E402


def _rand(gen, *shape):
    return torch.from_numpy(gen.standard_normal(shape).astype(np.float32))


h = hashlib.sha1()
for N, H, W, split in [(2, 12, 12, 1), (4, 16, 32, 2), (3, 9, 21, 3), (8, 48, 48, 1), (1, 8, 16, 1), (5, 40, 56, 5)]:
    gen = np.random.default_rng(17 + H)
    x, gy = _rand(gen, N, 64, H, W), _rand(gen, N, 64, H, W)
    xd, gd = nhwc(x), nhwc(gy)
    per = (N + split - 1) // split
    jobs = [dict(x=xd, dy=gd, n0=n0, n1=min(N, n0 + per), H=H, W=W, x_cstride=64, x_coff=0, dy_mode=0, dy_cstride=64, dy_coff=0) for n0 in range(0, N, per)]
    gw = torch.full((64, 64, 3, 3), float('nan'), device=DEV)
    gb = torch.full((64,), float('nan'), device=DEV)
    hip_wgrad(jobs, 4, [dict(first_job=0, njobs=len(jobs), co_count=64, co_mode=0, co_off=0, ci_total=64, ci_off=0, write_bias=1, scale=0.5)], gw, gb, 1)
    h.update(gw.cpu().numpy().tobytes()); h.update(gb.cpu().numpy().tobytes())
    assert torch.isfinite(gw).all() and torch.isfinite(gb).all()
for W in (20, 23, 34):
    gen = np.random.default_rng(19)
    N, H, C = 2, 24, 3
    x = _rand(gen, N, 64, H, W)
    g4 = torch.zeros(N, H, W, 4, dtype=BF16)
    g4[..., :C] = torch.sign(_rand(gen, N, C, H, W)).permute(0, 2, 3, 1).to(BF16)
    xd, gd = nhwc(x), g4.to(DEV)
    jobs = [dict(x=xd, dy=gd, n0=n, n1=n + 1, H=H, W=W, x_cstride=64, x_coff=0, dy_mode=2, dy_cstride=4, dy_coff=0) for n in range(N)]
    gw = torch.full((C, 64, 3, 3), float('nan'), device=DEV)
    gb = torch.full((C,), float('nan'), device=DEV)
    hip_wgrad(jobs, 1, [dict(first_job=0, njobs=N, co_count=C, co_mode=0, co_off=0, ci_total=64, ci_off=0, write_bias=1, scale=1.0 / 7)], gw, gb, 1)
    h.update(gw.cpu().numpy().tobytes()); h.update(gb.cpu().numpy().tobytes())
print('wgrad sha1', h.hexdigest())

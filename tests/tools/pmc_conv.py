"""Run only the headline 64->64 conv a few times (for rocprofv3 --pmc): python tests/tools/pmc_conv.py [H]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_utils import BF16, DEV, PackedConv, stream
from rumpy_amd import _lib as L
H = int(sys.argv[1]) if len(sys.argv) > 1 else 48
N, W = 32, H
gen = np.random.default_rng(0)
pc = PackedConv(torch.from_numpy(gen.uniform(-0.04, 0.04, (64, 64, 3, 3)).astype(np.float32)), torch.zeros(64))
x = torch.randn(N, H, W, 64, device=DEV).to(BF16)
res = torch.randn(N, H, W, 64, device=DEV).to(BF16)
out = torch.empty(N, H, W, 64, dtype=BF16, device=DEV)
a = L.ConvArgs(x=x.data_ptr(), w=pc.w_fwd.data_ptr(), bias=pc.b_packed.data_ptr(), out=out.data_ptr(), res1=res.data_ptr(),
               N=N, H=H, W=W, cin_chunks=1, cout_tiles=1, scale=1.0, grid_x=0)
for _ in range(20):
    L.call('rumpy_conv3x3', a, stream())
torch.cuda.synchronize()

"""Run only the residual-block kernel on the headline shape a few times (for rocprofv3 --pmc): python tests/tools/pmc_block.py [fwd|bwd]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_utils import BF16, DEV, PackedConv, stream
from rumpy_amd import _lib as L
mode = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
N, H, W = 32, 48, 48
gen = np.random.default_rng(0)
mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.04, 0.04, (64, 64, 3, 3)).astype(np.float32)), torch.zeros(64))
pa, pb = mk(), mk()
x = torch.randn(N, H, W, 64, device=DEV).to(BF16)
m = torch.randn(N, H, W, 64, device=DEV).to(BF16)
t = torch.empty(N, H, W, 64, dtype=BF16, device=DEV)
out = torch.empty(N, H, W, 64, dtype=BF16, device=DEV)
if mode == 'fwd':
    a = L.BlockArgs(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                    t=t.data_ptr(), out=out.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=0.1)
else:
    a = L.BlockArgs(x=x.data_ptr(), w1=pb.w_dgrad.data_ptr(), w2=pa.w_dgrad.data_ptr(), mask=m.data_ptr(), t=t.data_ptr(),
                    out=out.data_ptr(), N=N, H=H, W=W, relu1=0, scale1=0.1, scale2=1.0)
for _ in range(20):
    L.call('rumpy_conv_block', a, stream())
torch.cuda.synchronize()

"""What does it take to make a chain hand-off time out?  rumpy_res_chain (6 blocks, 32 x 48 x 48 = 256 strips) next to rumpy_debug_occupy with K workgroups for 1.5 s:
status word and launch time per K."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_utils import BF16, DEV, PackedConv, stream, to_dev_bytes
from rumpy_amd import _lib as L
N, H, W, nblk = 32, 48, 48, 6
gen = np.random.default_rng(3)
mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.05, 0.05, (64, 64, 3, 3)).astype(np.float32)), torch.from_numpy(gen.uniform(-0.1, 0.1, 64).astype(np.float32)))
convs = [(mk(), mk()) for _ in range(nblk)]
x0 = torch.randn(N, H, W, 64, device=DEV).to(BF16)
ts = [torch.empty(N, H, W, 64, dtype=BF16, device=DEV) for _ in range(nblk)]
ys = [torch.empty(N, H, W, 64, dtype=BF16, device=DEV) for _ in range(nblk)]
mbs = [torch.zeros(N, H, W, 8, dtype=torch.uint8, device=DEV) for _ in range(nblk)]
recs = [dict(x=(x0 if b == 0 else ys[b - 1]).data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(), res2=None,
             t=ts[b].data_ptr(), out=ys[b].data_ptr(), maskbits=mbs[b].data_ptr(), scale1=1.0, scale2=0.1) for b, (pa, pb) in enumerate(convs)]
tab = to_dev_bytes((L.ResChainBlock * nblk)(*[L.ResChainBlock(**r) for r in recs]))
work = torch.zeros(int(L.lib().rumpy_res_chain_work_bytes(N, H)), dtype=torch.uint8, device=DEV)
status = torch.zeros(1, dtype=torch.int32, device=DEV)
a = L.ResChainArgs(blocks=tab.data_ptr(), nblocks=nblk, N=N, H=H, W=W, backward=0, fmt=0, work=work.data_ptr(), work_bytes=work.numel(), status=status.data_ptr(), fake_xcc=0, force_sc1=0)
L.call('rumpy_res_chain', a, stream()); torch.cuda.synchronize()
side = torch.cuda.Stream()
for k in (0, 8, 24, 48, 96, 160, 256, 400):
    status.zero_(); torch.cuda.synchronize()
    if k:
        L.check(L.lib().rumpy_debug_occupy(k, 1.5e6, side.cuda_stream), 'occupy')
        time.sleep(0.05)
    t0 = time.time()
    L.call('rumpy_res_chain', a, stream())
    torch.cuda.current_stream().synchronize()
    dt = time.time() - t0
    torch.cuda.synchronize()
    print('occupiers %3d: chain launch took %.3f s, status 0x%x' % (k, dt, int(status.item())), flush=True)

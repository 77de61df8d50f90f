"""Isolated time of a run of residual blocks (32 x 48 x 48, training forms) per launch form: one launch per block, the persistent chain with two waves per
SIMD (rumpy_res_chain, conv_chain.hip) and with one wave per SIMD (rumpy_res_chain1, conv_chain1.hip) - alternating, same buffers, HIP events.

  python tests/tools/chain_forms_time.py [nblk] [N] [reps]       (conv_chain1.hip lives in the experimental library: make -C tests/tools/csrc EXTRA=-DC1_STAMPS clean all builds its stamps form, C1_STAMPS=1 prints them;
   RUMPY_EXP_LIB selects another build)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_utils import BF16, DEV, PackedConv, exp_call, exp_lib, stream, to_dev_bytes  # noqa: E402
from rumpy_amd import _lib as L  # noqa: E402

nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
H = W = 48
gen = np.random.default_rng(5)
mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.05, 0.05, (64, 64, 3, 3)).astype(np.float32)), torch.from_numpy(gen.uniform(-0.1, 0.1, 64).astype(np.float32)))
convs = [(mk(), mk()) for _ in range(nblk)]
rnd = lambda: torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
x0 = rnd()
forms = ['blocks', 'rumpy_res_chain', 'rumpy_res_chain1']
for backward in (0, 1):
    ts = [torch.empty(N, H, W, 64, dtype=BF16, device=DEV) for _ in range(nblk)]
    ys = [torch.empty(N, H, W, 64, dtype=BF16, device=DEV) for _ in range(nblk)]
    mbs = [torch.from_numpy(gen.integers(0, 256, (N, H, W, 8), dtype=np.uint8)).to(DEV) for _ in range(nblk)]
    recs = []
    for b, (pa, pb) in enumerate(convs):
        xin = x0 if b == 0 else ys[b - 1]
        if backward:
            recs.append(dict(x=xin.data_ptr(), w1=pb.w_dgrad.data_ptr(), b1=None, w2=pa.w_dgrad.data_ptr(), b2=None, res2=None, t=ts[b].data_ptr(), out=ys[b].data_ptr(),
                             maskbits=mbs[b].data_ptr(), scale1=0.1, scale2=1.0))
        else:
            recs.append(dict(x=xin.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(), res2=None, t=ts[b].data_ptr(),
                             out=ys[b].data_ptr(), maskbits=mbs[b].data_ptr(), scale1=1.0, scale2=0.1))
    tab = to_dev_bytes((L.ResChainBlock * nblk)(*[L.ResChainBlock(**r) for r in recs]))
    work = torch.zeros(int(L.lib().rumpy_res_chain_work_bytes(N, H)), dtype=torch.uint8, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    a = L.ResChainArgs(blocks=tab.data_ptr(), nblocks=nblk, N=N, H=H, W=W, backward=backward, fmt=0, work=work.data_ptr(), work_bytes=work.numel(), status=status.data_ptr(),
                       fake_xcc=0, force_sc1=0)
    blocks = [L.BlockArgs(N=N, H=H, W=W, relu1=0 if backward else 1, fmt=0, **r) for r in recs]

    def run(form):
        if form == 'blocks':
            for ba in blocks:
                L.call('rumpy_conv_block', ba, stream())
        else:
            (exp_call if form == 'rumpy_res_chain1' else L.call)(form, a, stream())
    ref = None
    for form in forms:
        run(form)
        torch.cuda.synchronize()
        cur = [t.clone() for t in ts + ys]
        if ref is None:
            ref = cur
        else:
            diff = [i for i, (p, q) in enumerate(zip(ref, cur)) if not torch.equal(p.view(torch.int16), q.view(torch.int16))]
            bad = len(diff)
            print('%s %s: %d of %d buffers differ from the per-block launches; status 0x%x' % ('backward' if backward else 'forward', form, bad, len(cur), int(status.item())), flush=True)
            if diff and os.environ.get('C1_DEBUG'):
                for i in diff[:4]:
                    d = (ref[i].view(torch.int16) != cur[i].view(torch.int16))
                    rows = d.any(dim=3).any(dim=2).any(dim=0).nonzero().flatten().tolist()      # image rows with a difference
                    cols = d.any(dim=3).any(dim=1).any(dim=0).nonzero().flatten().tolist()
                    chans = d.any(dim=2).any(dim=1).any(dim=0).nonzero().flatten().tolist()
                    imgs = d.any(dim=3).any(dim=2).any(dim=1).nonzero().flatten().tolist()
                    print('   %s[%d]: rows %s cols %s channels %s images %s' % ('t' if i < nblk else 'y', i % nblk, rows, cols[:6] + ['..'] + cols[-3:], chans[:6] + ['..'] + chans[-3:], imgs[:8]), flush=True)
    times = {f: [] for f in forms}
    for rnd_ in range(3):
        for form in forms:
            for _ in range(3):
                run(form)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(form)
            e1.record()
            torch.cuda.synchronize()
            times[form].append(e0.elapsed_time(e1) / reps * 1e3)
    for form in forms:
        print('%-8s %-18s %s us per %d blocks = %.2f us per block' % ('backward' if backward else 'forward', form, ' '.join('%.1f' % t for t in times[form]), nblk,
                                                                         min(times[form]) / nblk), flush=True)
    if os.environ.get('C1_STAMPS'):
        import ctypes
        fn = getattr(exp_lib(), 'rumpy_debug_c1_stamps')
        fn.argtypes = [ctypes.c_void_p]
        nwg = N * ((H + 5) // 6)
        buf = torch.zeros(2 * nwg * 4 * 16, dtype=torch.int64, device=DEV)
        assert fn(buf.data_ptr()) == 0
        run('rumpy_res_chain1')
        torch.cuda.synchronize()
        both = buf.cpu().numpy().reshape(2, nwg, 4, 16).astype(np.float64)
        raw, cyc = both[0], both[1]
        names = ['block start', '(1) sweep T rows 2-5', '(2) acks + publish', '(3) epilogue of T rows 2-5', '(4) flags seen, rows requested', '(5) sweep OUT rows 2,3',
                 '(6) epilogue of OUT rows 2,3', 'halo rows in LDS + gate', '(7) sweep T rows 0,1,6,7', 'epilogue of them + T gate', '(8) sweep OUT rows 0,1,4,5', 'epilogue of them',
                 'OUT gate + stores']
        nst = len(names) - 1
        rel = (raw[:, :, 1:nst + 1] - raw[:, :, 0:nst]) * 0.01
        print('   phase durations in the middle block, us (mean over all waves | max):')
        crel = cyc[:, :, 1:nst + 1] - cyc[:, :, 0:nst]
        for k in range(nst):
            print('   %-34s %6.2f | %6.2f   %7.0f shader cycles' % (names[k + 1], rel[:, :, k].mean(), rel[:, :, k].max(), crel[:, :, k].mean()))
        tot_us, tot_cyc = ((raw[:, :, nst] - raw[:, :, 0]) * 0.01).mean(), (cyc[:, :, nst] - cyc[:, :, 0]).mean()
        print('   shader clock over the block: %.0f cycles / %.2f us = %.2f GHz' % (tot_cyc, tot_us, tot_cyc / tot_us / 1e3))
        print('   %-28s %6.2f' % ('block (start -> stores issued)', ((raw[:, :, nst] - raw[:, :, 0]) * 0.01).mean()))

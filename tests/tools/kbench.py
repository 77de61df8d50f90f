"""Micro-benchmarks of single kernels on the bench shapes (GPU box only): python tests/tools/kbench.py [conv|wgrad|all]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_utils import BF16, DEV, SPLIT_FORK, SPLIT_JOIN, SPLIT_ONE_STREAM, BlockChainArgs, BlockSplitArgs, ChainArgs, ChainLayer, PackedConv, exp_call, exp_lib, hip_wgrad, stream, to_dev_bytes  # noqa: E402
from rumpy_amd import _lib as L  # noqa: E402


def time_fn(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters  # us


def conv_bench(N=32, H=48, W=48):
    gen = np.random.default_rng(0)
    w = torch.from_numpy(gen.uniform(-0.04, 0.04, (64, 64, 3, 3)).astype(np.float32))
    b = torch.zeros(64)
    pc = PackedConv(w, b)
    xs = [torch.randn(N, H, W, 64, device=DEV).to(BF16) for _ in range(4)]
    out = torch.empty(N, H, W, 64, dtype=BF16, device=DEV)
    res = torch.randn(N, H, W, 64, device=DEV).to(BF16)
    flop = 2.0 * N * H * W * 64 * 576
    for gx in (0, 128, 192, 256, 288, 384, 512, 576):
        k = [0]

        def fn():
            a = L.ConvArgs(x=xs[k[0] % 4].data_ptr(), w=pc.w_fwd.data_ptr(), bias=pc.b_packed.data_ptr(), out=out.data_ptr(),
                           res1=res.data_ptr(), N=N, H=H, W=W, cin_chunks=1, cout_tiles=1, in_mode=0, out_mode=0, relu=0,
                           scale=0.1, grid_x=gx)
            L.call('rumpy_conv3x3', a, stream())
            k[0] += 1
        us = time_fn(fn)
        print('conv 64->64 %dx%dx%d grid_x=%3d: %7.2f us  %6.1f TFLOP/s' % (N, H, W, gx, us, flop / us / 1e6))


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('conv', 'all'):
        conv_bench()
        conv_bench(32, 96, 96)


def conv_stamps(N=32, H=48, W=48):
    gen = np.random.default_rng(0)
    w = torch.from_numpy(gen.uniform(-0.04, 0.04, (64, 64, 3, 3)).astype(np.float32))
    pc = PackedConv(w, torch.zeros(64))
    x = torch.randn(N, H, W, 64, device=DEV).to(BF16)
    out = torch.empty(N, H, W, 64, dtype=BF16, device=DEV)
    res = torch.randn(N, H, W, 64, device=DEV).to(BF16)
    gx = 256
    dbg = torch.zeros(gx * 8 * 8, dtype=torch.int64, device=DEV)
    a = L.ConvArgs(x=x.data_ptr(), w=pc.w_fwd.data_ptr(), bias=pc.b_packed.data_ptr(), out=out.data_ptr(), res1=res.data_ptr(),
                   pool=dbg.data_ptr(), N=N, H=H, W=W, cin_chunks=1, cout_tiles=1, scale=0.1, grid_x=gx)
    for variant in (0, 1, 2):
        a.relu = variant
        for _ in range(3):
            L.call('rumpy_debug_conv_stamps', a, stream())
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(gx, 8, 8).astype(np.float64)
        t0 = d[:, :, 0].min()
        rel = (d - t0) * 10.0 / 1000.0    # us
        names = ['start', 'loads issued', 'stage in LDS', 'mfma issued', '-', 'epilogue+barrier', '-', '-']
        print('variant %d (0 = product code, 1 = MFMAs without LDS reads, 2 = LDS reads without MFMAs)' % variant)
        for i in (2, 3, 5):
            v = rel[:, :, i]
            print('  stamp %d %-18s median %6.2f us  min %6.2f  max %6.2f' % (i, names[i], np.median(v), v.min(), v.max()))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'stamps':
    conv_stamps()


def chain_bench(N=32, H=48, W=48, nlayers=33):
    gen = np.random.default_rng(0)
    x = torch.randn(N, H, W, 64, device=DEV).to(BF16)
    pcs = [PackedConv(torch.from_numpy(gen.uniform(-0.04, 0.04, (64, 64, 3, 3)).astype(np.float32)), torch.zeros(64)) for _ in range(nlayers)]
    outs = [torch.empty(N, H, W, 64, dtype=BF16, device=DEV) for _ in range(nlayers)]
    layers = []
    for l in range(nlayers):
        res = (outs[l - 2] if l >= 2 else x) if (l % 2 == 1) else None
        layers.append(ChainLayer(w=pcs[l].w_fwd.data_ptr(), bias=pcs[l].b_packed.data_ptr(), out=outs[l].data_ptr(),
                                   res1=(res.data_ptr() if res is not None else None), relu=1 if l % 2 == 0 else 0,
                                   scale=1.0 if l % 2 == 0 else 0.1))
    ldev = to_dev_bytes((ChainLayer * nlayers)(*layers))
    nstrips = N * ((H + 5) // 6)
    xchg = torch.zeros(int(exp_lib().rumpy_conv_chain_xchg_bytes(nstrips)), dtype=torch.uint8, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    a = ChainArgs(x=x.data_ptr(), layers=ldev.data_ptr(), nlayers=nlayers, N=N, H=H, W=W,
                    xchg=xchg.data_ptr(), status=status.data_ptr())
    us = time_fn(lambda: exp_call('rumpy_conv_chain', a, stream()), iters=20)
    st = torch.zeros(nstrips * 8 * 8 * 8, dtype=torch.int64, device=DEV)
    a.stamps = st.data_ptr()
    exp_call('rumpy_conv_chain', a, stream())
    torch.cuda.synchronize()
    a.stamps = None
    d = st.cpu().numpy().reshape(nstrips, 8, 8, 8).astype(np.float64)
    t0 = d[:, :, 0, 0].min()
    names = ['layer start', 'A done', 'early epilogue', 'halo in LDS', 'mid barrier', 'B done', 'late epilogue', 'end barrier']
    for l in (0, 1, 2, 5):
        for kk in range(8):
            v = (d[:, :, l, kk] - t0) / 100.0
            v = v[d[:, :, l, kk] > 0]
            if v.size:
                print('layer %d %-16s median %7.2f us  min %7.2f  max %7.2f' % (l, names[kk], np.median(v), v.min(), v.max()))
    flop = 2.0 * N * H * W * 64 * 576 * nlayers
    print('chain of %d layers: %8.1f us  = %6.2f us/layer  %6.1f TFLOP/s  status %d' % (nlayers, us, us / nlayers, flop / us / 1e6, int(status.item())))

    def per_layer():
        cur = x
        for l in range(nlayers):
            res = (outs[l - 2] if l >= 2 else x) if (l % 2 == 1) else None
            aa = L.ConvArgs(x=cur.data_ptr(), w=pcs[l].w_fwd.data_ptr(), bias=pcs[l].b_packed.data_ptr(), out=outs[l].data_ptr(),
                            res1=(res.data_ptr() if res is not None else None), N=N, H=H, W=W, cin_chunks=1, cout_tiles=1,
                            relu=1 if l % 2 == 0 else 0, scale=1.0 if l % 2 == 0 else 0.1, grid_x=0)
            L.call('rumpy_conv3x3', aa, stream())
            cur = outs[l]
    us2 = time_fn(per_layer, iters=20)
    print('same %d layers, one launch each: %8.1f us = %6.2f us/layer' % (nlayers, us2, us2 / nlayers))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'chain':
    chain_bench()


def conv4_bench(N=32, H=96, W=96):
    """data gradient of the second upsampler conv (64 -> 256 + PixelShuffle): 256 -> 64 gathered from [N,2H,2W,64]"""
    gen = np.random.default_rng(0)
    pc = PackedConv(torch.from_numpy(gen.uniform(-0.04, 0.04, (256, 64, 3, 3)).astype(np.float32)), torch.zeros(256), shuffle=True)
    dy = torch.randn(N, 2 * H, 2 * W, 64, device=DEV).to(BF16)
    out = torch.empty(N, H, W, 64, dtype=BF16, device=DEV)
    a = L.ConvArgs(x=dy.data_ptr(), w=pc.w_dgrad.data_ptr(), out=out.data_ptr(), N=N, H=H, W=W, cin_chunks=4, cout_tiles=1,
                   in_mode=1, out_mode=0, relu=0, scale=1.0, grid_x=0)
    us = time_fn(lambda: L.call('rumpy_conv3x3', a, stream()), iters=20)
    flop = 2.0 * N * H * W * 64 * 256 * 9
    print('conv 256->64 (PixelShuffle^T gather) %dx%dx%d: %7.2f us  %6.1f TFLOP/s  %5.2f TB/s input' % (N, H, W, us, flop / us / 1e6, dy.numel() * 2 / us / 1e6))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'conv4':
    conv4_bench()
    conv4_bench(32, 48, 48)


def patches_bench(N=32, crop=48, scale=4):
    """device-side patch pipeline against the CPU oracle doing the reference's per-item work"""
    import random
    import time
    from oracle import patch_oracle as PO
    from rumpy_amd.sr_tools.device_patches import DevicePatchSource
    lrs, hrs = PO.synthetic_images(3, [(192, 256)] * 16, scale)
    src = DevicePatchSource(lrs, hrs, scale, crop, device=DEV)
    order = [i % len(src) for i in range(N)]
    random.seed(1)
    params = [src.draw(i, random) for i in order]
    us_kernel = time_fn(lambda: src.gather(order, params), iters=50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        src.sample(order, random)
    torch.cuda.synchronize()
    us_total = (time.perf_counter() - t0) / 50 * 1e6
    t0 = time.perf_counter()
    for _ in range(5):
        for i in order:
            PO.sample_patch(lrs[i], hrs[i], crop, scale, random)
    us_cpu = (time.perf_counter() - t0) / 5 * 1e6
    out_mb = N * 3 * (crop * crop + (crop * scale) ** 2) * 4 / 1e6
    print('patch batch N=%d: gather (host item table + 2 launches) %7.1f us  incl. random draws %7.1f us  = %6.0f k patches/s; '
          'output %.1f MB; CPU oracle (ToTensor of the whole image + flips + crop, 1 thread) %9.1f us = %5.2f k patches/s'
          % (N, us_kernel, us_total, N / us_total * 1e3, out_mb, us_cpu, N / us_cpu * 1e3))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'patches':
    patches_bench()


def block_bench(N=32, H=48, W=48, nblocks=16):
    """EDSR body as residual blocks: one launch per block (conv_block.hip) against two launches per block"""
    gen = np.random.default_rng(0)
    mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.04, 0.04, (64, 64, 3, 3)).astype(np.float32)), torch.zeros(64))
    pcs = [(mk(), mk()) for _ in range(nblocks)]
    bufs = [torch.randn(N, H, W, 64, device=DEV).to(BF16) for _ in range(nblocks + 1)]
    ts = [torch.empty(N, H, W, 64, dtype=BF16, device=DEV) for _ in range(nblocks)]

    dbg = torch.zeros(N * ((H + 5) // 6) * 8 * 16, dtype=torch.int64, device=DEV)     # stamp builds only (BLOCK_ABL=9)

    def fused():
        for b, (pa, pb) in enumerate(pcs):
            a = L.BlockArgs(x=bufs[b].data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(),
                            b2=pb.b_packed.data_ptr(), t=ts[b].data_ptr(), out=bufs[b + 1].data_ptr(), N=N, H=H, W=W, relu1=1,
                            scale1=1.0, scale2=0.1, res1=dbg.data_ptr())
            L.call('rumpy_conv_block', a, stream())

    def separate():
        for b, (pa, pb) in enumerate(pcs):
            a1 = L.ConvArgs(x=bufs[b].data_ptr(), w=pa.w_fwd.data_ptr(), bias=pa.b_packed.data_ptr(), out=ts[b].data_ptr(), N=N, H=H, W=W,
                            cin_chunks=1, cout_tiles=1, relu=1, scale=1.0, grid_x=0)
            L.call('rumpy_conv3x3', a1, stream())
            a2 = L.ConvArgs(x=ts[b].data_ptr(), w=pb.w_fwd.data_ptr(), bias=pb.b_packed.data_ptr(), out=bufs[b + 1].data_ptr(),
                            res1=bufs[b].data_ptr(), N=N, H=H, W=W, cin_chunks=1, cout_tiles=1, relu=0, scale=0.1, grid_x=0)
            L.call('rumpy_conv3x3', a2, stream())
    bits = torch.zeros(N, H, W, 8, dtype=torch.uint8, device=DEV)

    def split(one_stream=False):
        # half-strip launches of the two batch halves (conv_hblock.hip): two chains between one fork and one join
        for b, (pa, pb) in enumerate(pcs):
            a = L.BlockArgs(x=bufs[b].data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(),
                            b2=pb.b_packed.data_ptr(), t=ts[b].data_ptr(), out=bufs[b + 1].data_ptr(), N=N, H=H, W=W, relu1=1,
                            scale1=1.0, scale2=0.1, maskbits=bits.data_ptr())
            fl = SPLIT_ONE_STREAM if one_stream else ((SPLIT_FORK if b == 0 else 0) | (SPLIT_JOIN if b == nblocks - 1 else 0))
            exp_call('rumpy_conv_block_split', BlockSplitArgs(block=a, flags=fl), stream())
    for _ in range(2):
        us_f = time_fn(fused, iters=20)
        us_s = time_fn(separate, iters=20) if not os.environ.get('KBENCH_FUSED_ONLY') else 0.0
        print('%d residual blocks %dx%dx%d: one launch per block %8.1f us = %6.2f us/block ; two launches per block %8.1f us = %6.2f us/block'
              % (nblocks, N, H, W, us_f, us_f / nblocks, us_s, us_s / nblocks))
        if W <= 48:
            us_2 = time_fn(split, iters=20)
            us_1 = time_fn(lambda: split(True), iters=20)
            print('   half-strip launches of the two batch halves: on two streams %8.1f us = %6.2f us/block ; on one stream %8.1f us = %6.2f us/block'
                  % (us_2, us_2 / nblocks, us_1, us_1 / nblocks))
    if W <= 48:
        # the same chains replayed from captured graphs: no host time between the launches
        def graphed(fn):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            return g
        gf, gs = graphed(fused), graphed(split)
        for _ in range(2):
            us_f = time_fn(gf.replay, iters=20)
            us_2 = time_fn(gs.replay, iters=20)
            print('   graph replays: one launch per block %8.1f us = %6.2f us/block ; half-strip launches on two streams %8.1f us = %6.2f us/block'
                  % (us_f, us_f / nblocks, us_2, us_2 / nblocks))
    if 'BLOCK_ABL_9' in os.environ.get('RUMPY_AMD_LIB', ''):      # in-kernel phase stamps (100 MHz), written over the first bytes of t
        fused()
        torch.cuda.synchronize()
        nst = N * ((H + 5) // 6)
        rawu = dbg.cpu().numpy().view(np.uint64).reshape(nst, 8, 16)
        raw = rawu[:, :, :8].astype(np.float64)
        cyc = rawu[:, :, 8:12].astype(np.float64)
        k = int((raw[0, 0] > 0).sum())
        t0 = raw[:, :, 0].min()
        rel = (raw[:, :, :k] - raw[:, :, :1]) * 0.01
        print('   stamps, us from each wave\'s start (mean over waves): ' + ' '.join('%.2f' % v for v in rel.mean((0, 1))))
        print('   per row half rh=0: ' + ' '.join('%.2f' % v for v in rel[:, :4].mean((0, 1))) + ' | rh=1: ' + ' '.join('%.2f' % v for v in rel[:, 4:].mean((0, 1))))
        for name, a, b, sa, sb, nm in (('first', 0, 1, 1, 2, 216), ('second', 2, 3, 4, 5, 162)):
            dc, dt = cyc[:, :, b] - cyc[:, :, a], (raw[:, :, sb] - raw[:, :, sa]) * 0.01
            print('   %s sweep: %.0f shader cycles per wave (= %.1f per MFMA of this wave; the SIMD\'s other wave issues as many) in %.2f us -> clock %.2f GHz'
                  % (name, dc.mean(), dc.mean() / nm, dt.mean(), dc.mean() / dt.mean() / 1e3))
        print('   wave start spread over the launch %.2f us; last end - first start %.2f us' % ((raw[:, :, 0].max() - t0) * 0.01, (raw[:, :, k - 1].max() - t0) * 0.01))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'block':
    block_bench()


def bchain_bench(N=32, H=48, W=48, nblocks=16):
    """EDSR body: one launch for the whole chain of residual blocks against one launch per block"""
    gen = np.random.default_rng(0)
    mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.04, 0.04, (64, 64, 3, 3)).astype(np.float32)), torch.zeros(64))
    pcs = [(mk(), mk()) for _ in range(nblocks)]
    bufs = [torch.randn(N, H, W, 64, device=DEV).to(BF16) for _ in range(nblocks + 1)]
    ts = [torch.empty(N, H, W, 64, dtype=BF16, device=DEV) for _ in range(nblocks)]
    items = [L.BlockArgs(x=bufs[b].data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(),
                         b2=pb.b_packed.data_ptr(), t=ts[b].data_ptr(), out=bufs[b + 1].data_ptr(), N=N, H=H, W=W, relu1=1,
                         scale1=1.0, scale2=0.1) for b, (pa, pb) in enumerate(pcs)]
    tdev = to_dev_bytes((L.BlockArgs * nblocks)(*items))
    nstrips = N * ((H + 5) // 6)
    xb = int(exp_lib().rumpy_block_chain_xchg_bytes(nstrips))
    xchg = torch.zeros(xb + nstrips * 8 * 16 * 8, dtype=torch.uint8, device=DEV)       # + room for the stamps of a BCHAIN_ABL=9 build
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    a = BlockChainArgs(blocks=tdev.data_ptr(), nblocks=nblocks, N=N, H=H, W=W, masked=0, xchg=xchg.data_ptr(), status=status.data_ptr())

    def per_block():
        for it in items:
            L.call('rumpy_conv_block', it, stream())
    for _ in range(2):
        us_c = time_fn(lambda: exp_call('rumpy_block_chain', a, stream()), iters=20)
        us_b = time_fn(per_block, iters=20)
        print('%d residual blocks %dx%dx%d: one launch for the chain %8.1f us = %6.2f us/block (status %d); one launch per block %8.1f us = %6.2f us/block'
              % (nblocks, N, H, W, us_c, us_c / nblocks, int(status.item()), us_b, us_b / nblocks))


    if 'BCHAIN_ABL_9' in os.environ.get('RUMPY_EXP_LIB', ''):
        raw = xchg[xb:].cpu().numpy().view(np.uint64).reshape(nstrips, 8, 16)[:, :, :10].astype(np.float64)
        k = int((raw[0, 0] > 0).sum())
        rel = (raw[:, :, :k] - raw[:, :, :1]) * 0.01
        print('   block 8 stamps, us from the block start: rh=0: ' + ' '.join('%.2f' % v for v in rel[:, :4].mean((0, 1))) + ' | rh=1: ' + ' '.join('%.2f' % v for v in rel[:, 4:].mean((0, 1))))
        print('   block-8 start spread over the strips %.2f us' % ((raw[:, :, 0].max() - raw[:, :, 0].min()) * 0.01))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'bchain':
    bchain_bench()


def eval_bench():
    """whole-image inference (run_eval) of EDSR-baseline x4 and RCAN x4 on DIV2K-sized and Set5-sized LR images"""
    import tempfile
    import time
    from rumpy_amd.shared_framework.models import define_model
    for name in ('edsr', 'rcan'):
        h = define_model(name, model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=True, checkpoint_load=False, loss_masking=False,
                         metadata_list=None, scale=4)
        for (n, hh, ww) in ((1, 339, 510), (1, 128, 128), (8, 48, 48)):
            x = torch.rand(n, 3, hh, ww, device=DEV)
            for _ in range(3):
                h.run_eval(x=x, keep_on_device=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            iters = 20
            for _ in range(iters):
                h.run_eval(x=x, keep_on_device=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / iters * 1e3
            print('%s x4 eval %dx%dx%d LR -> %dx%d HR: %7.2f ms per batch = %6.1f images/s, %6.1f HR Mpixel/s'
                  % (name, n, hh, ww, 4 * hh, 4 * ww, ms, n / ms * 1e3, n * 16 * hh * ww / ms / 1e3))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'eval':
    eval_bench()


def rcab_bench(N=32, H=48, W=48, reps=40):
    """one-launch RCAB kernels (conv_rcab.hip) on the headline shape; with RUMPY_AMD_LIB=build_abl/RCAB_ABL_9 also the in-kernel phase stamps"""
    gen = np.random.default_rng(0)
    mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.04, 0.04, (64, 64, 3, 3)).astype(np.float32)), torch.zeros(64))
    pa, pb = mk(), mk()
    f32 = lambda *s: torch.from_numpy(gen.uniform(-0.3, 0.3, s).astype(np.float32)).to(DEV)
    cw1, cb1, cw2, cb2 = f32(4, 64), f32(4), f32(64, 4), f32(64)
    act = lambda: torch.randn(N, H, W, 64, device=DEV).to(BF16)
    x, t1, t2, out, dy, dt1, dt2, dx = (act() for _ in range(8))
    mean, hid, gate, dz = torch.zeros(N, 64, device=DEV), torch.zeros(N, 4, device=DEV), torch.zeros(N, 64, device=DEV), torch.zeros(N, 64, device=DEV)
    xchg = torch.zeros(int(L.lib().rumpy_rcab_xchg_bytes(N, H, W)), dtype=torch.uint8, device=DEV)
    nst = N * ((H + 5) // 6)
    epoch, status = torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(16 + nst * 8 * 16 * 2, dtype=torch.int32, device=DEV)   # + stamps of a RCAB_ABL=9 build
    common = dict(N=N, H=H, W=W, cr=4, ca_w1=cw1.data_ptr(), ca_b1=cb1.data_ptr(), ca_w2=cw2.data_ptr(), ca_b2=cb2.data_ptr(), hidden=hid.data_ptr(),
                  gate=gate.data_ptr(), xchg=xchg.data_ptr(), xchg_bytes=xchg.numel(), epoch=epoch.data_ptr(), status=status.data_ptr())
    fwd = L.RcabArgs(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                     t=t1.data_ptr(), t2=t2.data_ptr(), out=out.data_ptr(), mean=mean.data_ptr(), seq=0, **common)
    bwd = L.RcabArgs(x=dy.data_ptr(), w1=pb.w_dgrad.data_ptr(), w2=pa.w_dgrad.data_ptr(), t=dt1.data_ptr(), t2=dt2.data_ptr(), t2_in=t2.data_ptr(),
                     mask=t1.data_ptr(), out=dx.data_ptr(), dz=dz.data_ptr(), seq=1, **common)
    for name, fn, a, tbuf in (('fwd', 'rumpy_rcab_fwd', fwd, t1), ('bwd', 'rumpy_rcab_bwd', bwd, dt1)):
        def run():
            L.check(L.lib().rumpy_rcab_epoch_advance(epoch.data_ptr(), stream()), 'epoch')
            for _ in range(reps):
                L.call(fn, a, stream())
        us = time_fn(run, iters=5, warm=2) / reps
        print('rcab %s %dx%dx%d: %.2f us per launch, status %d' % (name, N, H, W, us, int(status[0].item())))
        if 'RCAB_ABL_9' in os.environ.get('RUMPY_AMD_LIB', ''):
            raw = status[16:].cpu().numpy().view(np.uint64).reshape(nst, 8, 16)[:, :, :10].astype(np.float64)
            k = int((raw[0, 0] > 0).sum())
            rel = (raw[:, :, :k] - raw[:, :, :1]) * 0.01          # s_memrealtime: 100 MHz
            print('   stamps, us from each wave\'s start: rh=0: ' + ' '.join('%.2f' % v for v in rel[:, :4].mean((0, 1))) + ' | rh=1: ' + ' '.join('%.2f' % v for v in rel[:, 4:].mean((0, 1))))
            print('   wave start spread %.2f us; last end - first start %.2f us' % ((raw[:, :, 0].max() - raw[:, :, 0].min()) * 0.01, (raw[:, :, k - 1].max() - raw[:, :, 0].min()) * 0.01))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'rcab':
    rcab_bench()


def up_bench():
    """the upsampler convs (64 -> 256, PixelShuffle fused into the store) on the headline shapes, against the same conv with a plain [N,H,W,256] output"""
    gen = np.random.default_rng(0)
    pc = PackedConv(torch.from_numpy(gen.uniform(-0.04, 0.04, (256, 64, 3, 3)).astype(np.float32)), torch.zeros(256), shuffle=True)
    stamped = 'UP_ABL_9' in os.environ.get('RUMPY_AMD_LIB', '')
    if stamped:        # the stamp build writes [workgroup][wave][8] u64 behind the 256 bias values
        big = torch.zeros(256 + 2 * 256 * 8 * 16, dtype=torch.float32, device=DEV)
        big[:256] = pc.b_packed
        pc.b_packed = big
    for (N, H, W) in ((32, 48, 48), (32, 96, 96)):
        x = torch.randn(N, H, W, 64, device=DEV).to(BF16)
        out = torch.empty(N, 2 * H, 2 * W, 64, dtype=BF16, device=DEV)
        flop = 2.0 * N * H * W * 256 * 576
        for mode in (1, 0):
            a = L.ConvArgs(x=x.data_ptr(), w=pc.w_fwd.data_ptr(), bias=pc.b_packed.data_ptr(), out=out.data_ptr(), N=N, H=H, W=W, cin_chunks=1, cout_tiles=4,
                           in_mode=0, out_mode=mode, relu=0, scale=1.0, grid_x=0)
            us = time_fn(lambda: L.call('rumpy_conv3x3', a, stream()), iters=20)
            print('conv 64->256 %s %dx%dx%d: %7.2f us  %6.1f TFLOP/s  output %.2f TB/s' % ('+ PixelShuffle' if mode else 'plain output ', N, H, W, us, flop / us / 1e6, out.numel() * 2 / us / 1e6))
            if stamped and mode == 1 and H == 96:
                torch.cuda.synchronize()
                raw = pc.b_packed[256:].cpu().numpy().view(np.uint64).reshape(-1, 8, 16)[:, :, :10].astype(np.float64)
                raw = raw[raw[:, 0, 0] > 0]
                rel = (raw - raw[:, :, :1]) * 0.01
                print('   iteration 5, us from its start: rh=0: ' + ' '.join('%.2f' % v for v in rel[:, :4, :10].mean((0, 1))) + ' | rh=1: ' + ' '.join('%.2f' % v for v in rel[:, 4:, :10].mean((0, 1))))
                print('   rh=1 starts %.2f us after rh=0' % ((raw[:, 4:, 0].mean(1) - raw[:, :4, 0].mean(1)).mean() * 0.01))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'up':
    up_bench()


def up1_bench():
    """write-pattern probe: conv_up_kernel on a 64 -> 64 conv (ONE output tile: contiguous lines) at 32x192x192 - the output volume of the second upsampler conv"""
    gen = np.random.default_rng(0)
    pc = PackedConv(torch.from_numpy(gen.uniform(-0.04, 0.04, (64, 64, 3, 3)).astype(np.float32)), torch.zeros(64))
    N, H, W = 32, 192, 192
    x = torch.randn(N, H, W, 64, device=DEV).to(BF16)
    out = torch.empty(N, H, W, 64, dtype=BF16, device=DEV)
    a = L.ConvArgs(x=x.data_ptr(), w=pc.w_fwd.data_ptr(), bias=pc.b_packed.data_ptr(), out=out.data_ptr(), N=N, H=H, W=W, cin_chunks=1, cout_tiles=1,
                   in_mode=0, out_mode=0, relu=0, scale=1.0, grid_x=0)
    for env in ('0', '1'):
        os.environ['RUMPY_UP_FORCE'] = env
        us = time_fn(lambda: L.call('rumpy_conv3x3', a, stream()), iters=10)
        print('conv 64->64 32x192x192 (%s): %7.2f us  %6.1f TFLOP/s  output %.2f TB/s' % ('conv_up_kernel' if env == '1' else 'strip kernel', us, 2.0 * N * H * W * 64 * 576 / us / 1e6, out.numel() * 2 / us / 1e6))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'up1':
    up1_bench()


def tail_bench(N=32, H=192, W=192):
    """tail conv 64 -> 3 (+ L1 loss, sign gradient, its own weight gradient) at the headline output size: which part costs what"""
    gen = np.random.default_rng(0)
    w = torch.from_numpy(gen.uniform(-0.04, 0.04, (3, 64, 3, 3)).astype(np.float32))
    pt = PackedConv(w, torch.zeros(3), kind=2)
    x = torch.randn(N, H, W, 64, device=DEV).to(BF16)
    y = torch.rand(N, 3, H, W, device=DEV)
    out = torch.empty(N, 3, H, W, device=DEV)
    dy4 = torch.empty(N, H, W, 4, dtype=BF16, device=DEV)
    grid = int(L.lib().rumpy_tail_fwd_grid(N, H, W, 0))
    part = torch.zeros(grid, device=DEV)
    loss = torch.zeros(1, device=DEV)
    wslab = torch.zeros(grid * (16 * 576 + 16), device=DEV)
    base = dict(x=x.data_ptr(), w=pt.w_fwd.data_ptr(), bias=pt.b.data_ptr(), out=out.data_ptr(), N=N, C=3, H=H, W=W, grid_x=0)
    for lab, extra in (('forward only', {}), ('+ L1 loss, sign gradient', dict(target=y.data_ptr(), dy4=dy4.data_ptr(), loss_partial=part.data_ptr(), loss=loss.data_ptr())),
                       ('+ weight gradient', dict(target=y.data_ptr(), dy4=dy4.data_ptr(), loss_partial=part.data_ptr(), loss=loss.data_ptr(), wslab=wslab.data_ptr()))):
        a = L.TailFwdArgs(**base, **extra)
        us = time_fn(lambda: L.call('rumpy_tail_fwd', a, stream()), iters=20)
        print('tail %dx%dx%d %-28s %7.2f us  %5.2f TB/s of input' % (N, H, W, lab, us, x.numel() * 2 / us / 1e6))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'tail':
    tail_bench()


def rcab2_bench(N=32, H=48, W=48, reps=40):
    """conv_rcab2.hip (gate applied by the consuming launch) as the dependent chains a network runs: forward launch k reads (x, u) of launch k - 1
    and writes its own; backward launch k reads the gradient and the product rows launch k + 1 wrote.  Same shapes and stores as `rcab`."""
    gen = np.random.default_rng(0)
    mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.04, 0.04, (64, 64, 3, 3)).astype(np.float32)), torch.zeros(64))
    pa, pb = mk(), mk()
    f32 = lambda *s: torch.from_numpy(gen.uniform(-0.3, 0.3, s).astype(np.float32)).to(DEV)
    cw1, cb1, cw2, cb2 = f32(4, 64), f32(4), f32(64, 4), f32(64)
    act = lambda: (torch.randn(N, H, W, 64, device=DEV) * 0.1).to(BF16)
    X, U, G = [act(), act()], [act(), act()], [act(), act()]
    t1, dt1, dt2, uprev = act(), act(), act(), act()
    mb = torch.zeros(N, H, W, 8, dtype=torch.uint8, device=DEV)
    npr = int(L.lib().rumpy_rcab2_partials(N, H, W))
    P = [torch.zeros(N, npr, 64, device=DEV), torch.zeros(N, npr, 64, device=DEV)]
    mean, hid, gate, dz, scratch = torch.zeros(N, 64, device=DEV), torch.ones(N, 4, device=DEV), torch.full((N, 64), 0.5, device=DEV), torch.zeros(N, 64, device=DEV), torch.zeros(N, 64, device=DEV)
    ca = dict(ca_w1=cw1.data_ptr(), ca_b1=cb1.data_ptr(), ca_w2=cw2.data_ptr(), ca_b2=cb2.data_ptr(), cr=4, N=N, H=H, W=W, part_scratch=scratch.data_ptr())
    fwd = [L.Rcab2Args(x=X[k].data_ptr(), u_in=U[k].data_ptr(), part_in=P[k].data_ptr(), np_in=npr, part_out=P[1 - k].data_ptr(), w1=pa.w_fwd.data_ptr(),
                       b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(), x_out=X[1 - k].data_ptr(), t=t1.data_ptr(), u_out=U[1 - k].data_ptr(),
                       maskbits=mb.data_ptr(), mean=mean.data_ptr(), hidden=hid.data_ptr(), gate=gate.data_ptr(), **ca) for k in (0, 1)]
    bwd = [L.Rcab2Args(x=G[k].data_ptr(), u_in=uprev.data_ptr(), part_in=P[k].data_ptr(), np_in=npr, part_out=P[1 - k].data_ptr(), w1=pb.w_dgrad.data_ptr(),
                       w2=pa.w_dgrad.data_ptr(), x_out=dt2.data_ptr(), t=dt1.data_ptr(), u_out=G[1 - k].data_ptr(), maskbits=mb.data_ptr(),
                       hidden=hid.data_ptr(), gate=gate.data_ptr(), dz=dz.data_ptr(), **ca) for k in (0, 1)]
    for name, fn, pair in (('fwd', 'rumpy_rcab2_fwd', fwd), ('bwd', 'rumpy_rcab2_bwd', bwd)):
        if globals().get('_R2_ONLY') not in (None, name):
            continue

        def run():
            for i in range(reps):
                L.call(fn, pair[i & 1], stream())
        us = time_fn(run, iters=5, warm=2) / reps
        print('rcab2 %s %dx%dx%d: %.2f us per launch (chain of dependent launches)' % (name, N, H, W, us))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'rcab2':
    rcab2_bench()


def rcab2_stamps(N=32, H=48, W=48):
    """phase stamps of the conv_rcab2.hip launches (a -DRCAB2_STAMPS build: RUMPY_AMD_LIB=build_abl/R2_STAMPS/librumpy_amd.so)"""
    import ctypes
    lib = L.lib()
    fn = getattr(ctypes.CDLL(os.environ['RUMPY_AMD_LIB']), 'rumpy_debug_rcab2_stamps')
    fn.argtypes = [ctypes.c_void_p]
    nwg = N * ((H + 5) // 6)
    buf = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device=DEV)
    assert fn(buf.data_ptr()) == 0
    rcab2_bench(N, H, W, reps=3)
    names = ['start', 'gate ready', 'tile merged', 'barrier', 'sweep 1', 'T in LDS', 'sweep 2', 'OUT in LDS', 'stores issued', 'product rows']
    # the buffer holds the LAST launch of the bench: a backward launch; run one forward chain again for its stamps
    for which in ('bwd', 'fwd'):
        if which == 'fwd':
            buf.zero_()
            rcab2_bench_one(N, H, W, 'fwd')
        torch.cuda.synchronize()
        raw = buf.cpu().numpy().reshape(nwg, 8, 16).astype(np.float64)
        k = int((raw[0, 0] > 0).sum())
        rel = (raw[:, :, :k] - raw[:, :, :1]) * 0.01
        print('rcab2 %s stamps, us from each wave\'s start (%s):' % (which, ', '.join(names[:k])))
        print('   rh=0: ' + ' '.join('%.2f' % v for v in rel[:, :4].mean((0, 1))) + ' | rh=1: ' + ' '.join('%.2f' % v for v in rel[:, 4:].mean((0, 1))))
        print('   wave start spread %.2f us; last end - first start %.2f us' % ((raw[:, :, 0].max() - raw[:, :, 0].min()) * 0.01, (raw[:, :, k - 1].max() - raw[:, :, 0].min()) * 0.01))


def rcab2_bench_one(N, H, W, which):
    global _R2_ONLY
    _R2_ONLY = which
    try:
        rcab2_bench(N, H, W, reps=4)
    finally:
        _R2_ONLY = None


_R2_ONLY = None

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'rcab2stamps':
    rcab2_stamps()


def tailfuse_bench(N=32, H=96, W=96):
    """rumpy_tail_dgrad + rumpy_conv3x3(in_mode 1) against rumpy_conv4d_tail (round 5) on the last upsampler stage of the x4 step"""
    gen = np.random.default_rng(0)
    wt = torch.from_numpy(gen.uniform(-0.04, 0.04, (3, 64, 3, 3)).astype(np.float32))
    pt = PackedConv(wt, torch.zeros(3), 2)
    wu = torch.from_numpy(gen.uniform(-0.04, 0.04, (256, 64, 3, 3)).astype(np.float32))
    pu = PackedConv(wu, torch.zeros(256), 0, True)
    g4 = torch.sign(torch.randn(N, 2 * H, 2 * W, 4, device=DEV)).to(BF16)
    g4[..., 3] = 0
    dx = torch.empty(N, 2 * H, 2 * W, 64, dtype=BF16, device=DEV)
    out = torch.empty(N, H, W, 64, dtype=BF16, device=DEV)
    a_t = L.TailDgradArgs(dy4=g4.data_ptr(), w=pt.w_dgrad.data_ptr(), dx=dx.data_ptr(), N=N, H=2 * H, W=2 * W)
    a_c = L.ConvArgs(x=dx.data_ptr(), w=pu.w_dgrad.data_ptr(), bias=None, out=out.data_ptr(), N=N, H=H, W=W, cin_chunks=4, cout_tiles=1,
                     in_mode=1, out_mode=0, relu=0, scale=1.0, grid_x=0)
    a_f = L.Conv4dTailArgs(dy4=g4.data_ptr(), w_tail=pt.w_dgrad.data_ptr(), dx=dx.data_ptr(), w=pu.w_dgrad.data_ptr(), out=out.data_ptr(),
                           N=N, H=H, W=W, grid_x=0)
    for rep in range(3):
        t1 = time_fn(lambda: L.call('rumpy_tail_dgrad', a_t, stream()), iters=100)
        t2 = time_fn(lambda: L.call('rumpy_conv3x3', a_c, stream()), iters=100)

        def both():
            L.call('rumpy_tail_dgrad', a_t, stream())
            L.call('rumpy_conv3x3', a_c, stream())
        t12 = time_fn(both, iters=100)
        t3 = time_fn(lambda: L.call('rumpy_conv4d_tail', a_f, stream()), iters=100)
        print('%dx%dx%d: tail_dgrad %6.1f us + conv4d %6.1f us (back to back %6.1f) | conv4d_tail %6.1f us' % (N, H, W, t1, t2, t12, t3))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'tailfuse':
    tailfuse_bench()
    tailfuse_bench(8, 128, 128)
    tailfuse_bench(32, 48, 48)

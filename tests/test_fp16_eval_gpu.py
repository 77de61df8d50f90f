"""-m gpu: the IEEE-fp16 instantiations of the kernels an EVALUATION plan launches (include/rumpy_amd.h RUMPY_FMT_F16), each through
the C ABI against torch fp32 on fp16-rounded operands.  One fp16 rounding of the result: relative Frobenius error about
2^-12 / sqrt(3) = 1.4e-4 (bf16: 1.1e-3), so the bounds here are 8x tighter than in tests/test_kernels_gpu.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from rumpy_amd import _lib as L
try:
    from tests.gpu_utils import DEV, pack_ref, rel_err, stream, to_dev_bytes
except ImportError:
    from gpu_utils import DEV, pack_ref, rel_err, stream, to_dev_bytes

F16 = torch.float16


def f16r(t):
    return t.to(F16).float()


def nhwc16(t):
    return t.permute(0, 2, 3, 1).contiguous().to(F16).to(DEV)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def close16(got, ref, what, rel=6e-4, amax=2.0 ** -9):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all(), what
    r, m, scale = rel_err(got, ref), float((got - ref).abs().max()), float(ref.abs().max()) + 1e-30
    assert r < rel and m <= amax * scale, '%s: relative error %.3e (limit %.1e), max abs %.3e / %.3e' % (what, r, rel, m, scale)


class Packed16:
    def __init__(self, w, b, kind=0, shuffle=False):
        self.w, self.b = w.float().contiguous().to(DEV), b.float().contiguous().to(DEV)
        self.cout, self.cin = w.shape[0], w.shape[1]
        self.w_fwd = torch.zeros(self.cout * self.cin * 9 if kind == 0 else 2 * 18 * 64 * 8, dtype=F16, device=DEV)       # tail: filter + rounding-residual image
        self.b_packed = torch.zeros(self.cout, dtype=torch.float32, device=DEV) if kind == 0 else None
        it = L.PackItem(w=self.w.data_ptr(), b=self.b.data_ptr(), w_fwd=self.w_fwd.data_ptr(), w_dgrad=None,
                        b_packed=self.b_packed.data_ptr() if kind == 0 else None, cout=self.cout, cin=self.cin, kind=kind,
                        shuffle=1 if shuffle else 0, fmt=L.FMT_F16)
        self.items = to_dev_bytes((L.PackItem * 1)(it))
        L.check(L.lib().rumpy_pack_weights(self.items.data_ptr(), 1, stream()), 'pack')
        torch.cuda.synchronize()


def _wb(gen, co, ci):
    b = 1.0 / np.sqrt(ci * 9)
    return (torch.from_numpy(gen.uniform(-b, b, (co, ci, 3, 3)).astype(np.float32)), torch.from_numpy(gen.uniform(-b, b, (co,)).astype(np.float32)))


def conv16(x, pc, N, H, W, relu=False, scale=1.0, res1=None, res2=None, pool=False, out_mode=0, grid_x=0):
    ct = pc.cout // 64
    out = torch.full((N, 2 * H, 2 * W, 64) if out_mode else (N, H, W, 64 * ct), float('nan'), dtype=F16, device=DEV)
    tiles = int(L.lib().rumpy_conv_pool_tiles(H, W, 1))
    pl = torch.full((N, tiles, 64 * ct), float('nan'), dtype=torch.float32, device=DEV) if pool else None
    p = lambda t: None if t is None else t.data_ptr()
    L.call('rumpy_conv3x3', L.ConvArgs(x=x.data_ptr(), w=pc.w_fwd.data_ptr(), bias=pc.b_packed.data_ptr(), out=out.data_ptr(), res1=p(res1), res2=p(res2),
                                       pool=p(pl), N=N, H=H, W=W, cin_chunks=1, cout_tiles=ct, out_mode=out_mode, relu=1 if relu else 0,
                                       scale=float(scale), grid_x=grid_x, fmt=L.FMT_F16), stream())
    torch.cuda.synchronize()
    return out, pl


def test_fp16_pack_layout_and_forward_only_formats():
    gen = np.random.default_rng(1)
    w, b = _wb(gen, 256, 64)
    pc = Packed16(w, b, 0, True)
    fwd, _ = pack_ref(w.numpy(), True)
    assert torch.equal(pc.w_fwd.float().cpu(), f16r(torch.from_numpy(fwd.reshape(-1))))
    t = torch.zeros(4096, dtype=F16, device=DEV)
    # fp16 is a forward-only format: masks, Cin = 256 chunks and the backward entry points refuse it
    a = L.ConvArgs(x=t.data_ptr(), w=t.data_ptr(), out=t.data_ptr(), mask=t.data_ptr(), N=1, H=1, W=1, cin_chunks=1, cout_tiles=1, fmt=L.FMT_F16)
    assert L.lib().rumpy_conv3x3(a, None) == -1 and b'fmt' in L.lib().rumpy_last_error()
    a = L.ConvArgs(x=t.data_ptr(), w=t.data_ptr(), out=t.data_ptr(), N=1, H=1, W=1, cin_chunks=1, cout_tiles=1, fmt=7)
    assert L.lib().rumpy_conv3x3(a, None) == -1
    a = L.BlockArgs(x=t.data_ptr(), w1=t.data_ptr(), w2=t.data_ptr(), out=t.data_ptr(), mask=t.data_ptr(), N=1, H=1, W=1, relu1=0, scale1=1.0, scale2=1.0, fmt=L.FMT_F16)
    assert L.lib().rumpy_conv_block(a, None) == -1 and b'fmt' in L.lib().rumpy_last_error()


@pytest.mark.parametrize('N,H,W', [(2, 12, 12), (1, 48, 48), (3, 10, 21), (2, 5, 3), (1, 30, 100)])
def test_fp16_conv3x3_forward_epilogues(N, H, W):
    gen = np.random.default_rng(10 + H)
    w, b = _wb(gen, 64, 64)
    x = torch.from_numpy(gen.standard_normal((N, 64, H, W)).astype(np.float32))
    r1 = torch.from_numpy(gen.standard_normal((N, 64, H, W)).astype(np.float32))
    r2 = torch.from_numpy(gen.standard_normal((N, 64, H, W)).astype(np.float32))
    pc = Packed16(w, b)
    xd = nhwc16(x)
    ref = F.conv2d(f16r(x), f16r(w), b, padding=1)
    out, _ = conv16(xd, pc, N, H, W)
    close16(nchw(out), ref, 'fp16 conv')
    out, pl = conv16(xd, pc, N, H, W, relu=True, scale=0.5, res1=nhwc16(r1), res2=nhwc16(r2), pool=True)
    v = F.relu(ref) * 0.5
    close16(nchw(out), v + f16r(r1) + f16r(r2), 'fp16 conv relu/scale/residuals')
    assert rel_err(pl.sum(1).cpu(), v.sum(dim=(2, 3))) < 1e-5
    for gx in (1, 3):                  # persistent launches: same bits
        o2, _ = conv16(xd, pc, N, H, W, relu=True, scale=0.5, res1=nhwc16(r1), res2=nhwc16(r2), grid_x=gx)
        assert torch.equal(o2, out)


def test_fp16_upsampler_conv_pixel_shuffle_and_head_and_tail():
    gen = np.random.default_rng(12)
    N, H, W = 2, 9, 14
    w, b = _wb(gen, 256, 64)
    x = torch.from_numpy(gen.standard_normal((N, 64, H, W)).astype(np.float32))
    out, _ = conv16(nhwc16(x), Packed16(w, b, 0, True), N, H, W, out_mode=1)
    close16(nchw(out), F.pixel_shuffle(F.conv2d(f16r(x), f16r(w), b, padding=1), 2), 'fp16 upsampler conv')
    # head: exact fp32 arithmetic, fp16 store
    hw_, hb_ = torch.from_numpy(gen.uniform(-0.2, 0.2, (64, 3, 3, 3)).astype(np.float32)), torch.from_numpy(gen.uniform(-0.2, 0.2, (64,)).astype(np.float32))
    img = torch.from_numpy(gen.random((N, 3, H, W), dtype=np.float32))
    o = torch.full((N, H, W, 64), float('nan'), dtype=F16, device=DEV)
    imd, hwd, hbd = img.to(DEV), hw_.to(DEV), hb_.to(DEV)
    L.call('rumpy_head_fwd', L.HeadFwdArgs(x=imd.data_ptr(), w=hwd.data_ptr(), b=hbd.data_ptr(), out=o.data_ptr(), N=N, C=3, H=H, W=W, cout=64, fmt=L.FMT_F16), stream())
    torch.cuda.synchronize()
    close16(nchw(o), F.conv2d(img, hw_, hb_, padding=1), 'fp16 head')
    # tail: fp16 operands, fp32 NCHW out, fused L1 loss value (evaluation with a loss request), non-finite flag
    tw, tb = torch.from_numpy(gen.uniform(-0.04, 0.04, (3, 64, 3, 3)).astype(np.float32)), torch.from_numpy(gen.uniform(-0.04, 0.04, (3,)).astype(np.float32))
    pt = Packed16(tw, tb, 2)
    y = torch.from_numpy(gen.random((N, 3, H, W), dtype=np.float32)).to(DEV)
    res = torch.full((N, 3, H, W), float('nan'), device=DEV)
    part, loss, flag = torch.zeros(2048, device=DEV), torch.zeros(1, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    xd = nhwc16(x)
    a = L.TailFwdArgs(x=xd.data_ptr(), w=pt.w_fwd.data_ptr(), bias=pt.b.data_ptr(), out=res.data_ptr(), target=y.data_ptr(), loss_partial=part.data_ptr(),
                      loss=loss.data_ptr(), N=N, C=3, H=H, W=W, nonfinite=flag.data_ptr(), fmt=L.FMT_F16)
    L.call('rumpy_tail_fwd', a, stream())
    torch.cuda.synchronize()
    # the filter enters as fp16 image + fp16 image of its rounding residual: 22 significant bits - the reference is the UNROUNDED filter
    ref = F.conv2d(f16r(x).double(), tw.double(), tb.double(), padding=1).float()
    assert rel_err(res.cpu(), ref) < 2e-6 and int(flag.item()) == 0
    assert rel_err(res.cpu(), F.conv2d(f16r(x), f16r(tw), tb, padding=1)) > 1e-4          # and visibly not the fp16-rounded one
    assert abs(float(loss.item()) - float((res - y).abs().mean())) < 1e-6
    xd[1, 3, 5, 7] = float('inf')
    L.call('rumpy_tail_fwd', a, stream())
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    a.dy4 = xd.data_ptr()              # training outputs are bf16 only
    assert L.lib().rumpy_tail_fwd(a, None) == -1 and b'fmt' in L.lib().rumpy_last_error()


@pytest.mark.parametrize('N,H,W,shuffle', [(2, 9, 14, True), (1, 20, 37, True), (3, 13, 60, True), (1, 7, 50, False)])
def test_fp16_upsampler_conv_with_both_filter_images_in_one_launch(N, H, W, shuffle):
    """rumpy_conv_args.w_lo (round 3): an upsampler stage of an evaluation plan sweeps the filter's rounding-residual image and the fp16 filter in
    ONE launch, same fp32 accumulators.  The result is the conv with the UNROUNDED filter rounded once to fp16 (checked against float64), is at
    least as close to it as round 2's two launches through a scratch tensor (residual launch, then main launch with res1), and differs visibly
    from the conv with the fp16-rounded filter alone."""
    gen = np.random.default_rng(40 + H + W)
    w, b = _wb(gen, 256, 64)
    x = torch.from_numpy(gen.standard_normal((N, 64, H, W)).astype(np.float32))
    xd = nhwc16(x)
    pc = Packed16(w, b, 0, shuffle)
    lo = torch.zeros(256 * 64 * 9, dtype=F16, device=DEV)
    it = L.PackItem(w=pc.w.data_ptr(), b=pc.b.data_ptr(), w_fwd=lo.data_ptr(), w_dgrad=None, b_packed=None, cout=256, cin=64, kind=0,
                    shuffle=1 if shuffle else 0, fmt=L.FMT_F16_RESIDUAL)
    items = to_dev_bytes((L.PackItem * 1)(it))
    L.check(L.lib().rumpy_pack_weights(items.data_ptr(), 1, stream()), 'pack residual image')
    om = 1 if shuffle else 0
    shape = (N, 2 * H, 2 * W, 64) if shuffle else (N, H, W, 256)

    def launch(**kw):
        out = torch.full(shape, float('nan'), dtype=F16, device=DEV)
        a = L.ConvArgs(x=xd.data_ptr(), w=pc.w_fwd.data_ptr(), bias=pc.b_packed.data_ptr(), out=out.data_ptr(), N=N, H=H, W=W, cin_chunks=1, cout_tiles=4,
                       out_mode=om, relu=0, scale=1.0, grid_x=0, fmt=L.FMT_F16)
        for k, v in kw.items():
            setattr(a, k, v)
        L.call('rumpy_conv3x3', a, stream())
        torch.cuda.synchronize()
        return out
    one = launch(w_lo=lo.data_ptr())
    if shuffle:
        sc = torch.full((N, H, W, 256), float('nan'), dtype=F16, device=DEV)
        a = L.ConvArgs(x=xd.data_ptr(), w=lo.data_ptr(), bias=None, out=sc.data_ptr(), N=N, H=H, W=W, cin_chunks=1, cout_tiles=4, out_mode=0, relu=0,
                       scale=1.0, grid_x=0, fmt=L.FMT_F16)
        L.call('rumpy_conv3x3', a, stream())
        two = launch(res1=sc.data_ptr())
    plain = launch()
    ref = F.conv2d(f16r(x).double(), w.double(), b.double(), padding=1)
    ref = F.pixel_shuffle(ref, 2) if shuffle else ref
    got = lambda t: (nchw(t) if shuffle else nchw(t)).double()
    e_one, e_plain = rel_err(got(one), ref), rel_err(got(plain), ref)
    e_round = rel_err(ref.to(F16).double(), ref)                # what ONE fp16 rounding of the exact result costs
    print('one launch %.3e, fp16 filter alone %.3e, rounding of the exact result %.3e' % (e_one, e_plain, e_round))
    assert e_one < 1.05 * e_round, (e_one, e_round)
    assert e_plain > 1.3 * e_one                                # the fp16-rounded filter alone is visibly further away
    if shuffle:
        e_two = rel_err(got(two), ref)
        assert e_one <= e_two * 1.02, (e_one, e_two)
        assert rel_err(got(one), got(two)) < 1.5e-4
    bad = L.ConvArgs(x=xd.data_ptr(), w=pc.w_fwd.data_ptr(), bias=None, out=one.data_ptr(), N=N, H=H, W=W, cin_chunks=1, cout_tiles=4, out_mode=om,
                     relu=0, scale=1.0, grid_x=0, fmt=L.FMT_BF16, w_lo=lo.data_ptr())
    assert L.lib().rumpy_conv3x3(bad, None) == -1 and b'w_lo' in L.lib().rumpy_last_error()


@pytest.mark.parametrize('N,H,W', [(1, 6, 16), (2, 13, 48), (3, 20, 37)])
def test_fp16_residual_block_and_rcab_launches(N, H, W):
    gen = np.random.default_rng(100 + H + W)
    (w1, b1), (w2, b2) = _wb(gen, 64, 64), _wb(gen, 64, 64)
    pa, pb = Packed16(w1, b1), Packed16(w2, b2)
    x = torch.from_numpy(gen.standard_normal((N, 64, H, W)).astype(np.float32))
    xd = nhwc16(x)
    # ResBlock forward form (conv_block.hip)
    y = torch.full((N, H, W, 64), float('nan'), dtype=F16, device=DEV)
    L.call('rumpy_conv_block', L.BlockArgs(x=xd.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                                           out=y.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=0.1, fmt=L.FMT_F16), stream())
    torch.cuda.synchronize()
    t1 = f16r(F.relu(F.conv2d(f16r(x), f16r(w1), b1, padding=1)))
    t2 = F.conv2d(t1, f16r(w2), b2, padding=1)
    close16(nchw(y), f16r(x) + 0.1 * t2, 'fp16 residual block')
    # RCAB forward (conv_rcab.hip): gate from the mean of the fp32 accumulators
    Cr = 4
    cw1, cb1 = torch.from_numpy(gen.uniform(-0.3, 0.3, (Cr, 64)).astype(np.float32)), torch.from_numpy(gen.uniform(-0.3, 0.3, (Cr,)).astype(np.float32))
    cw2, cb2 = torch.from_numpy(gen.uniform(-0.3, 0.3, (64, Cr)).astype(np.float32)), torch.from_numpy(gen.uniform(-0.3, 0.3, (64,)).astype(np.float32))
    d = lambda t: t.contiguous().to(DEV)
    cw1d, cb1d, cw2d, cb2d = d(cw1), d(cb1), d(cw2), d(cb2)
    mean, hid, gate = (torch.zeros(N, k, device=DEV) for k in (64, Cr, 64))
    xchg = torch.zeros(int(L.lib().rumpy_rcab_xchg_bytes(N, H, W)), dtype=torch.uint8, device=DEV)
    epoch, status = torch.ones(1, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    out = torch.full((N, H, W, 64), float('nan'), dtype=F16, device=DEV)
    L.call('rumpy_rcab_fwd', L.RcabArgs(x=xd.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                                        out=out.data_ptr(), N=N, H=H, W=W, cr=Cr, ca_w1=cw1d.data_ptr(), ca_b1=cb1d.data_ptr(), ca_w2=cw2d.data_ptr(),
                                        ca_b2=cb2d.data_ptr(), mean=mean.data_ptr(), hidden=hid.data_ptr(), gate=gate.data_ptr(), xchg=xchg.data_ptr(),
                                        xchg_bytes=xchg.numel(), epoch=epoch.data_ptr(), status=status.data_ptr(), seq=0, fmt=L.FMT_F16), stream())
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    m = t2.mean(dim=(2, 3))
    g_ref = torch.sigmoid(F.relu(m @ cw1.t() + cb1) @ cw2.t() + cb2)
    assert rel_err(gate.cpu(), g_ref) < 1e-4
    close16(nchw(out), f16r(x) + t2 * g_ref[:, :, None, None], 'fp16 RCAB forward')
    a = L.RcabArgs(x=xd.data_ptr(), w1=pa.w_fwd.data_ptr(), w2=pb.w_fwd.data_ptr(), out=out.data_ptr(), N=N, H=H, W=W, cr=Cr, ca_w1=cw1d.data_ptr(),
                   ca_b1=cb1d.data_ptr(), ca_w2=cw2d.data_ptr(), ca_b2=cb2d.data_ptr(), hidden=hid.data_ptr(), gate=gate.data_ptr(), xchg=xchg.data_ptr(),
                   xchg_bytes=xchg.numel(), epoch=epoch.data_ptr(), status=status.data_ptr(), t2_in=xd.data_ptr(), mask=xd.data_ptr(), t2=out.data_ptr(),
                   dz=mean.data_ptr(), seq=1, fmt=L.FMT_F16)
    assert L.lib().rumpy_rcab_bwd(a, None) == -1 and b'forward-only' in L.lib().rumpy_last_error()


def test_fp16_channel_attention_fused_forward():
    gen = np.random.default_rng(21)
    N, H, W, Cc, Cr = 2, 11, 70, 64, 4
    t2 = torch.from_numpy(gen.standard_normal((N, Cc, H, W)).astype(np.float32))
    xr = torch.from_numpy(gen.standard_normal((N, Cc, H, W)).astype(np.float32))
    w1, b1 = torch.from_numpy(gen.uniform(-0.3, 0.3, (Cr, Cc)).astype(np.float32)), torch.from_numpy(gen.uniform(-0.3, 0.3, (Cr,)).astype(np.float32))
    w2, b2 = torch.from_numpy(gen.uniform(-0.3, 0.3, (Cc, Cr)).astype(np.float32)), torch.from_numpy(gen.uniform(-0.3, 0.3, (Cc,)).astype(np.float32))
    pool = f16r(t2).sum(dim=(2, 3)).reshape(N, 1, Cc).contiguous().to(DEV)
    d = lambda t: t.contiguous().to(DEV)
    w1d, b1d, w2d, b2d = d(w1), d(b1), d(w2), d(b2)
    mean, hid, gate = (torch.zeros(N, k, device=DEV) for k in (Cc, Cr, Cc))
    y = torch.full((N, H, W, Cc), float('nan'), dtype=F16, device=DEV)
    td, xd = nhwc16(t2), nhwc16(xr)
    L.call('rumpy_ca_fwd_fused', L.CaFwdFusedArgs(pool=pool.data_ptr(), w1=w1d.data_ptr(), b1=b1d.data_ptr(), w2=w2d.data_ptr(), b2=b2d.data_ptr(),
                                                  mean=mean.data_ptr(), hidden=hid.data_ptr(), gate=gate.data_ptr(), t=td.data_ptr(), res=xd.data_ptr(),
                                                  out=y.data_ptr(), N=N, HW=H * W, C=Cc, Cr=Cr, ntiles=1, inv_hw=1.0 / (H * W), fmt=L.FMT_F16), stream())
    torch.cuda.synchronize()
    g_ref = torch.sigmoid(F.relu(f16r(t2).mean(dim=(2, 3)) @ w1.t() + b1) @ w2.t() + b2)
    close16(nchw(y), f16r(xr) + f16r(t2) * g_ref[:, :, None, None], 'fp16 CA fused forward')

"""-m gpu parity tests of the individual C-ABI kernels against the CPU oracle arithmetic (torch fp32 on the same
bf16-rounded operands).  Tolerances: outputs stored as bf16 -> one bf16 rounding (relative Frobenius error < 4e-3,
max error < 2^-6 of the tensor's max); fp32 outputs from bf16 operands -> summation order only (< 2e-3 relative,
typically 1e-6)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

try:
    from tests.gpu_utils import (BF16, DEV, BlockChainArgs, ChainArgs, ChainLayer, PackedConv, assert_bf16_close, assert_f32_close, bf16r,
                                 exp_call, exp_lib, hip_conv, hip_wgrad, nchw, nhwc, pack_bias_ref, pack_ref, stream, to_dev_bytes)
except ImportError:  # pytest rootdir import mode
    from gpu_utils import (BF16, DEV, BlockChainArgs, ChainArgs, ChainLayer, PackedConv, assert_bf16_close, assert_f32_close, bf16r,
                           exp_call, exp_lib, hip_conv, hip_wgrad, nchw, nhwc, pack_bias_ref, pack_ref, stream, to_dev_bytes)
from rumpy_amd import _lib as L


def _rand(gen, *shape, scale=1.0):
    return torch.from_numpy(gen.standard_normal(shape).astype(np.float32) * scale)


def _wb(gen, co, ci):
    b = 1.0 / np.sqrt(ci * 9)
    return (torch.from_numpy(gen.uniform(-b, b, (co, ci, 3, 3)).astype(np.float32)),
            torch.from_numpy(gen.uniform(-b, b, (co,)).astype(np.float32)))


@pytest.mark.parametrize('co,ci,shuffle', [(64, 64, False), (256, 64, True), (64, 256, False)])
def test_pack_weights_layout(co, ci, shuffle):
    gen = np.random.default_rng(1)
    w, b = _wb(gen, co, ci)
    pc = PackedConv(w, b, 0, shuffle)
    fwd, dgr = pack_ref(w.numpy(), shuffle)
    assert torch.equal(pc.w_fwd.float().cpu(), bf16r(torch.from_numpy(fwd.reshape(-1))))
    assert torch.equal(pc.w_dgrad.float().cpu(), bf16r(torch.from_numpy(dgr.reshape(-1))))
    assert np.array_equal(pc.b_packed.cpu().numpy(), pack_bias_ref(b.numpy(), shuffle))


@pytest.mark.parametrize('N,H,W', [(2, 12, 12), (1, 48, 48), (3, 10, 21), (1, 8, 16), (2, 5, 3)])
def test_conv3x3_forward_plain(N, H, W):
    gen = np.random.default_rng(10 + H)
    w, b = _wb(gen, 64, 64)
    x = _rand(gen, N, 64, H, W)
    pc = PackedConv(w, b)
    out, _ = hip_conv(nhwc(x), pc, N, H, W)
    ref = F.conv2d(bf16r(x), bf16r(w), b, padding=1)
    assert_bf16_close(nchw(out), ref, 'conv fwd %dx%dx%d' % (N, H, W))


def test_conv3x3_forward_persistent_grid_variants():
    """every grid size must give the same bits (tiles are independent)"""
    gen = np.random.default_rng(11)
    w, b = _wb(gen, 64, 64)
    x = _rand(gen, 4, 64, 24, 40)
    pc = PackedConv(w, b)
    xd = nhwc(x)
    base, _ = hip_conv(xd, pc, 4, 24, 40)
    for gx in (1, 3, 7, 36):
        o, _ = hip_conv(xd, pc, 4, 24, 40, grid_x=gx)
        assert torch.equal(o, base), gx


def test_conv3x3_persistent_launches_with_every_epilogue_are_bitwise_the_one_strip_launch():
    """persistent launches (a workgroup runs several strips) defer the epilogue of half the waves to the next iteration (phase skew,
    DESIGN.md 4.1 (5)): mask, residuals, pool partials, ReLU + scale, and the 64 -> 256 PixelShuffle store must come out bit for bit
    as from the launch with one strip per workgroup - for one, several and a ragged number of strips per workgroup."""
    gen = np.random.default_rng(14)
    N, H, W = 3, 26, 52
    w, b = _wb(gen, 64, 64)
    x, r1, r2, m = (nhwc(_rand(gen, N, 64, H, W)) for _ in range(4))
    pc = PackedConv(w, b)
    variants = [dict(relu=True, scale=0.5), dict(scale=0.1, res1=r1, pool=True), dict(res1=r1, res2=r2),
                dict(use_bias=False, scale=0.1, mask=m, res1=r1, res2=r2), dict(mask=m, pool=True)]
    for kw in variants:
        base, bp = hip_conv(x, pc, N, H, W, grid_x=10 ** 6, **kw)          # clamped to the strip count: one strip per workgroup
        for gx in (1, 4, 7):
            o, pl = hip_conv(x, pc, N, H, W, grid_x=gx, **kw)
            assert torch.equal(o, base), (sorted(kw), gx)
            if bp is not None:
                assert torch.equal(pl, bp), (sorted(kw), gx)
    w4, b4 = _wb(gen, 256, 64)
    pc4 = PackedConv(w4, b4, 0, True)
    base, _ = hip_conv(x, pc4, N, H, W, out_mode=1, grid_x=10 ** 6)
    for gx in (1, 5):
        o, _ = hip_conv(x, pc4, N, H, W, out_mode=1, grid_x=gx)
        assert torch.equal(o, base), gx


def test_conv3x3_epilogue_relu_scale_residuals_mask_pool():
    gen = np.random.default_rng(12)
    N, H, W = 2, 20, 18
    w, b = _wb(gen, 64, 64)
    x, r1, r2, m = (_rand(gen, N, 64, H, W) for _ in range(4))
    pc = PackedConv(w, b)
    conv = F.conv2d(bf16r(x), bf16r(w), b, padding=1)
    # ResBlock conv1: relu
    o, _ = hip_conv(nhwc(x), pc, N, H, W, relu=True)
    assert_bf16_close(nchw(o), conv.clamp_min(0), 'relu')
    # ResBlock conv2: *0.1 + residual
    o, _ = hip_conv(nhwc(x), pc, N, H, W, scale=0.1, res1=nhwc(r1))
    assert_bf16_close(nchw(o), conv * 0.1 + bf16r(r1), 'scale+res')
    # two residuals
    o, _ = hip_conv(nhwc(x), pc, N, H, W, res1=nhwc(r1), res2=nhwc(r2))
    assert_bf16_close(nchw(o), conv + bf16r(r1) + bf16r(r2), 'res1+res2')
    # backward style: no bias, scale, relu mask
    o, _ = hip_conv(nhwc(x), pc, N, H, W, use_bias=False, scale=0.1, mask=nhwc(m))
    ref = F.conv2d(bf16r(x), bf16r(w), None, padding=1) * 0.1 * (bf16r(m) > 0).float()
    assert_bf16_close(nchw(o), ref, 'mask')
    # per-tile channel sums (global average pool partials), taken before the residual add
    o, pl = hip_conv(nhwc(x), pc, N, H, W, res1=nhwc(r1), pool=True)
    # strip kernel: one partial row per (6-row strip, 3-row group, 48-column strip)
    rows_n, cols_n = 2 * ((H + 5) // 6), (W + 47) // 48
    assert pl.shape[1] == rows_n * cols_n
    refp = torch.zeros(N, rows_n * cols_n, 64)
    for ry in range(rows_n):
        for tx in range(cols_n):
            refp[:, ry * cols_n + tx] = conv[:, :, ry * 3:ry * 3 + 3, tx * 48:tx * 48 + 48].sum(dim=(2, 3))
    assert_f32_close(pl, refp, 'pool partials', rel=1e-4)
    assert_bf16_close(nchw(o), conv + bf16r(r1), 'pool+res output')


def test_conv_up_kernel_is_bitwise_the_strip_kernel(monkeypatch):
    """conv_up.hip (output image in LDS -> whole non-temporal lines; what large launches without mask / pool sums use) against conv_strip.hip on
    the same launches: plain, ReLU + scale + two residual operands, several output tiles, PixelShuffle - ragged sizes, several strips per workgroup"""
    gen = np.random.default_rng(31)
    for (N, H, W, co) in ((3, 20, 100, 64), (2, 13, 37, 128), (1, 50, 49, 64)):
        w = torch.from_numpy(gen.uniform(-0.05, 0.05, (co, 64, 3, 3)).astype(np.float32))
        b = torch.from_numpy(gen.uniform(-0.1, 0.1, (co,)).astype(np.float32))
        pc = PackedConv(w, b)
        x = nhwc(_rand(gen, N, 64, H, W))
        r1, r2 = nhwc(_rand(gen, N, co, H, W)), nhwc(_rand(gen, N, co, H, W))
        outs = []
        for force in ('0', '1'):
            monkeypatch.setenv('RUMPY_UP_FORCE', force)
            if co == 128:
                monkeypatch.setenv('RUMPY_UP_OLD' if force == '0' else 'RUMPY_UP_XX', '1')       # multi-tile launches use conv_up by default
            o1, _ = hip_conv(x, pc, N, H, W, grid_x=3)
            o2, _ = hip_conv(x, pc, N, H, W, relu=True, scale=0.3, res1=r1, res2=r2, grid_x=5)
            outs.append((o1, o2))
            monkeypatch.delenv('RUMPY_UP_OLD', raising=False)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (N, H, W, co)
        ref = F.conv2d(bf16r(nchw(x)), bf16r(w), b, padding=1)
        assert_bf16_close(nchw(outs[1][0]), ref, 'conv_up plain')
    monkeypatch.delenv('RUMPY_UP_FORCE', raising=False)


def test_conv3x3_upsampler_forward_pixel_shuffle_fused():
    gen = np.random.default_rng(13)
    N, H, W = 2, 11, 19
    w, b = _wb(gen, 256, 64)
    x = _rand(gen, N, 64, H, W)
    pc = PackedConv(w, b, 0, True)
    o, _ = hip_conv(nhwc(x), pc, N, H, W, out_mode=1)
    ref = F.pixel_shuffle(F.conv2d(bf16r(x), bf16r(w), b, padding=1), 2)
    assert_bf16_close(nchw(o), ref, 'conv 64->256 + PixelShuffle(2)')


def test_conv3x3_upsampler_pixel_shuffle_with_a_residual_operand_in_conv_output_order():
    """the second launch of an fp16 evaluation plan's upsampler stage: conv + residual operand + PixelShuffle in the store.  The operand is
    indexed like the conv's own (pre-shuffle) output, channel tile = sub-pixel position: operand[n, y, x, 64 t + c] belongs to natural
    channel 4 c + t.  Several strips per workgroup, ragged sizes, one strip; and still refused: two residual operands, a mask."""
    gen = np.random.default_rng(131)
    w, b = _wb(gen, 256, 64)
    pc = PackedConv(w, b, 0, True)
    for (N, H, W, gx) in ((2, 11, 19, 0), (3, 40, 50, 5), (1, 5, 3, 1), (9, 48, 48, 0)):
        x = _rand(gen, N, 64, H, W)
        r_nat = _rand(gen, N, 256, H, W)                                        # natural channel order, NCHW
        r_perm = bf16r(r_nat).view(N, 64, 4, H, W).permute(0, 3, 4, 2, 1).reshape(N, H, W, 256).contiguous().to(DEV).to(BF16)
        o, _ = hip_conv(nhwc(x), pc, N, H, W, out_mode=1, res1=r_perm, grid_x=gx)
        ref = F.pixel_shuffle(F.conv2d(bf16r(x), bf16r(w), b, padding=1) + bf16r(r_nat), 2)
        assert_bf16_close(nchw(o), ref, 'conv 64->256 + residual + PixelShuffle(2) %s' % ((N, H, W, gx),))
    with pytest.raises(RuntimeError, match='out_mode 1'):
        hip_conv(nhwc(x), pc, N, H, W, out_mode=1, res1=r_perm, res2=r_perm)
    with pytest.raises(RuntimeError, match='out_mode 1'):
        hip_conv(nhwc(x), pc, N, H, W, out_mode=1, mask=r_perm)


def test_conv3x3_dgrad_plain_and_unshuffle():
    gen = np.random.default_rng(14)
    N, H, W = 2, 13, 17
    # plain 64->64: dgrad == autograd grad wrt input
    w, b = _wb(gen, 64, 64)
    gy = _rand(gen, N, 64, H, W)
    x = torch.zeros(N, 64, H, W, requires_grad=True)
    F.conv2d(x, bf16r(w), None, padding=1).backward(bf16r(gy))
    pc = PackedConv(w, b)
    o, _ = hip_conv(nhwc(gy), pc, N, H, W, dgrad=True)
    assert_bf16_close(nchw(o), x.grad, 'dgrad 64->64')
    # upsampler: grad wrt conv input of PixelShuffle(conv 64->256)
    w, b = _wb(gen, 256, 64)
    gy = _rand(gen, N, 64, 2 * H, 2 * W)
    x = torch.zeros(N, 64, H, W, requires_grad=True)
    F.pixel_shuffle(F.conv2d(x, bf16r(w), None, padding=1), 2).backward(bf16r(gy))
    pc = PackedConv(w, b, 0, True)
    o, _ = hip_conv(nhwc(gy), pc, N, H, W, dgrad=True, in_mode=1)
    assert_bf16_close(nchw(o), x.grad, 'dgrad upsampler (PixelShuffle^T gather)')


def test_streaming_cin256_kernel_is_bitwise_the_register_staged_kernel(monkeypatch):
    """conv_dgrad4.hip (LDS-DMA ring four stages ahead, pipelined fragment reads, one LDS counter per stage instead of workgroup barriers) against
    conv3x3_kernel<4> (RUMPY_CONV4_OLD=1: same MFMA order per accumulator) on the launches the engine makes with Cin = 256: PixelShuffle^T gather
    and plain input, with and without the residual operand, ragged sizes, one to several tiles per workgroup (grid_x), a single tile; and
    against torch for the gather case"""
    gen = np.random.default_rng(41)
    w, b = _wb(gen, 256, 64)
    pc = PackedConv(w, b, 0, True)
    for (N, H, W, gx) in ((2, 13, 17, 0), (3, 40, 50, 7), (1, 8, 16, 0), (2, 24, 33, 2), (1, 5, 3, 1), (2, 96, 96, 0), (4, 48, 48, 0)):
        gy = nhwc(_rand(gen, N, 64, 2 * H, 2 * W))
        gp = nhwc(_rand(gen, N, 256, H, W))
        r1 = nhwc(_rand(gen, N, 64, H, W))
        outs = []
        for env in ('RUMPY_CONV4_OLD', 'RUMPY_CONV4_TH8', None):      # register-staged kernel | streaming, 8-row tiles forced | streaming, tile height chosen (6 here)
            monkeypatch.delenv('RUMPY_CONV4_OLD', raising=False)
            monkeypatch.delenv('RUMPY_CONV4_TH8', raising=False)
            if env:
                monkeypatch.setenv(env, '1')
            outs.append((hip_conv(gy, pc, N, H, W, dgrad=True, in_mode=1, grid_x=gx)[0],
                         hip_conv(gy, pc, N, H, W, dgrad=True, in_mode=1, scale=0.7, res1=r1, grid_x=gx)[0],
                         hip_conv(gp, pc, N, H, W, dgrad=True, in_mode=0, res1=r1, grid_x=gx)[0]))
        monkeypatch.delenv('RUMPY_CONV4_TH8', raising=False)
        for k in range(3):
            assert torch.equal(outs[0][k].view(torch.int16), outs[1][k].view(torch.int16)), (N, H, W, gx, k)
            assert torch.equal(outs[0][k].view(torch.int16), outs[2][k].view(torch.int16)), (N, H, W, gx, k, 'six-row tiles')
        outs = [outs[0], outs[2]]
        x = torch.zeros(N, 64, H, W, requires_grad=True)
        F.pixel_shuffle(F.conv2d(x, bf16r(w), None, padding=1), 2).backward(nchw(gy).float())
        assert_bf16_close(nchw(outs[1][0]), x.grad, 'streaming dgrad upsampler')


def test_streaming_cin256_kernel_with_output_tiles_and_epilogues_is_bitwise_the_register_staged_kernel(monkeypatch):
    """the 256 -> 256 convs of the wide EDSR (four 64-channel output tiles; bias + ReLU, scale + residual, mask + two residuals): the streaming
    kernel against conv3x3_kernel<4> (RUMPY_CONV4_OLD=1), bit for bit, ragged sizes and several tiles per workgroup"""
    gen = np.random.default_rng(43)
    w, b = _wb(gen, 256, 256)
    pc = PackedConv(w, b)
    for (N, H, W, gx) in ((2, 13, 17, 0), (1, 48, 48, 0), (3, 24, 33, 3), (1, 5, 3, 1)):
        x = nhwc(_rand(gen, N, 256, H, W))
        r1, r2, m = (nhwc(_rand(gen, N, 256, H, W)) for _ in range(3))
        variants = [dict(relu=True), dict(scale=0.1, res1=r1), dict(dgrad=True, scale=0.1, mask=m), dict(dgrad=True, res1=r1, res2=r2),
                    dict(use_bias=False, scale=0.5, mask=m, res1=r1, res2=r2)]
        outs = []
        for env in ('RUMPY_CONV4_OLD', None):
            monkeypatch.delenv('RUMPY_CONV4_OLD', raising=False)
            if env:
                monkeypatch.setenv(env, '1')
            outs.append([hip_conv(x, pc, N, H, W, grid_x=gx, **kw)[0] for kw in variants])
        for k in range(len(variants)):
            assert torch.equal(outs[0][k].view(torch.int16), outs[1][k].view(torch.int16)), (N, H, W, gx, sorted(variants[k]))


@pytest.mark.parametrize('cin,cout', [(128, 128), (192, 64), (128, 192), (256, 256), (192, 192)])
def test_conv3x3_two_three_and_four_input_chunks_with_epilogues(cin, cout):
    """EDSR widths above 64 features (128, 192, the shipped 256): one launch per layer with the epilogues of the 64-feature strip kernel,
    forward and - on the dgrad image - data gradient, on a ragged image"""
    gen = np.random.default_rng(cin + cout)
    N, H, W = 2, 13, 21
    w, b = _wb(gen, cout, cin)
    x = _rand(gen, N, cin, H, W)
    r1, r2, m = (_rand(gen, N, cout, H, W) for _ in range(3))
    pc = PackedConv(w, b)
    conv = F.conv2d(bf16r(x), bf16r(w), b, padding=1)
    o, _ = hip_conv(nhwc(x), pc, N, H, W, relu=True)
    assert_bf16_close(nchw(o), conv.clamp_min(0), 'relu')
    o, _ = hip_conv(nhwc(x), pc, N, H, W, scale=0.1, res1=nhwc(r1))
    assert_bf16_close(nchw(o), conv * 0.1 + bf16r(r1), 'scale + residual')
    # data gradient of this layer: input = a gradient with `cout` channels, output `cin` channels, ReLU mask + two residuals
    gy = _rand(gen, N, cout, H, W)
    mi, ri, ri2 = (_rand(gen, N, cin, H, W) for _ in range(3))
    xr = bf16r(x).clone().requires_grad_(True)
    (F.conv2d(xr, bf16r(w), None, padding=1) * bf16r(gy)).sum().backward()
    o, _ = hip_conv(nhwc(gy), pc, N, H, W, dgrad=True, scale=0.1, mask=nhwc(mi), res1=nhwc(ri), res2=nhwc(ri2))
    assert_bf16_close(nchw(o), xr.grad * 0.1 * (bf16r(mi) > 0).float() + bf16r(ri) + bf16r(ri2), 'dgrad + mask + residuals')


def test_conv3x3_rejects_bad_arguments():
    a = L.ConvArgs(x=None, w=None, out=None, N=1, H=1, W=1, cin_chunks=1, cout_tiles=1)
    assert L.lib().rumpy_conv3x3(a, None) == -1
    assert b'null' in L.lib().rumpy_last_error()
    t = torch.zeros(64, dtype=BF16, device=DEV)
    a = L.ConvArgs(x=t.data_ptr(), w=t.data_ptr(), out=t.data_ptr(), N=1, H=1, W=1, cin_chunks=5, cout_tiles=1)
    assert L.lib().rumpy_conv3x3(a, None) == -1


@pytest.mark.parametrize('C', [1, 3])
def test_head_forward_and_wgrad(C):
    gen = np.random.default_rng(15)
    N, H, W = 3, 14, 22
    b_ = 1.0 / np.sqrt(C * 9)
    w = torch.from_numpy(gen.uniform(-b_, b_, (64, C, 3, 3)).astype(np.float32))
    b = torch.from_numpy(gen.uniform(-b_, b_, (64,)).astype(np.float32))
    x = torch.from_numpy(gen.random((N, C, H, W), dtype=np.float32))
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    out = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    L.call('rumpy_head_fwd', L.HeadFwdArgs(x=xd.data_ptr(), w=wd.data_ptr(), b=bd.data_ptr(), out=out.data_ptr(), N=N, C=C, H=H, W=W, cout=64), stream())
    ref = F.conv2d(x, w, b, padding=1)
    assert_bf16_close(nchw(out), ref, 'head fwd')
    # weight gradient
    gy = _rand(gen, N, 64, H, W)
    wp = w.clone().requires_grad_(True)
    bp = b.clone().requires_grad_(True)
    F.conv2d(x, wp, bp, padding=1).backward(bf16r(gy) * 0.25)
    slab = torch.zeros(int(L.lib().rumpy_head_wgrad_slab_floats(C, 64)), dtype=torch.float32, device=DEV)
    gw = torch.full((64, C, 3, 3), float('nan'), device=DEV)
    gb = torch.full((64,), float('nan'), device=DEV)
    L.call('rumpy_head_wgrad', L.HeadWgradArgs(x=xd.data_ptr(), dy=nhwc(gy).data_ptr(), slab=slab.data_ptr(), gw=gw.data_ptr(),
                                               gb=gb.data_ptr(), N=N, C=C, H=H, W=W, cout=64, scale=0.25), stream())
    torch.cuda.synchronize()
    assert_f32_close(gw, wp.grad, 'head wgrad', rel=1e-4)
    assert_f32_close(gb, bp.grad, 'head bgrad', rel=1e-4)


@pytest.mark.parametrize('C', [3, 1])
def test_tail_forward_l1_and_dgrad(C):
    gen = np.random.default_rng(16)
    N, H, W = 2, 21, 35
    b_ = 1.0 / np.sqrt(64 * 9)
    w = torch.from_numpy(gen.uniform(-b_, b_, (C, 64, 3, 3)).astype(np.float32))
    b = torch.from_numpy(gen.uniform(-b_, b_, (C,)).astype(np.float32))
    x = _rand(gen, N, 64, H, W)
    y = torch.from_numpy(gen.random((N, C, H, W), dtype=np.float32))
    pc = PackedConv(w, b, 2)
    out = torch.full((N, C, H, W), float('nan'), device=DEV)
    dy4 = torch.full((N, H, W, 4), float('nan'), dtype=BF16, device=DEV)
    part = torch.zeros(2048, device=DEV)
    loss = torch.zeros(1, device=DEV)
    yd, xd = y.to(DEV), nhwc(x)
    a = L.TailFwdArgs(x=xd.data_ptr(), w=pc.w_fwd.data_ptr(), bias=pc.b.data_ptr(), out=out.data_ptr(), target=yd.data_ptr(),
                      dy4=dy4.data_ptr(), loss_partial=part.data_ptr(), loss=loss.data_ptr(), N=N, C=C, H=H, W=W, grid_x=0)
    L.call('rumpy_tail_fwd', a, stream())
    torch.cuda.synchronize()
    ref = F.conv2d(bf16r(x), bf16r(w), b, padding=1)
    assert_f32_close(out, ref, 'tail fwd', rel=1e-5)
    got = out.cpu()
    assert abs(float(loss.item()) - float((got - y).abs().mean())) < 1e-6 * max(1.0, float(loss.item()))
    sg = torch.sign(got - y)
    d4 = dy4.float().cpu()
    assert torch.equal(d4[..., :C].permute(0, 3, 1, 2), sg)
    assert float(d4[..., C:].abs().sum()) == 0.0
    # plain variant (eval): no target
    out2 = torch.full((N, C, H, W), float('nan'), device=DEV)
    a2 = L.TailFwdArgs(x=xd.data_ptr(), w=pc.w_fwd.data_ptr(), bias=pc.b.data_ptr(), out=out2.data_ptr(), N=N, C=C, H=H, W=W)
    L.call('rumpy_tail_fwd', a2, stream())
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    # dgrad of the tail conv from an arbitrary 4-channel gradient
    gy = _rand(gen, N, C, H, W)
    g4 = torch.zeros(N, H, W, 4, dtype=BF16, device=DEV)
    gyd = gy.to(DEV)
    L.call('rumpy_nchw_to_nhwc4', L.NchwToNhwc4Args(src=gyd.data_ptr(), dst=g4.data_ptr(), N=N, C=C, H=H, W=W), stream())
    dx = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    L.call('rumpy_tail_dgrad', L.TailDgradArgs(dy4=g4.data_ptr(), w=pc.w_dgrad.data_ptr(), dx=dx.data_ptr(), N=N, H=H, W=W), stream())
    torch.cuda.synchronize()
    xin = torch.zeros(N, 64, H, W, requires_grad=True)
    F.conv2d(xin, bf16r(w), None, padding=1).backward(bf16r(gy))
    assert_bf16_close(nchw(dx), xin.grad, 'tail dgrad')


@pytest.mark.parametrize('N,H,W,gx', [(2, 13, 17, 0), (3, 40, 50, 7), (1, 8, 16, 0), (2, 24, 33, 2), (1, 5, 3, 1), (2, 96, 96, 0), (4, 48, 48, 0), (1, 6, 16, 0),
                                      (32, 96, 96, 0), (40, 12, 16, 0)])
def test_tail_dgrad_inside_the_last_upsampler_dgrad_is_bitwise_the_two_launches(N, H, W, gx):
    """rumpy_conv4d_tail (round 5): dx = conv^T_tail(dy4) made tile by tile INSIDE the PixelShuffle^T data-gradient launch of the last upsampler
    stage, against rumpy_tail_dgrad followed by rumpy_conv3x3(in_mode 1): dx and the stage's input gradient bit for bit - ragged sizes (edge tiles
    whose halo pixels lie outside the image must see ZERO there, not the tail gradient of an outside pixel), one and several tiles per
    workgroup, both tile heights, more tiles than workgroups; and the result against torch autograd through conv + PixelShuffle + tail conv"""
    gen = np.random.default_rng(77 + H + W)
    C = 3
    b_ = 1.0 / np.sqrt(64 * 9)
    wt = torch.from_numpy(gen.uniform(-b_, b_, (C, 64, 3, 3)).astype(np.float32))
    bt = torch.from_numpy(gen.uniform(-b_, b_, (C,)).astype(np.float32))
    pt = PackedConv(wt, bt, 2)
    wu, bu = _wb(gen, 256, 64)
    pu = PackedConv(wu, bu, 0, True)
    gy = _rand(gen, N, C, 2 * H, 2 * W)
    g4 = torch.zeros(N, 2 * H, 2 * W, 4, dtype=BF16, device=DEV)
    gyd = gy.to(DEV)
    L.call('rumpy_nchw_to_nhwc4', L.NchwToNhwc4Args(src=gyd.data_ptr(), dst=g4.data_ptr(), N=N, C=C, H=2 * H, W=2 * W), stream())
    dx_ref = torch.full((N, 2 * H, 2 * W, 64), float('nan'), dtype=BF16, device=DEV)
    L.call('rumpy_tail_dgrad', L.TailDgradArgs(dy4=g4.data_ptr(), w=pt.w_dgrad.data_ptr(), dx=dx_ref.data_ptr(), N=N, H=2 * H, W=2 * W), stream())
    o_ref, _ = hip_conv(dx_ref, pu, N, H, W, dgrad=True, in_mode=1, grid_x=gx)
    dx = torch.full((N, 2 * H, 2 * W, 64), float('nan'), dtype=BF16, device=DEV)
    out = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    L.call('rumpy_conv4d_tail', L.Conv4dTailArgs(dy4=g4.data_ptr(), w_tail=pt.w_dgrad.data_ptr(), dx=dx.data_ptr(), w=pu.w_dgrad.data_ptr(),
                                                 out=out.data_ptr(), N=N, H=H, W=W, grid_x=gx), stream())
    torch.cuda.synchronize()
    assert torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16)), 'dx'
    assert torch.equal(out.view(torch.int16), o_ref.view(torch.int16)), 'stage input gradient'
    if N * H * W <= 4 * 48 * 48:
        x = torch.zeros(N, 64, H, W, requires_grad=True)
        mid = F.pixel_shuffle(F.conv2d(x, bf16r(wu), None, padding=1), 2)
        mid.retain_grad()
        F.conv2d(mid, bf16r(wt), None, padding=1).backward(bf16r(gy))
        assert_bf16_close(nchw(dx), mid.grad, 'tail dgrad inside the upsampler dgrad')
        # the stage gradient from the bf16-ROUNDED dx, as both HIP paths compute it
        x2 = torch.zeros(N, 64, H, W, requires_grad=True)
        F.pixel_shuffle(F.conv2d(x2, bf16r(wu), None, padding=1), 2).backward(nchw(dx).float())
        assert_bf16_close(nchw(out), x2.grad, 'upsampler dgrad from the in-kernel tail gradient')


def test_conv4d_tail_refuses_bad_arguments():
    with pytest.raises(RuntimeError, match='rumpy_conv4d_tail'):
        L.call('rumpy_conv4d_tail', L.Conv4dTailArgs(dy4=None, w_tail=None, dx=None, w=None, out=None, N=1, H=8, W=8, grid_x=0), stream())


def _wgrad_ref(x, gy, co, scale=1.0):
    w = torch.zeros(co, 64, 3, 3, requires_grad=True)
    b = torch.zeros(co, requires_grad=True)
    F.conv2d(bf16r(x), w, b, padding=1).backward(bf16r(gy) * scale)
    return w.grad, b.grad


@pytest.mark.parametrize('variant', [0, 1])
@pytest.mark.parametrize('N,H,W,split', [(2, 12, 12, 1), (4, 16, 32, 2), (3, 9, 21, 3), (1, 48, 48, 1)])
def test_wgrad_grouped_plain(N, H, W, split, variant):
    gen = np.random.default_rng(17 + H)
    x, gy = _rand(gen, N, 64, H, W), _rand(gen, N, 64, H, W)
    xd, gd = nhwc(x), nhwc(gy)
    per = (N + split - 1) // split
    jobs = [dict(x=xd, dy=gd, n0=n0, n1=min(N, n0 + per), H=H, W=W, x_cstride=64, x_coff=0, dy_mode=0, dy_cstride=64, dy_coff=0)
            for n0 in range(0, N, per)]
    gw = torch.full((64, 64, 3, 3), float('nan'), device=DEV)
    gb = torch.full((64,), float('nan'), device=DEV)
    hip_wgrad(jobs, 4, [dict(first_job=0, njobs=len(jobs), co_count=64, co_mode=0, co_off=0, ci_total=64, ci_off=0,
                             write_bias=1, scale=0.5)], gw, gb, variant)
    rw, rb = _wgrad_ref(x, gy, 64, 0.5)
    assert_f32_close(gw, rw, 'wgrad 64x64', rel=1e-4)
    assert_f32_close(gb, rb, 'bgrad', rel=1e-4)


def test_wgrad_upsampler_unshuffle_view():
    gen = np.random.default_rng(18)
    N, H, W = 2, 10, 18
    x = _rand(gen, N, 64, H, W)
    gy = _rand(gen, N, 64, 2 * H, 2 * W)            # gradient of the shuffled output
    w = torch.zeros(256, 64, 3, 3, requires_grad=True)
    b = torch.zeros(256, requires_grad=True)
    F.pixel_shuffle(F.conv2d(bf16r(x), w, b, padding=1), 2).backward(bf16r(gy))
    xd, gd = nhwc(x), nhwc(gy)
    jobs, red = [], []
    for q in range(4):
        jobs.append(dict(x=xd, dy=gd, n0=0, n1=N, H=H, W=W, x_cstride=64, x_coff=0, dy_mode=1, dy_cstride=64, dy_coff=q))
        red.append(dict(first_job=q, njobs=1, co_count=64, co_mode=1, co_off=q, ci_total=64, ci_off=0, write_bias=1, scale=1.0))
    gw = torch.full((256, 64, 3, 3), float('nan'), device=DEV)
    gb = torch.full((256,), float('nan'), device=DEV)
    hip_wgrad(jobs, 4, red, gw, gb)
    assert_f32_close(gw, w.grad, 'wgrad upsampler', rel=1e-4)
    assert_f32_close(gb, b.grad, 'bgrad upsampler', rel=1e-4)


@pytest.mark.parametrize('W,variant', [(20, 0), (20, 1), (23, 1), (34, 0)])
def test_wgrad_tail_dy4(W, variant):
    gen = np.random.default_rng(19)
    N, H, C = 2, 24, 3
    x = _rand(gen, N, 64, H, W)
    gy = torch.sign(_rand(gen, N, C, H, W))
    g4 = torch.zeros(N, H, W, 4, dtype=BF16)
    g4[..., :C] = gy.permute(0, 2, 3, 1).to(BF16)
    xd, gd = nhwc(x), g4.to(DEV)
    jobs = [dict(x=xd, dy=gd, n0=n, n1=n + 1, H=H, W=W, x_cstride=64, x_coff=0, dy_mode=2, dy_cstride=4, dy_coff=0) for n in range(N)]
    gw = torch.full((C, 64, 3, 3), float('nan'), device=DEV)
    gb = torch.full((C,), float('nan'), device=DEV)
    hip_wgrad(jobs, 1, [dict(first_job=0, njobs=N, co_count=C, co_mode=0, co_off=0, ci_total=64, ci_off=0, write_bias=1, scale=1.0 / 7)], gw, gb, variant)
    rw, rb = _wgrad_ref(x, gy, C, 1.0 / 7)
    assert_f32_close(gw, rw, 'tail wgrad', rel=1e-4)
    assert_f32_close(gb, rb, 'tail bgrad', rel=1e-4)


@pytest.mark.parametrize('Cc,Cr', [(64, 4), (128, 8), (256, 16), (128, 6)])
def test_channel_attention_forward_backward(Cc, Cr):
    gen = np.random.default_rng(20)
    N, H, W = 3, 10, 23
    t2, xres, gy = _rand(gen, N, Cc, H, W), _rand(gen, N, Cc, H, W), _rand(gen, N, Cc, H, W)
    w1 = torch.from_numpy(gen.uniform(-0.3, 0.3, (Cr, Cc, 1, 1)).astype(np.float32)).requires_grad_(True)
    b1 = torch.from_numpy(gen.uniform(-0.3, 0.3, (Cr,)).astype(np.float32)).requires_grad_(True)
    w2 = torch.from_numpy(gen.uniform(-0.3, 0.3, (Cc, Cr, 1, 1)).astype(np.float32)).requires_grad_(True)
    b2 = torch.from_numpy(gen.uniform(-0.3, 0.3, (Cc,)).astype(np.float32)).requires_grad_(True)
    tt = bf16r(t2).requires_grad_(True)
    gate_ref = torch.sigmoid(F.conv2d(F.relu(F.conv2d(tt.mean(dim=(2, 3), keepdim=True), w1, b1)), w2, b2))
    y_ref = bf16r(xres) + tt * gate_ref
    y_ref.backward(bf16r(gy))
    # device: pool partials as the conv epilogue would leave them (one "tile" = whole image here)
    pool = bf16r(t2).sum(dim=(2, 3)).reshape(N, 1, Cc).contiguous().to(DEV)
    d = lambda t: t.detach().contiguous().to(DEV)
    w1d, b1d, w2d, b2d = d(w1), d(b1), d(w2), d(b2)
    mean, hid, gate = (torch.zeros(N, k, device=DEV) for k in (Cc, Cr, Cc))
    L.call('rumpy_ca_mlp_fwd', L.CaMlpFwdArgs(pool=pool.data_ptr(), w1=w1d.data_ptr(), b1=b1d.data_ptr(), w2=w2d.data_ptr(), b2=b2d.data_ptr(),
                                              mean=mean.data_ptr(), hidden=hid.data_ptr(), gate=gate.data_ptr(), N=N, C=Cc, Cr=Cr, ntiles=1,
                                              inv_hw=1.0 / (H * W)), stream())
    torch.cuda.synchronize()
    assert_f32_close(gate, gate_ref.reshape(N, Cc), 'CA gate', rel=1e-5)
    t2d, xd, gd = nhwc(t2), nhwc(xres), nhwc(gy)
    y = torch.zeros(N, H, W, Cc, dtype=BF16, device=DEV)
    L.call('rumpy_ca_scale_res_fwd', L.CaScaleArgs(t=t2d.data_ptr(), res=xd.data_ptr(), gate=gate.data_ptr(), out=y.data_ptr(), N=N, HW=H * W, C=Cc), stream())
    torch.cuda.synchronize()
    assert_bf16_close(nchw(y), y_ref, 'CA scale+res')
    # backward
    nchunks = (H * W + 127) // 128
    part = torch.zeros(N, nchunks, Cc, device=DEV)
    L.call('rumpy_ca_bwd_reduce', L.CaBwdReduceArgs(dy=gd.data_ptr(), t=t2d.data_ptr(), partial=part.data_ptr(), N=N, HW=H * W, C=Cc), stream())
    dpool = torch.zeros(N, Cc, device=DEV)
    gw1, gb1, gw2, gb2 = (torch.full(s, float('nan'), device=DEV) for s in ((Cr, Cc, 1, 1), (Cr,), (Cc, Cr, 1, 1), (Cc,)))
    L.call('rumpy_ca_mlp_bwd', L.CaMlpBwdArgs(partial=part.data_ptr(), mean=mean.data_ptr(), hidden=hid.data_ptr(), gate=gate.data_ptr(),
                                              w1=w1d.data_ptr(), w2=w2d.data_ptr(), dpool=dpool.data_ptr(), gw1=gw1.data_ptr(), gb1=gb1.data_ptr(),
                                              gw2=gw2.data_ptr(), gb2=gb2.data_ptr(), N=N, C=Cc, Cr=Cr, nchunks=nchunks, inv_hw=1.0 / (H * W),
                                              scale=1.0), stream())
    dt = torch.zeros(N, H, W, Cc, dtype=BF16, device=DEV)
    L.call('rumpy_ca_bwd_apply', L.CaBwdApplyArgs(dy=gd.data_ptr(), gate=gate.data_ptr(), dpool=dpool.data_ptr(), dt=dt.data_ptr(), N=N, HW=H * W, C=Cc), stream())
    torch.cuda.synchronize()
    assert_bf16_close(nchw(dt), tt.grad, 'CA d(t2)')
    assert_f32_close(gw1, w1.grad, 'CA gW1', rel=1e-4)
    assert_f32_close(gb1, b1.grad, 'CA gb1', rel=1e-4)
    assert_f32_close(gw2, w2.grad, 'CA gW2', rel=1e-4)
    assert_f32_close(gb2, b2.grad, 'CA gb2', rel=1e-4)
    # ---- the fused forms the engine launches: identical results ----
    mean2, hid2, gate2 = (torch.full((N, k), float('nan'), device=DEV) for k in (Cc, Cr, Cc))
    y2 = torch.full((N, H, W, Cc), float('nan'), dtype=BF16, device=DEV)
    L.call('rumpy_ca_fwd_fused', L.CaFwdFusedArgs(pool=pool.data_ptr(), w1=w1d.data_ptr(), b1=b1d.data_ptr(), w2=w2d.data_ptr(), b2=b2d.data_ptr(),
                                                  mean=mean2.data_ptr(), hidden=hid2.data_ptr(), gate=gate2.data_ptr(), t=t2d.data_ptr(),
                                                  res=xd.data_ptr(), out=y2.data_ptr(), N=N, HW=H * W, C=Cc, Cr=Cr, ntiles=1,
                                                  inv_hw=1.0 / (H * W)), stream())
    torch.cuda.synchronize()
    assert torch.equal(y2, y) and torch.equal(mean2, mean) and torch.equal(hid2, hid) and torch.equal(gate2, gate)
    L.call('rumpy_ca_bwd_reduce', L.CaBwdReduceArgs(dy=gd.data_ptr(), t=t2d.data_ptr(), partial=part.data_ptr(), N=N, HW=H * W, C=Cc), stream())
    dz = torch.full((N, Cc), float('nan'), device=DEV)
    dt2 = torch.full((N, H, W, Cc), float('nan'), dtype=BF16, device=DEV)
    L.call('rumpy_ca_bwd_fused', L.CaBwdFusedArgs(dy=gd.data_ptr(), partial=part.data_ptr(), hidden=hid.data_ptr(), gate=gate.data_ptr(),
                                                  w1=w1d.data_ptr(), w2=w2d.data_ptr(), dz=dz.data_ptr(), dt=dt2.data_ptr(), N=N, HW=H * W, C=Cc,
                                                  Cr=Cr, nchunks=nchunks, inv_hw=1.0 / (H * W)), stream())
    torch.cuda.synchronize()
    assert torch.equal(dt2, dt)
    # parameter gradients of (here two copies of) the layer in one batched launch, from dz
    g2 = [tuple(torch.full(sh, float('nan'), device=DEV) for sh in ((Cr, Cc, 1, 1), (Cr,), (Cc, Cr, 1, 1), (Cc,))) for _ in range(2)]
    items = (L.CaMlpBwdArgs * 2)(*[L.CaMlpBwdArgs(partial=dz.data_ptr(), mean=mean.data_ptr(), hidden=hid.data_ptr(), gate=gate.data_ptr(),
                                                  w1=w1d.data_ptr(), w2=w2d.data_ptr(), dpool=dz.data_ptr(), gw1=q[0].data_ptr(), gb1=q[1].data_ptr(),
                                                  gw2=q[2].data_ptr(), gb2=q[3].data_ptr(), N=N, C=Cc, Cr=Cr, nchunks=1, inv_hw=1.0 / (H * W),
                                                  scale=0.5 * (k + 1)) for k, q in enumerate(g2)])
    idev = to_dev_bytes(items)
    L.check(L.lib().rumpy_ca_mlp_bwd_params(idev.data_ptr(), 2, N, Cc, Cr, stream()), 'params')
    torch.cuda.synchronize()
    for k, q in enumerate(g2):
        for got, ref in zip(q, (gw1, gb1, gw2, gb2)):
            assert_f32_close(got, ref * (0.5 * (k + 1)), 'batched CA parameter gradients', rel=1e-6)


def test_adam_matches_torch_and_clips():
    gen = np.random.default_rng(21)
    n = 100003
    p0 = torch.from_numpy(gen.standard_normal(n).astype(np.float32))
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    hyper = torch.zeros(8, device=DEV)
    sumsq, part = torch.zeros(1, device=DEV), torch.zeros(1024, device=DEV)
    for t in range(1, 4):
        g = torch.from_numpy(gen.standard_normal(n).astype(np.float32) * 10 ** (-t))
        max_norm = 0.5 if t == 2 else 0.0
        ref_p.grad = g.clone()
        if max_norm:
            torch.nn.utils.clip_grad_norm_([ref_p], max_norm)
        opt.step()
        gd = g.to(DEV)
        hyper.copy_(torch.tensor([1e-3, 0.9, 0.999, 1e-8, 1 - 0.9 ** t, float(np.sqrt(1 - 0.999 ** t)), 1.0, max_norm]))
        if max_norm:
            L.call('rumpy_sumsq', L.SumsqArgs(g=gd.data_ptr(), n=n, partial=part.data_ptr(), out=sumsq.data_ptr()), stream())
            torch.cuda.synchronize()
            assert abs(float(sumsq.item()) - float((g.double() ** 2).sum())) < 1e-4 * float((g.double() ** 2).sum())
        L.call('rumpy_adam_step', L.AdamArgs(p=p.data_ptr(), g=gd.data_ptr(), m=m.data_ptr(), v=v.data_ptr(), n=n, hyper=hyper.data_ptr(),
                                             sumsq=sumsq.data_ptr() if max_norm else None), stream())
        torch.cuda.synchronize()
        assert float((p.cpu() - ref_p.detach()).abs().max()) < 2e-6, t


def test_eval_post_clip_ycbcr_psnr():
    from oracle import sr_oracle as O
    gen = np.random.default_rng(22)
    out = torch.from_numpy(gen.uniform(-0.2, 1.2, (2, 3, 31, 17)).astype(np.float32))
    ref = torch.from_numpy(gen.random((2, 3, 31, 17), dtype=np.float32))
    from rumpy_amd.SISR.models.interface import SISRInterface
    rgb, ycbcr, p = SISRInterface.postprocess(out.to(DEV), ref.to(DEV))
    rgb_ref = O.clip01(out.numpy())
    yc_ref = np.copy(rgb_ref)
    hr_ref = O.clip01(ref.numpy())
    for i in range(2):
        yc_ref[i] = O.rgb_to_ycbcr_jpg(yc_ref[i])
        hr_ref[i] = O.rgb_to_ycbcr_jpg(hr_ref[i])
    assert np.array_equal(rgb.cpu().numpy(), rgb_ref)
    np.testing.assert_allclose(ycbcr.cpu().numpy(), yc_ref, rtol=0, atol=2e-7)
    assert abs(p - O.y_psnr(yc_ref, hr_ref)) < 1e-3


def test_wgrad_jobs_split_by_tile_subranges():
    """jobs may cover any contiguous tile sub-range of the image range; the reduction adds them up"""
    gen = np.random.default_rng(23)
    N, H, W = 3, 20, 40          # 3 x 3 x 3 = 27 tiles
    x, gy = _rand(gen, N, 64, H, W), _rand(gen, N, 64, H, W)
    xd, gd = nhwc(x), nhwc(gy)
    cuts = [0, 1, 8, 20, 27]
    jobs = [dict(x=xd, dy=gd, n0=0, n1=N, t0=a, t1=b, H=H, W=W, x_cstride=64, x_coff=0, dy_mode=0, dy_cstride=64, dy_coff=0)
            for a, b in zip(cuts[:-1], cuts[1:])]
    gw = torch.full((64, 64, 3, 3), float('nan'), device=DEV)
    gb = torch.full((64,), float('nan'), device=DEV)
    hip_wgrad(jobs, 4, [dict(first_job=0, njobs=len(jobs), co_count=64, co_mode=0, co_off=0, ci_total=64, ci_off=0,
                             write_bias=1, scale=1.0)], gw, gb)
    rw, rb = _wgrad_ref(x, gy, 64, 1.0)
    assert_f32_close(gw, rw, 'wgrad tile sub-ranges', rel=1e-4)
    assert_f32_close(gb, rb, 'bgrad tile sub-ranges', rel=1e-4)


def test_wgrad_shares_run_job_lists_bitwise_like_one_job_per_workgroup():
    """rumpy_wgrad_shares: a workgroup runs a LIST of jobs (here of two layers with different shapes, incl. an empty share and a share
    that ends one layer and begins the next) - every slab equals the one-job-per-workgroup launch's bit for bit"""
    gen = np.random.default_rng(29)
    N = 2
    xa, ga = nhwc(_rand(gen, N, 64, 20, 40)), nhwc(_rand(gen, N, 64, 20, 40))        # 18 tiles
    xb, gb_ = nhwc(_rand(gen, N, 64, 9, 17)), nhwc(_rand(gen, N, 64, 9, 17))         # 8 tiles
    mk = lambda x, dy, H, W, a, b: dict(x=x, dy=dy, n0=0, n1=N, t0=a, t1=b, H=H, W=W, x_cstride=64, x_coff=0, dy_mode=0, dy_cstride=64, dy_coff=0)
    jobs = [mk(xa, ga, 20, 40, 0, 7), mk(xa, ga, 20, 40, 7, 13), mk(xa, ga, 20, 40, 13, 18), mk(xb, gb_, 9, 17, 0, 3), mk(xb, gb_, 9, 17, 3, 8)]
    red = [dict(first_job=0, njobs=3, co_count=64, co_mode=0, co_off=0, ci_total=64, ci_off=0, write_bias=1, scale=1.0)]
    gw1, gb1 = torch.zeros(64, 64, 3, 3, device=DEV), torch.zeros(64, device=DEV)
    gw2, gb2 = torch.zeros(64, 64, 3, 3, device=DEV), torch.zeros(64, device=DEV)
    s1 = hip_wgrad(jobs, 4, red, gw1, gb1)
    s2 = hip_wgrad(jobs, 4, red, gw2, gb2, shares=[0, 1, 1, 4, 5])          # shares: [job 0], [], [jobs 1-3: two layers], [job 4]
    assert torch.equal(s1.view(torch.int32), s2.view(torch.int32)) and torch.equal(gw1, gw2) and torch.equal(gb1, gb2)


# ---------------------------------------------------------------------------------------------------------------------
# residual block in one launch (conv_block.hip) against the same two layers through rumpy_conv3x3
# ---------------------------------------------------------------------------------------------------------------------
def _run_block(x, pa, pb, N, H, W, fwd, rs, mask=None, extra=None, store_t=True):
    t = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV) if store_t else None
    out = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    p = lambda z: None if z is None else z.data_ptr()
    if fwd:
        a = L.BlockArgs(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                        mask=None, res2=None, t=p(t), out=out.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=float(rs))
    else:   # data gradient: first through conv2 (pb) masked, then through conv1 (pa)
        a = L.BlockArgs(x=x.data_ptr(), w1=pb.w_dgrad.data_ptr(), b1=None, w2=pa.w_dgrad.data_ptr(), b2=None, mask=p(mask), res2=p(extra),
                        t=p(t), out=out.data_ptr(), N=N, H=H, W=W, relu1=0, scale1=float(rs), scale2=1.0)
    L.call('rumpy_conv_block', a, stream())
    torch.cuda.synchronize()
    return t, out


@pytest.mark.parametrize('N,H,W', [(1, 6, 16), (2, 13, 48), (3, 20, 37), (1, 5, 9), (32, 48, 48),
                                   # wider than one strip: column tiles of 32 / 48 output columns with the activation's halo columns computed
                                   (1, 7, 49), (2, 13, 64), (1, 20, 100), (2, 9, 128), (8, 64, 64), (1, 31, 170)])
def test_conv_block_matches_two_layer_launches(N, H, W):
    gen = np.random.default_rng(100 + H + W)
    mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.06, 0.06, (64, 64, 3, 3)).astype(np.float32)),
                            torch.from_numpy(gen.uniform(-0.1, 0.1, 64).astype(np.float32)))
    pa, pb = mk(), mk()
    x = torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    rs = 0.1
    # forward: t = relu(conv1 x + b1); y = x + rs * (conv2 t + b2)
    t_ref, _ = hip_conv(x, pa, N, H, W, relu=True)
    y_ref, _ = hip_conv(t_ref, pb, N, H, W, scale=rs, res1=x)
    t, y = _run_block(x, pa, pb, N, H, W, True, rs)
    assert torch.equal(t, t_ref), 'activation between the two convs'
    assert_bf16_close(y.float(), y_ref.float(), 'block forward', rel=2e-3, amax=2.0 ** -7)
    _, y2 = _run_block(x, pa, pb, N, H, W, True, rs, store_t=False)        # inference: activation not stored
    assert torch.equal(y2, y)
    # data gradient: gt = mask(t) . rs * conv2^T(g); gx = g + conv1^T(gt) + extra
    g = torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    extra = torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    gt_ref, _ = hip_conv(g, pb, N, H, W, dgrad=True, scale=rs, mask=t_ref)
    gx_ref, _ = hip_conv(gt_ref, pa, N, H, W, dgrad=True, res1=g, res2=extra)
    gt, gx = _run_block(g, pa, pb, N, H, W, False, rs, mask=t_ref, extra=extra)
    assert torch.equal(gt, gt_ref), 'gradient w.r.t. the activation'
    assert_bf16_close(gx.float(), gx_ref.float(), 'block data gradient', rel=2e-3, amax=2.0 ** -7)


@pytest.mark.parametrize('N,H,W', [(2, 13, 48), (3, 20, 37), (1, 5, 9), (2, 13, 64), (1, 20, 100)])
def test_conv_block_geometries_agree_bitwise(N, H, W):
    """one strip across the image (W <= 48), column tiles of 32 and column tiles of 48 columns (col_tile = 2 / 3 forces them at any W)
    compute every output element with the same MFMA sequence: activation and output are bitwise equal, forward and data gradient, with
    the ReLU mask as the stored activation and as bytes"""
    gen = np.random.default_rng(500 + H + W)
    pa, pb = PackedConv(*_wb(gen, 64, 64)), PackedConv(*_wb(gen, 64, 64))
    rnd = lambda: torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    x, g, extra = rnd(), rnd(), rnd()
    res = []
    for ct in ((0, 2, 3) if W <= 48 else (2, 3)):
        t, y = (torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV) for _ in range(2))
        mb = torch.full((N, H, W, 8), 0xAA, dtype=torch.uint8, device=DEV)
        L.call('rumpy_conv_block', L.BlockArgs(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(),
                                               b2=pb.b_packed.data_ptr(), t=t.data_ptr(), out=y.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0,
                                               scale2=0.1, maskbits=mb.data_ptr(), col_tile=ct), stream())
        outs = [t, y, mb]
        for bits in (False, True):
            dt, dx = (torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV) for _ in range(2))
            L.call('rumpy_conv_block', L.BlockArgs(x=g.data_ptr(), w1=pb.w_dgrad.data_ptr(), w2=pa.w_dgrad.data_ptr(), mask=t.data_ptr(),
                                                   res2=extra.data_ptr(), t=dt.data_ptr(), out=dx.data_ptr(), N=N, H=H, W=W, relu1=0, scale1=0.1,
                                                   scale2=1.0, maskbits=mb.data_ptr() if bits else None, col_tile=ct), stream())
            outs += [dt, dx]
        torch.cuda.synchronize()
        assert all(torch.isfinite(o.float()).all() for o in outs if o.dtype == BF16), ct
        res.append(outs)
    for other in res[1:]:
        for i, (a, b) in enumerate(zip(res[0], other)):
            assert torch.equal(a.view(torch.int16) if a.dtype == BF16 else a, b.view(torch.int16) if b.dtype == BF16 else b), i
    a = L.BlockArgs(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), w2=pb.w_fwd.data_ptr(), out=x.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=1.0, col_tile=4)
    assert L.lib().rumpy_conv_block(a, None) == -1 and b'col_tile' in L.lib().rumpy_last_error()


@pytest.mark.parametrize('N,H,W', [(2, 13, 64), (1, 20, 100), (16, 64, 64), (1, 7, 49), (3, 33, 130)])
def test_conv_block_strip_heights_agree_bitwise(N, H, W, monkeypatch):
    """round 4: strips of 4 / 6 / 8 rows x column tiles of 32 / 48 columns (the geometry rumpy_conv_block picks by itself so that the
    workgroup count fits the CUs; RUMPY_BLOCK_GEO forces one) compute every output element with the same MFMA sequence: activation, output
    and mask bytes of the forward launch (bf16 and fp16) and both tensors of the mask-byte data-gradient launch are bitwise equal."""
    gen = np.random.default_rng(900 + H + W)
    pa, pb = PackedConv(*_wb(gen, 64, 64)), PackedConv(*_wb(gen, 64, 64))
    rnd = lambda: torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    x, g, extra = rnd(), rnd(), rnd()
    xh = x.float().to(torch.float16)
    wh = lambda pc: pc.w_fwd.float().to(torch.float16)      # (any fp16 bit patterns do for a bitwise comparison between geometries)
    pah, pbh = wh(pa), wh(pb)
    res = []
    for geo in ('6,3', '6,2', '8,2', '4,2', None):
        if geo is None:
            monkeypatch.delenv('RUMPY_BLOCK_GEO', raising=False)
        else:
            monkeypatch.setenv('RUMPY_BLOCK_GEO', geo)
        t, y, dt, dx = (torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV) for _ in range(4))
        yh = torch.full((N, H, W, 64), float('nan'), dtype=torch.float16, device=DEV)
        mb = torch.full((N, H, W, 8), 0xAA, dtype=torch.uint8, device=DEV)
        L.call('rumpy_conv_block', L.BlockArgs(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(),
                                               b2=pb.b_packed.data_ptr(), t=t.data_ptr(), out=y.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0,
                                               scale2=0.1, maskbits=mb.data_ptr()), stream())
        L.call('rumpy_conv_block', L.BlockArgs(x=xh.data_ptr(), w1=pah.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pbh.data_ptr(),
                                               b2=pb.b_packed.data_ptr(), out=yh.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=0.1,
                                               fmt=L.FMT_F16), stream())
        L.call('rumpy_conv_block', L.BlockArgs(x=g.data_ptr(), w1=pb.w_dgrad.data_ptr(), w2=pa.w_dgrad.data_ptr(), res2=extra.data_ptr(),
                                               t=dt.data_ptr(), out=dx.data_ptr(), N=N, H=H, W=W, relu1=0, scale1=0.1, scale2=1.0,
                                               maskbits=mb.data_ptr()), stream())
        torch.cuda.synchronize()
        outs = [t, y, mb, yh, dt, dx]
        assert all(torch.isfinite(o.float()).all() for o in outs if o.dtype != torch.uint8), geo
        res.append(outs)
    for gi, other in enumerate(res[1:]):
        for i, (a, b) in enumerate(zip(res[0], other)):
            assert torch.equal(a.view(torch.int16) if a.dtype != torch.uint8 else a, b.view(torch.int16) if b.dtype != torch.uint8 else b), (gi, i)


@pytest.mark.parametrize('N,H,W', [(2, 13, 48), (3, 20, 37), (8, 48, 48), (2, 13, 64), (1, 20, 100)])
def test_conv_block_rcab_form_matches_two_layer_launches(N, H, W):
    """general form of the block kernel (RCAB): no residual + pool partial sums forward; external residual operand backward"""
    gen = np.random.default_rng(300 + H + W)
    mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.06, 0.06, (64, 64, 3, 3)).astype(np.float32)),
                            torch.from_numpy(gen.uniform(-0.1, 0.1, 64).astype(np.float32)))
    pa, pb = mk(), mk()
    rnd = lambda: torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    x = rnd()
    # forward: t1 = relu(conv1 x + b1); t2 = conv2 t1 + b2 with per-(strip, row half) channel sums
    t_ref, _ = hip_conv(x, pa, N, H, W, relu=True)
    y_ref, pool_ref = hip_conv(t_ref, pb, N, H, W, pool=True)
    tiles = int(L.lib().rumpy_block_pool_tiles(H, W))       # (W <= 48: the strip kernel's count and layout)
    assert W > 48 or tiles == int(L.lib().rumpy_conv_pool_tiles(H, W, 1))
    t = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    y = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    pool = torch.full((N, tiles, 64), float('nan'), dtype=torch.float32, device=DEV)
    a = L.BlockArgs(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                    t=t.data_ptr(), out=y.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=1.0, res_mode=1, pool=pool.data_ptr())
    L.call('rumpy_conv_block', a, stream())
    torch.cuda.synchronize()
    assert torch.equal(t, t_ref) and torch.equal(y, y_ref)
    if W <= 48:
        assert_f32_close(pool, pool_ref, 'pool partial sums', rel=1e-5)
    else:       # column tiles: other partial rows, the same sums
        assert_f32_close(pool.sum(1), pool_ref.sum(1), 'pool sums', rel=1e-5)
    # data gradient: dt1 = mask(t1) . conv2^T(dt2); dx = conv1^T(dt1) + g + extra
    dt2, g, extra = rnd(), rnd(), rnd()
    d1_ref, _ = hip_conv(dt2, pb, N, H, W, dgrad=True, mask=t_ref)
    dx_ref, _ = hip_conv(d1_ref, pa, N, H, W, dgrad=True, res1=g, res2=extra)
    d1 = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    dx = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    a = L.BlockArgs(x=dt2.data_ptr(), w1=pb.w_dgrad.data_ptr(), w2=pa.w_dgrad.data_ptr(), mask=t_ref.data_ptr(), res2=extra.data_ptr(),
                    t=d1.data_ptr(), out=dx.data_ptr(), N=N, H=H, W=W, relu1=0, scale1=1.0, scale2=1.0, res_mode=2, res1=g.data_ptr())
    L.call('rumpy_conv_block', a, stream())
    torch.cuda.synchronize()
    assert torch.equal(d1, d1_ref) and torch.equal(dx, dx_ref)


@pytest.mark.parametrize('N,H,W', [(2, 13, 48), (3, 20, 37), (32, 48, 48), (2, 24, 24), (2, 20, 20), (1, 5, 9), (2, 24, 64), (1, 13, 100)])
def test_conv_block_mask_bytes_equal_the_activation_mask(N, H, W):
    """the ReLU mask handed from the forward block launch to the data-gradient launch as one byte per 8 channels (maskbits) gives
    bit-identical results to masking with the stored bf16 activation"""
    gen = np.random.default_rng(N * 100 + W)
    pa, pb = PackedConv(*_wb(gen, 64, 64)), PackedConv(*_wb(gen, 64, 64))
    x = torch.randn(N, H, W, 64, device=DEV).to(BF16)
    g = torch.randn(N, H, W, 64, device=DEV).to(BF16)
    t1 = torch.empty(N, H, W, 64, dtype=BF16, device=DEV)
    y = torch.empty_like(t1)
    mb = torch.full((N, H, W, 8), 0xAA, dtype=torch.uint8, device=DEV)
    L.call('rumpy_conv_block', L.BlockArgs(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                                           t=t1.data_ptr(), out=y.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=0.1, maskbits=mb.data_ptr()), stream())
    torch.cuda.synchronize()
    # the bytes are exactly the sign pattern of the stored activation
    want = ((t1.float() > 0).reshape(N, H, W, 8, 8).to(torch.int32) << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(mb, want)
    outs = []
    for bits in (False, True):
        dt1, dx = torch.zeros_like(t1), torch.zeros_like(t1)
        L.call('rumpy_conv_block', L.BlockArgs(x=g.data_ptr(), w1=pb.w_dgrad.data_ptr(), w2=pa.w_dgrad.data_ptr(), mask=t1.data_ptr(), t=dt1.data_ptr(),
                                               out=dx.data_ptr(), N=N, H=H, W=W, relu1=0, scale1=0.1, scale2=1.0, maskbits=mb.data_ptr() if bits else None), stream())
        torch.cuda.synchronize()
        outs.append((dt1, dx))
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16))


def test_device_image_save_is_the_host_expression(tmp_path):
    """rumpy_to_uint8_hwc (sr_tools/visualization.py: what saves evaluation outputs that are still on the GPU) against numpy's
    clip(im * 255 / max_val, 0, 255).astype(uint8) of visualization.py:56 - truncation, values outside [0, max_val], odd sizes, 1 and 3 channels"""
    from PIL import Image
    from rumpy_amd.sr_tools.visualization import safe_image_save, to_uint8_hwc
    gen = np.random.default_rng(6)
    for shape, mv in (((2, 3, 37, 53), 1), ((1, 1, 5, 7), 1), ((3, 3, 64, 64), 2.5)):
        im = gen.uniform(-0.3 * mv, 1.3 * mv, shape).astype(np.float32)
        im.reshape(-1)[:4] = np.array([0.999, 254.9999 / 255, 1.0, 0.5], dtype=np.float32) * mv
        ref = np.clip(im.transpose(0, 2, 3, 1) * 255 / mv, 0, 255).astype(np.uint8)
        assert np.array_equal(to_uint8_hwc(torch.from_numpy(im).to(DEV), mv), ref), shape
    im = gen.uniform(0, 1, (2, 3, 20, 30)).astype(np.float32)
    safe_image_save(torch.from_numpy(im).to(DEV), str(tmp_path), ['x.png', 'y.png'], config='rgb')
    assert np.array_equal(np.asarray(Image.open(tmp_path / 'y.png')), np.clip(im[1].transpose(1, 2, 0) * 255, 0, 255).astype(np.uint8))

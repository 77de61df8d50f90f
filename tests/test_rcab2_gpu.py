"""-m gpu parity of conv_rcab2.hip (round 5): a residual channel-attention block per launch with NO exchange between the workgroups of a launch -
the attention gate of a block is applied by the launch that consumes its output (rumpy_rcab2_fwd / rumpy_rcab2_bwd, include/rumpy_amd.h).
Reference: rumpy/SISR/models/advanced/architectures.py:24-44 (CALayer), :60-84 (RCAB).

Single launches are checked against a torch restatement of their arithmetic (gate MLP, gated tile) and, behind the gated tile, BITWISE against the
strip-conv launches (tests/test_kernels_gpu.py checks those against torch fp32): the sweeps are the residual-block kernel's.  Whole networks on
this form against the oracle: tests/test_network_gpu.py (every RCAN / QRCAN test runs it - it is the default form)."""
import numpy as np
import pytest
import torch

from gpu_utils import BF16, DEV, PackedConv, assert_bf16_close, assert_f32_close, stream
from rumpy_amd import _lib as L
from tests.test_kernels_gpu import hip_conv

pytestmark = pytest.mark.gpu


def _mlp(gen, cr):
    f = lambda *s: torch.from_numpy(gen.uniform(-0.3, 0.3, s).astype(np.float32)).to(DEV)
    return f(cr, 64), f(cr), f(64, cr), f(64)


@pytest.mark.parametrize('N,H,W,cr,geo', [(2, 13, 48, 4, None), (3, 20, 37, 4, None), (8, 48, 48, 4, None), (2, 13, 64, 4, None), (1, 20, 100, 4, None),
                                          (2, 33, 70, 4, '4,2'), (2, 33, 70, 4, '8,2'), (2, 33, 70, 4, '6,3'), (2, 24, 24, 3, None), (1, 5, 9, 1, None)])
def test_rcab2_launches_against_their_arithmetic(N, H, W, cr, geo, monkeypatch):
    if geo:
        monkeypatch.setenv('RUMPY_BLOCK_GEO', geo)
    gen = np.random.default_rng(500 + H + W + cr)
    mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.06, 0.06, (64, 64, 3, 3)).astype(np.float32)),
                            torch.from_numpy(gen.uniform(-0.1, 0.1, 64).astype(np.float32)))
    pa, pb = mk(), mk()
    rnd = lambda: torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    nan = lambda: torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    lib = L.lib()
    np_out = int(lib.rumpy_rcab2_partials(N, H, W))
    cw1, cb1, cw2, cb2 = _mlp(gen, cr)
    qg = torch.from_numpy(gen.uniform(0.2, 1.0, (N, 64)).astype(np.float32)).to(DEV)
    inv_hw = 1.0 / (H * W)

    # ---------------- forward, nothing pending: t1 = relu(conv1 x + b1), u = conv2 t1 + b2 (ungated), pool partial rows
    x = rnd()
    t_ref, _ = hip_conv(x, pa, N, H, W, relu=True)
    u_ref, pool_ref = hip_conv(t_ref, pb, N, H, W, pool=True)
    t1, u1, mb = nan(), nan(), torch.zeros(N, H, W, 8, dtype=torch.uint8, device=DEV)
    part1 = torch.full((N, np_out, 64), float('nan'), dtype=torch.float32, device=DEV)
    L.call('rumpy_rcab2_fwd', L.Rcab2Args(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                                          t=t1.data_ptr(), u_out=u1.data_ptr(), part_out=part1.data_ptr(), maskbits=mb.data_ptr(), N=N, H=H, W=W, cr=cr), stream())
    torch.cuda.synchronize()
    assert torch.equal(t1, t_ref) and torch.equal(u1, u_ref)
    assert_f32_close(part1.sum(1), pool_ref.sum(1), 'pool sums', rel=1e-5)
    want = ((t1.float() > 0).reshape(N, H, W, 8, 8).to(torch.int32) << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(mb, want)

    # ---------------- forward with a pending branch: x' = x + gate(part1) * qgate * u1 on the way in
    mean = part1.sum(1) * inv_hw
    hid = torch.relu(mean @ cw1.t() + cb1)
    gate = torch.sigmoid(hid @ cw2.t() + cb2)
    xg_ref = torch.addcmul(x.float(), u1.float(), (gate * qg)[:, None, None, :])
    xg, t2, u2 = nan(), nan(), nan()
    part2 = torch.full((N, np_out, 64), float('nan'), dtype=torch.float32, device=DEV)
    mean_o, hid_o, gate_o = (torch.full(s, float('nan'), dtype=torch.float32, device=DEV) for s in ((N, 64), (N, cr), (N, 64)))
    scratch = torch.zeros(N, 64, dtype=torch.float32, device=DEV)
    mb2 = torch.zeros_like(mb)
    L.call('rumpy_rcab2_fwd', L.Rcab2Args(x=x.data_ptr(), u_in=u1.data_ptr(), part_in=part1.data_ptr(), np_in=np_out, part_out=part2.data_ptr(),
                                          part_scratch=scratch.data_ptr(), w1=pb.w_fwd.data_ptr(), b1=pb.b_packed.data_ptr(), w2=pa.w_fwd.data_ptr(),
                                          b2=pa.b_packed.data_ptr(), x_out=xg.data_ptr(), t=t2.data_ptr(), u_out=u2.data_ptr(), maskbits=mb2.data_ptr(),
                                          ca_w1=cw1.data_ptr(), ca_b1=cb1.data_ptr(), ca_w2=cw2.data_ptr(), ca_b2=cb2.data_ptr(), mean=mean_o.data_ptr(),
                                          hidden=hid_o.data_ptr(), gate=gate_o.data_ptr(), qgate=qg.data_ptr(), N=N, H=H, W=W, cr=cr), stream())
    torch.cuda.synchronize()
    assert_f32_close(mean_o, mean, 'mean', rel=1e-5)
    assert_f32_close(hid_o, hid, 'hidden', rel=1e-4)
    assert_f32_close(gate_o, gate, 'gate', rel=1e-5)
    assert_bf16_close(xg, xg_ref, 'x + gate * u', rel=3e-3)
    t2_ref, _ = hip_conv(xg, pb, N, H, W, relu=True)            # behind the gated tile (as the kernel rounded it): bitwise the strip-conv launches
    u2_ref, pool2_ref = hip_conv(t2_ref, pa, N, H, W, pool=True)
    assert torch.equal(t2, t2_ref) and torch.equal(u2, u2_ref)
    assert_f32_close(part2.sum(1), pool2_ref.sum(1), 'pool sums', rel=1e-5)

    # ---------------- backward of the FIRST block: G, partial rows of sum(G * u1), the saved gate / hidden; product rows against another tensor
    G, extra, uprev = rnd(), rnd(), rnd()
    npb = 7
    pin = torch.from_numpy(gen.standard_normal((N, npb, 64)).astype(np.float32)).to(DEV)
    ds = pin.sum(1)
    s_ = gate_o
    dz = ds * qg * s_ * (1 - s_)
    dh = (dz @ cw2) * (hid_o > 0).float()
    dp = dh @ cw1
    dzq = ds * s_ * qg * (1 - qg)
    dU_ref = G.float() * (s_ * qg)[:, None, None, :] + (dp * inv_hw)[:, None, None, :]
    dU, gt1, dx = nan(), nan(), nan()
    dz_o, dzq_o = torch.zeros(N, 64, device=DEV), torch.zeros(N, 64, device=DEV)
    pout = torch.full((N, np_out, 64), float('nan'), dtype=torch.float32, device=DEV)
    L.call('rumpy_rcab2_bwd', L.Rcab2Args(x=G.data_ptr(), u_in=uprev.data_ptr(), part_in=pin.data_ptr(), np_in=npb, part_out=pout.data_ptr(),
                                          part_scratch=scratch.data_ptr(), w1=pb.w_dgrad.data_ptr(), w2=pa.w_dgrad.data_ptr(), x_out=dU.data_ptr(),
                                          t=gt1.data_ptr(), u_out=dx.data_ptr(), res2=extra.data_ptr(), maskbits=mb.data_ptr(),
                                          ca_w1=cw1.data_ptr(), ca_b1=cb1.data_ptr(), ca_w2=cw2.data_ptr(), ca_b2=cb2.data_ptr(),
                                          hidden=hid_o.data_ptr(), gate=gate_o.data_ptr(), qgate=qg.data_ptr(), dz=dz_o.data_ptr(), dzq=dzq_o.data_ptr(),
                                          N=N, H=H, W=W, cr=cr), stream())
    torch.cuda.synchronize()
    assert_f32_close(dz_o, dz, 'dz', rel=1e-4)
    assert_f32_close(dzq_o, dzq, 'dzq', rel=1e-4)
    assert_bf16_close(dU, dU_ref, 'dU = G * gate + dp / HW', rel=3e-3)
    g1_ref, _ = hip_conv(dU, pb, N, H, W, dgrad=True, mask=t1)
    dx_ref, _ = hip_conv(g1_ref, pa, N, H, W, dgrad=True, res1=G, res2=extra)
    assert torch.equal(gt1, g1_ref) and torch.equal(dx, dx_ref)
    assert_f32_close(pout.sum(1), (dx.float() * uprev.float()).sum((1, 2)), 'sum(dx * u_prev)', rel=1e-4)
    # ... and without a previous block: no product rows are written
    pout.fill_(float('nan'))
    dx2 = nan()
    L.call('rumpy_rcab2_bwd', L.Rcab2Args(x=G.data_ptr(), part_in=pin.data_ptr(), np_in=npb, part_scratch=scratch.data_ptr(), w1=pb.w_dgrad.data_ptr(),
                                          w2=pa.w_dgrad.data_ptr(), x_out=dU.data_ptr(), t=gt1.data_ptr(), u_out=dx2.data_ptr(), res2=extra.data_ptr(),
                                          maskbits=mb.data_ptr(), ca_w1=cw1.data_ptr(), ca_b1=cb1.data_ptr(), ca_w2=cw2.data_ptr(), ca_b2=cb2.data_ptr(),
                                          hidden=hid_o.data_ptr(), gate=gate_o.data_ptr(), qgate=qg.data_ptr(), dz=dz_o.data_ptr(), N=N, H=H, W=W, cr=cr), stream())
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx) and bool(torch.isnan(pout).all())


def test_rcab2_many_partial_rows_and_argument_checks():
    """more than 64 partial rows per image (whole-image evaluation) are folded by the launch itself; what the entry points refuse"""
    gen = np.random.default_rng(77)
    N, H, W, cr = 2, 12, 20, 4
    pa = PackedConv(torch.from_numpy(gen.uniform(-0.06, 0.06, (64, 64, 3, 3)).astype(np.float32)), torch.zeros(64))
    x = torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    u = torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    cw1, cb1, cw2, cb2 = _mlp(gen, cr)
    outs = []
    for rows in (3, 200):
        part = torch.zeros(N, rows, 64, dtype=torch.float32, device=DEV)
        part[:, :3] = torch.from_numpy(np.random.default_rng(5).standard_normal((N, 3, 64)).astype(np.float32)).to(DEV) * 50
        xg, uo = torch.zeros_like(x), torch.zeros_like(x)
        mean, hid, gate, scratch = (torch.zeros(N, k, dtype=torch.float32, device=DEV) for k in (64, cr, 64, 64))
        a = L.Rcab2Args(x=x.data_ptr(), u_in=u.data_ptr(), part_in=part.data_ptr(), np_in=rows, part_scratch=scratch.data_ptr(), w1=pa.w_fwd.data_ptr(),
                        b1=pa.b_packed.data_ptr(), w2=pa.w_fwd.data_ptr(), b2=pa.b_packed.data_ptr(), x_out=xg.data_ptr(), u_out=uo.data_ptr(),
                        ca_w1=cw1.data_ptr(), ca_b1=cb1.data_ptr(), ca_w2=cw2.data_ptr(), ca_b2=cb2.data_ptr(), mean=mean.data_ptr(), hidden=hid.data_ptr(),
                        gate=gate.data_ptr(), N=N, H=H, W=W, cr=cr)
        L.call('rumpy_rcab2_fwd', a, stream())
        torch.cuda.synchronize()
        outs.append((xg, uo, gate))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    a.part_scratch = None
    assert L.lib().rumpy_rcab2_fwd(a, None) == -1 and b'part_scratch' in L.lib().rumpy_last_error()
    a.part_scratch, a.x_out = scratch.data_ptr(), None
    assert L.lib().rumpy_rcab2_fwd(a, None) == -1 and b'x_out' in L.lib().rumpy_last_error()
    a.x_out, a.fmt = xg.data_ptr(), L.FMT_F16
    assert L.lib().rumpy_rcab2_bwd(a, None) == -1
    a.fmt, a.cr = 0, 8          # attention MLPs with more than 4 hidden units run conv_rcab.hip (the engine picks the form per block)
    assert L.lib().rumpy_rcab2_fwd(a, None) == -1 and b'Cr' in L.lib().rumpy_last_error()

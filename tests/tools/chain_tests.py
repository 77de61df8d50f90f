"""Parity tests of the EXPERIMENTAL persistent-chain kernels (tests/tools/csrc -> librumpy_exp.so: measurement tools that tie with or lose
to the per-block launches).  They are not product code and not part of `pytest -m gpu` (the file name keeps
them out of the default collection); run them on the GPU box with

    python -m pytest tests/tools/chain_tests.py -q
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C                  # noqa: E402
from gpu_utils import (BF16, DEV, SPLIT_FORK, SPLIT_JOIN, SPLIT_ONE_STREAM, BlockChainArgs, BlockSplitArgs, ChainArgs, ChainLayer, PackedConv, RcabChainArgs, RcabChainBlock, assert_bf16_close, exp_call, exp_lib, hip_conv, nhwc,   # noqa: E402
                       stream, to_dev_bytes)
from rumpy_amd import _lib as L          # noqa: E402

pytestmark = pytest.mark.tools


def _rand(gen, *shape, scale=1.0):
    return torch.from_numpy(gen.standard_normal(shape).astype(np.float32) * scale)


def _wb(gen, co, ci):
    b = 1.0 / np.sqrt(ci * 9)
    return (torch.from_numpy(gen.uniform(-b, b, (co, ci, 3, 3)).astype(np.float32)),
            torch.from_numpy(gen.uniform(-b, b, (co,)).astype(np.float32)))


def _chain_reference_and_run(N, H, W, nlayers, seed):
    """Run a chain of 64->64 convs (alternating ResBlock-style epilogues) once through rumpy_conv_chain and once layer by
    layer through rumpy_conv3x3; returns the two lists of outputs."""
    gen = np.random.default_rng(seed)
    x = nhwc(_rand(gen, N, 64, H, W))
    extra_res = nhwc(_rand(gen, N, 64, H, W))
    mask_t = nhwc(_rand(gen, N, 64, H, W))
    pcs = [PackedConv(*_wb(gen, 64, 64)) for _ in range(nlayers)]
    outs_ref, cfgs = [], []
    cur = x
    for l in range(nlayers):
        kind = l % 4
        if kind == 0:
            cfg = dict(relu=True)                                   # ResBlock conv1
        elif kind == 1:
            cfg = dict(scale=0.1, res1=(outs_ref[l - 2] if l >= 2 else x))   # ResBlock conv2: + block input
        elif kind == 2:
            cfg = dict(use_bias=False, scale=0.1, mask=mask_t)      # data-gradient style
        else:
            cfg = dict(use_bias=False, res1=outs_ref[l - 2], res2=extra_res)
        o, _ = hip_conv(cur, pcs[l], N, H, W, **cfg)
        outs_ref.append(o)
        cfgs.append(cfg)
        cur = o
    # chain launch writing into fresh buffers; residual sources must be the CHAIN's own earlier outputs
    outs = [torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV) for _ in range(nlayers)]
    layers = []
    p = lambda t: None if t is None else t.data_ptr()
    for l, cfg in enumerate(cfgs):
        def remap(t):
            if t is None:
                return None
            for k, r in enumerate(outs_ref):
                if t is r:
                    return outs[k]
            return t
        use_bias = cfg.get('use_bias', True)
        layers.append(ChainLayer(w=pcs[l].w_fwd.data_ptr(), bias=(pcs[l].b_packed.data_ptr() if use_bias else None),
                                   out=outs[l].data_ptr(), mask=p(remap(cfg.get('mask'))), res1=p(remap(cfg.get('res1'))),
                                   res2=p(remap(cfg.get('res2'))), relu=1 if cfg.get('relu') else 0, scale=float(cfg.get('scale', 1.0))))
    ldev = to_dev_bytes((ChainLayer * nlayers)(*layers))
    nstrips = N * ((H + 5) // 6)
    xchg = torch.zeros(int(exp_lib().rumpy_conv_chain_xchg_bytes(nstrips)), dtype=torch.uint8, device=DEV)   # zeroed once
    status = torch.full((1,), 7, dtype=torch.int32, device=DEV)
    a = ChainArgs(x=x.data_ptr(), layers=ldev.data_ptr(), nlayers=nlayers, N=N, H=H, W=W,
                    xchg=xchg.data_ptr(), status=status.data_ptr())
    for _ in range(3):                 # repeated launches on the same exchange buffer: the epoch base moves on every call
        for o in outs:
            o.fill_(float('nan'))
        exp_call('rumpy_conv_chain', a, stream())
    torch.cuda.synchronize()
    assert int(status.item()) == 0, 'hand-off timed out: status %#x' % int(status.item())
    return outs_ref, outs


@pytest.mark.parametrize('N,H,W,nlayers', [(1, 6, 16, 2), (2, 12, 48, 5), (3, 20, 37, 8), (32, 48, 48, 33)])
def test_conv_chain_matches_layer_by_layer(N, H, W, nlayers):
    ref, got = _chain_reference_and_run(N, H, W, nlayers, 31 + H)
    for l, (r, g) in enumerate(zip(ref, got)):
        assert torch.isfinite(g.float()).all(), l
        # same bf16 operands, same fp32 products; only the order of the halo-row term in the sum differs
        assert_bf16_close(g.float(), r.float(), 'chain layer %d' % l, rel=2e-3 * (1 + l), amax=2.0 ** -6 * (1 + l))


def test_conv_chain_rejects_shapes_that_cannot_be_resident():
    t = torch.zeros(64, dtype=BF16, device=DEV)
    a = ChainArgs(x=t.data_ptr(), layers=t.data_ptr(), nlayers=1, N=1, H=6, W=49, xchg=t.data_ptr(), status=t.data_ptr())
    assert exp_lib().rumpy_conv_chain(a, None) == -1
    a = ChainArgs(x=t.data_ptr(), layers=t.data_ptr(), nlayers=1, N=64, H=48, W=48, xchg=t.data_ptr(), status=t.data_ptr())
    assert exp_lib().rumpy_conv_chain(a, None) == -1 and b'co-resident' in exp_lib().rumpy_last_error()




# ---------------------------------------------------------------------------------------------------------------------
# chain of residual blocks in one launch (conv_block_chain.hip) against one rumpy_conv_block launch per block
# ---------------------------------------------------------------------------------------------------------------------
def _block_chain_case(N, H, W, nblocks, fwd, seed):
    gen = np.random.default_rng(seed)
    mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.06, 0.06, (64, 64, 3, 3)).astype(np.float32)),
                            torch.from_numpy(gen.uniform(-0.1, 0.1, 64).astype(np.float32)))
    pcs = [(mk(), mk()) for _ in range(nblocks)]
    x = torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    masks = [torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16) for _ in range(nblocks)]
    rs = 0.1
    p = lambda z: None if z is None else z.data_ptr()

    def args(b, xin, t, out):
        pa, pb = pcs[b]
        if fwd:
            return L.BlockArgs(x=p(xin), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                               t=p(t), out=p(out), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=rs)
        return L.BlockArgs(x=p(xin), w1=pb.w_dgrad.data_ptr(), w2=pa.w_dgrad.data_ptr(), mask=p(masks[b]), t=p(t), out=p(out), N=N, H=H, W=W,
                           relu1=0, scale1=rs, scale2=1.0)
    mkbuf = lambda: torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    # reference: one launch per block
    ref_t, ref_o, cur = [], [], x
    for b in range(nblocks):
        t, o = mkbuf(), mkbuf()
        L.call('rumpy_conv_block', args(b, cur, t, o), stream())
        ref_t.append(t); ref_o.append(o); cur = o
    # chain
    ts, outs = [mkbuf() for _ in range(nblocks)], [mkbuf() for _ in range(nblocks)]
    table = (L.BlockArgs * nblocks)(*[args(b, x if b == 0 else outs[b - 1], ts[b], outs[b]) for b in range(nblocks)])
    tdev = to_dev_bytes(table)
    nstrips = N * ((H + 5) // 6)
    xchg = torch.zeros(int(exp_lib().rumpy_block_chain_xchg_bytes(nstrips)), dtype=torch.uint8, device=DEV)   # zeroed once
    status = torch.full((1,), 7, dtype=torch.int32, device=DEV)
    a = BlockChainArgs(blocks=tdev.data_ptr(), nblocks=nblocks, N=N, H=H, W=W, masked=0 if fwd else 1, xchg=xchg.data_ptr(),
                         status=status.data_ptr())
    for _ in range(3):                 # repeated launches on the same exchange buffer: the tag base moves on every call
        for o in outs + ts:
            o.fill_(float('nan'))
        exp_call('rumpy_block_chain', a, stream())
    torch.cuda.synchronize()
    assert int(status.item()) == 0, 'hand-off timed out: status %#x' % int(status.item())
    return ref_t, ref_o, ts, outs


@pytest.mark.parametrize('N,H,W,nblocks', [(1, 6, 16, 2), (2, 13, 48, 3), (3, 20, 37, 4), (1, 26, 9, 5), (32, 48, 48, 16)])
@pytest.mark.parametrize('fwd', [True, False])
def test_block_chain_matches_block_by_block(N, H, W, nblocks, fwd):
    ref_t, ref_o, ts, outs = _block_chain_case(N, H, W, nblocks, fwd, 7 + H + nblocks)
    for b in range(nblocks):
        # same operands and the same operation order per pixel: bit-identical
        assert torch.equal(ts[b], ref_t[b]), ('activation', b)
        assert torch.equal(outs[b], ref_o[b]), ('output', b)


def test_block_chain_rejects_shapes_that_cannot_be_resident():
    t = torch.zeros(64, dtype=BF16, device=DEV)
    a = BlockChainArgs(blocks=t.data_ptr(), nblocks=1, N=1, H=6, W=49, xchg=t.data_ptr(), status=t.data_ptr())
    assert exp_lib().rumpy_block_chain(a, None) == -1
    a = BlockChainArgs(blocks=t.data_ptr(), nblocks=1, N=64, H=48, W=48, xchg=t.data_ptr(), status=t.data_ptr())
    assert exp_lib().rumpy_block_chain(a, None) == -1 and b'co-resident' in exp_lib().rumpy_last_error()


@pytest.mark.parametrize('N,H,W', [(32, 48, 48), (2, 13, 48), (3, 20, 37), (1, 5, 9), (5, 7, 16), (1, 48, 48)])
def test_half_strip_block_launches_are_bitwise_the_whole_strip_launches(N, H, W):
    """rumpy_conv_block_split (tests/tools/csrc/conv_hblock.hip, experimental library): the two ResBlock forms of a training step as half-strip launches of the two halves of the
    batch - on two streams (fork / join inside the call) and on one - against rumpy_conv_block: OUT, T and the ReLU mask bytes bit for bit,
    forward and data gradient (mask bytes read, second residual operand)."""
    gen = np.random.default_rng(300 + H + W + N)
    mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.06, 0.06, (64, 64, 3, 3)).astype(np.float32)),
                            torch.from_numpy(gen.uniform(-0.1, 0.1, 64).astype(np.float32)))
    pa, pb = mk(), mk()
    x = torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV, BF16)
    extra = torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV, BF16)
    p = lambda z: None if z is None else z.data_ptr()

    def run(fwd, how, bits_in=None):
        t = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
        out = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
        bits = torch.full((N, H, W, 8), 0xAA, dtype=torch.uint8, device=DEV) if fwd else bits_in
        if fwd:
            a = L.BlockArgs(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                            mask=None, res2=None, t=p(t), out=out.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=0.1, maskbits=bits.data_ptr())
        else:
            a = L.BlockArgs(x=x.data_ptr(), w1=pb.w_dgrad.data_ptr(), b1=None, w2=pa.w_dgrad.data_ptr(), b2=None, mask=None, res2=p(extra),
                            t=p(t), out=out.data_ptr(), N=N, H=H, W=W, relu1=0, scale1=0.1, scale2=1.0, maskbits=bits.data_ptr())
        if how == 'whole':
            L.call('rumpy_conv_block', a, stream())
        else:
            sa = BlockSplitArgs(block=a, flags={'two': SPLIT_FORK | SPLIT_JOIN, 'one': SPLIT_ONE_STREAM}[how])
            exp_call('rumpy_conv_block_split', sa, stream())
        torch.cuda.synchronize()
        return t, out, bits
    for fwd in (True, False):
        bits_in = None
        if not fwd:
            bits_in = run(True, 'whole')[2]
        ref = run(fwd, 'whole', bits_in)
        for how in ('two', 'one'):
            got = run(fwd, how, bits_in)
            for i, (r, g_) in enumerate(zip(ref, got)):
                assert torch.equal(r.view(torch.int16) if r.dtype == BF16 else r, g_.view(torch.int16) if g_.dtype == BF16 else g_), (fwd, how, i)
    # a chain of launches between one fork and one join: the two halves only follow their own stream
    cur_w, cur_s = x.clone(), x.clone()
    for blk in range(4):
        nxt_w, nxt_s = torch.empty_like(x), torch.empty_like(x)
        mkargs = lambda src, dst: L.BlockArgs(x=src.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(), b2=pb.b_packed.data_ptr(),
                                              mask=None, res2=None, t=None, out=dst.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=0.1)
        L.call('rumpy_conv_block', mkargs(cur_w, nxt_w), stream())
        exp_call('rumpy_conv_block_split', BlockSplitArgs(block=mkargs(cur_s, nxt_s), flags=(SPLIT_FORK if blk == 0 else 0) | (SPLIT_JOIN if blk == 3 else 0)), stream())
        cur_w, cur_s = nxt_w, nxt_s
    torch.cuda.synchronize()
    assert torch.equal(cur_w.view(torch.int16), cur_s.view(torch.int16))
    bad = BlockSplitArgs(block=L.BlockArgs(x=x.data_ptr(), w1=pa.w_fwd.data_ptr(), w2=pb.w_fwd.data_ptr(), out=x.data_ptr(), N=1, H=8, W=64, relu1=1, scale1=1.0, scale2=1.0), flags=0)
    assert exp_lib().rumpy_conv_block_split(C.byref(bad), None) == -1 and b'W <= 48' in exp_lib().rumpy_last_error()


# ---- a run of RCABs as one persistent launch (conv_rcab_chain.hip; product library until round 6: measured slower than the per-block launches, profiles/r05_rcab_chain.txt)
def _rcab_chain_case(N, H, W, nblk, cr, hooks, seed, with_q=False):
    gen = np.random.default_rng(seed)
    f32 = lambda lo, hi, *s: torch.from_numpy(gen.uniform(lo, hi, s).astype(np.float32)).to(DEV)
    convs = [(PackedConv(f32(-0.05, 0.05, 64, 64, 3, 3).cpu(), f32(-0.1, 0.1, 64).cpu()), PackedConv(f32(-0.05, 0.05, 64, 64, 3, 3).cpu(), f32(-0.1, 0.1, 64).cpu()))
             for _ in range(nblk)]
    mlps = [(f32(-0.3, 0.3, cr, 64), f32(-0.3, 0.3, cr), f32(-0.3, 0.3, 64, cr), f32(-0.3, 0.3, 64)) for _ in range(nblk)]
    qgs = [f32(0.2, 1.0, N, 64) if with_q else None for _ in range(nblk)]
    rnd = lambda: torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
    nan = lambda: torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
    x0, dy0, extra = rnd(), rnd(), rnd()
    lib = L.lib()
    res = {}
    for form in ('blocks', 'chain'):
        t1s, t2s, ys = [nan() for _ in range(nblk)], [nan() for _ in range(nblk)], [nan() for _ in range(nblk)]
        mbs = [torch.zeros(N, H, W, 8, dtype=torch.uint8, device=DEV) for _ in range(nblk)]
        means, hids, gates = ([torch.full(s, float('nan'), device=DEV) for _ in range(nblk)] for s in ((N, 64), (N, cr), (N, 64)))
        dt2s, dt1s, dxs = [nan() for _ in range(nblk)], [nan() for _ in range(nblk)], [nan() for _ in range(nblk)]
        dzs, dzqs = [torch.zeros(N, 64, device=DEV) for _ in range(nblk)], [torch.zeros(N, 64, device=DEV) for _ in range(nblk)]
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        if form == 'blocks':
            xchg = torch.zeros(int(lib.rumpy_rcab_xchg_bytes(N, H, W)), dtype=torch.uint8, device=DEV)
            epoch = torch.zeros(1, dtype=torch.int32, device=DEV)
            common = lambda b: dict(N=N, H=H, W=W, cr=cr, ca_w1=mlps[b][0].data_ptr(), ca_b1=mlps[b][1].data_ptr(), ca_w2=mlps[b][2].data_ptr(), ca_b2=mlps[b][3].data_ptr(),
                                    hidden=hids[b].data_ptr(), gate=gates[b].data_ptr(), qgate=qgs[b].data_ptr() if with_q else None, xchg=xchg.data_ptr(),
                                    xchg_bytes=xchg.numel(), epoch=epoch.data_ptr(), status=status.data_ptr(), maskbits=mbs[b].data_ptr())
            L.check(lib.rumpy_rcab_epoch_advance(epoch.data_ptr(), stream()), 'epoch')
            for b, (pa, pb) in enumerate(convs):
                L.call('rumpy_rcab_fwd', L.RcabArgs(x=(x0 if b == 0 else ys[b - 1]).data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(),
                                                    b2=pb.b_packed.data_ptr(), t=t1s[b].data_ptr(), t2=t2s[b].data_ptr(), out=ys[b].data_ptr(), mean=means[b].data_ptr(),
                                                    seq=2 * b, **common(b)), stream())
            for k, b in enumerate(reversed(range(nblk))):
                pa, pb = convs[b]
                g_in = dy0 if k == 0 else dxs[b + 1]
                L.call('rumpy_rcab_bwd', L.RcabArgs(x=g_in.data_ptr(), w1=pb.w_dgrad.data_ptr(), w2=pa.w_dgrad.data_ptr(), t=dt1s[b].data_ptr(), t2=dt2s[b].data_ptr(),
                                                    t2_in=t2s[b].data_ptr(), mask=t1s[b].data_ptr(), res2=extra.data_ptr() if b == 0 else None, out=dxs[b].data_ptr(),
                                                    dz=dzs[b].data_ptr(), dzq=dzqs[b].data_ptr() if with_q else None, seq=2 * b + 1, **common(b)), stream())
        else:
            work = torch.zeros(int(exp_lib().rumpy_rcab_chain_work_bytes(N, H)), dtype=torch.uint8, device=DEV)
            xchg = torch.zeros(N * ((H + 5) // 6) * 512, dtype=torch.uint8, device=DEV)
            rec = lambda **kw: RcabChainBlock(**{k: (v.data_ptr() if torch.is_tensor(v) else v) for k, v in kw.items()})
            fwd = [rec(x=x0 if b == 0 else ys[b - 1], w1=pa.w_fwd, b1=pa.b_packed, w2=pb.w_fwd, b2=pb.b_packed, t=t1s[b], t2=t2s[b], out=ys[b], maskbits=mbs[b],
                       ca_w1=mlps[b][0], ca_b1=mlps[b][1], ca_w2=mlps[b][2], ca_b2=mlps[b][3], mean=means[b], hidden=hids[b], gate=gates[b],
                       qgate=qgs[b] if with_q else None) for b, (pa, pb) in enumerate(convs)]
            bwd = []
            for k, b in enumerate(reversed(range(nblk))):
                pa, pb = convs[b]
                bwd.append(rec(x=dy0 if k == 0 else dxs[b + 1], w1=pb.w_dgrad, w2=pa.w_dgrad, t=dt1s[b], t2=dt2s[b], t2_in=t2s[b], res2=extra if b == 0 else None, out=dxs[b],
                               maskbits=mbs[b], ca_w1=mlps[b][0], ca_b1=mlps[b][1], ca_w2=mlps[b][2], ca_b2=mlps[b][3], hidden=hids[b], gate=gates[b],
                               qgate=qgs[b] if with_q else None, dz=dzs[b], dzq=dzqs[b] if with_q else None))
            for recs, backward in ((fwd, 0), (bwd, 1)):
                tab = to_dev_bytes((RcabChainBlock * nblk)(*recs))
                a = RcabChainArgs(blocks=tab.data_ptr(), nblocks=nblk, N=N, H=H, W=W, cr=cr, backward=backward, work=work.data_ptr(), work_bytes=work.numel(),
                                    xchg=xchg.data_ptr(), xchg_bytes=xchg.numel(), status=status.data_ptr(), **hooks)
                for rep in range(2):
                    exp_call('rumpy_rcab_chain', a, stream())
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        assert int(status.item()) == 0, hex(int(status.item()))
        res[form] = t1s + t2s + ys + mbs + means + hids + gates + dt2s + dt1s + dxs + dzs + dzqs
    names = ['t1', 't2', 'y', 'mb', 'mean', 'hid', 'gate', 'dt2', 'dt1', 'dx', 'dz', 'dzq']
    for i, (p, q) in enumerate(zip(res['blocks'], res['chain'])):
        assert p.dtype == torch.uint8 or torch.isfinite(p.float()).all(), (names[i // nblk], i % nblk)
        assert torch.equal(p.view(torch.uint8), q.view(torch.uint8)), (names[i // nblk], i % nblk, hooks)


@pytest.mark.parametrize('hooks', [dict(fake_xcc=0, force_sc1=0), dict(fake_xcc=0, force_sc1=1), dict(fake_xcc=3, force_sc1=1)])
@pytest.mark.parametrize('N,H,W,nblk,cr,with_q', [(32, 48, 48, 5, 4, False), (3, 20, 37, 3, 4, True), (2, 5, 9, 2, 1, False), (6, 31, 24, 4, 3, True)])
def test_rcab_chain_is_bitwise_the_per_block_launches(N, H, W, nblk, cr, with_q, hooks):
    _rcab_chain_case(N, H, W, nblk, cr, hooks, 1200 + N + H, with_q)


# ---- round 6: the block chain with ONE wave per SIMD (conv_chain1.hip, rumpy_res_chain1; measured, not shipped: profiles/r06_block_body.txt): the product chain's
# cases, the same bitwise bar against one launch per block ----
from test_chain_gpu import _chain_case  # noqa: E402

@pytest.mark.parametrize('hooks', [dict(fake_xcc=0, force_sc1=0), dict(fake_xcc=3, force_sc1=1), dict(fake_xcc=16, force_sc1=1)])
@pytest.mark.parametrize('N,H,W,nblk,backward,fmt', [(32, 48, 48, 6, 0, 0), (32, 48, 48, 6, 1, 0), (5, 20, 37, 3, 0, 0), (5, 20, 37, 3, 1, 0), (3, 13, 48, 4, 0, 1),
                                                     (1, 5, 9, 2, 0, 0), (7, 31, 24, 5, 1, 0), (2, 6, 16, 2, 0, 0), (2, 6, 16, 2, 1, 0), (42, 36, 48, 16, 0, 0)])
def test_one_wave_per_simd_chain_is_bitwise_the_per_block_launches(N, H, W, nblk, backward, fmt, hooks):
    _chain_case(N, H, W, nblk, backward, fmt, hooks, 1900 + N + H, entry='rumpy_res_chain1', call=exp_call)


@pytest.mark.parametrize('backward', [0, 1])
def test_one_wave_per_simd_chain_next_to_a_foreign_kernel_that_holds_cus(backward):
    _chain_case(32, 48, 48, 8, backward, 0, dict(fake_xcc=0, force_sc1=0), 78, disturb=True, entry='rumpy_res_chain1', call=exp_call)

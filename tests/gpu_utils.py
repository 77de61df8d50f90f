"""Helpers for the -m gpu parity tests: thin wrappers that drive single C-ABI kernels with torch tensors, and the
numpy restatement of the filter packing layouts (what the MFMA fragments must contain)."""
import ctypes as C

import numpy as np
import torch

from rumpy_amd import _lib as L

BF16 = torch.bfloat16
DEV = 'cuda:0'


# ---- the experimental persistent-chain kernels (tests/tools/csrc/librumpy_exp.so: measurement tools, not the product library) ----
class ChainLayer(L._S):
    _fields_ = [('w', C.c_void_p), ('bias', C.c_void_p), ('out', C.c_void_p), ('mask', C.c_void_p), ('res1', C.c_void_p),
                ('res2', C.c_void_p), ('relu', C.c_int32), ('scale', C.c_float)]


class ChainArgs(L._S):
    _fields_ = [('x', C.c_void_p), ('layers', C.c_void_p), ('nlayers', C.c_int32), ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('xchg', C.c_void_p), ('status', C.c_void_p), ('stamps', C.c_void_p)]


class BlockChainArgs(L._S):
    _fields_ = [('blocks', C.c_void_p), ('nblocks', C.c_int32), ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('masked', C.c_int32),
                ('xchg', C.c_void_p), ('status', C.c_void_p)]


class RcabChainBlock(L._S):
    _fields_ = [(k, C.c_void_p) for k in ('x', 'w1', 'b1', 'w2', 'b2', 't', 't2', 't2_in', 'res2', 'out', 'maskbits', 'ca_w1', 'ca_b1', 'ca_w2', 'ca_b2',
                                          'mean', 'hidden', 'gate', 'qgate', 'dz', 'dzq')]


class RcabChainArgs(L._S):
    _fields_ = [('blocks', C.c_void_p), ('nblocks', C.c_int32), ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('cr', C.c_int32), ('backward', C.c_int32),
                ('work', C.c_void_p), ('work_bytes', C.c_int64), ('xchg', C.c_void_p), ('xchg_bytes', C.c_int64), ('status', C.c_void_p),
                ('fake_xcc', C.c_int32), ('force_sc1', C.c_int32)]


_exp = None


class ChainBlock(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w1', C.c_void_p), ('b1', C.c_void_p), ('w2', C.c_void_p), ('b2', C.c_void_p), ('res2', C.c_void_p),
                ('t', C.c_void_p), ('out', C.c_void_p), ('maskbits', C.c_void_p), ('scale1', C.c_float), ('scale2', C.c_float)]


class BodyChainArgs(C.Structure):
    _fields_ = [('blocks', C.c_void_p), ('nblocks', C.c_int32), ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('backward', C.c_int32),
                ('fmt', C.c_int32), ('flags', C.c_void_p), ('epoch', C.c_void_p), ('status', C.c_void_p)]


class BlockSplitArgs(C.Structure):
    _fields_ = [('block', L.BlockArgs), ('flags', C.c_int32), ('pad_', C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        for k, v in kw.items():
            setattr(self, k, v)


SPLIT_FORK, SPLIT_JOIN, SPLIT_ONE_STREAM = 1, 2, 4


def exp_lib():
    """ctypes handle of the experimental library (built by __graft_entry__.build() / make -C tests/tools/csrc)"""
    global _exp
    if _exp is None:
        import os
        path = os.environ.get('RUMPY_EXP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'csrc', 'librumpy_exp.so')
        h = C.CDLL(path)
        for name, res, args in (('rumpy_conv_chain', C.c_int, [C.POINTER(ChainArgs), C.c_void_p]), ('rumpy_conv_chain_xchg_bytes', C.c_int64, [C.c_int32]),
                                ('rumpy_block_chain', C.c_int, [C.POINTER(BlockChainArgs), C.c_void_p]), ('rumpy_block_chain_xchg_bytes', C.c_int64, [C.c_int32]),
                                ('rumpy_body_chain', C.c_int, [C.POINTER(BodyChainArgs), C.c_void_p]),
                                ('rumpy_body_chain_flag_bytes', C.c_int64, [C.c_int32, C.c_int32]),
                                ('rumpy_conv_block_split', C.c_int, [C.POINTER(BlockSplitArgs), C.c_void_p]),
                                ('rumpy_res_chain1', C.c_int, [C.POINTER(L.ResChainArgs), C.c_void_p]),
                                ('rumpy_rcab_chain', C.c_int, [C.POINTER(RcabChainArgs), C.c_void_p]), ('rumpy_rcab_chain_work_bytes', C.c_int64, [C.c_int32, C.c_int32]),
                                ('rumpy_last_error', C.c_char_p, [])):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _exp = h
    return _exp


def exp_call(name, args, stream_):
    rc = getattr(exp_lib(), name)(C.byref(args), stream_)
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, (exp_lib().rumpy_last_error() or b'?').decode()))


def stream():
    return torch.cuda.current_stream().cuda_stream


def to_dev_bytes(arr):
    raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
    return torch.from_numpy(raw).to(DEV)


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def assert_bf16_close(got, ref, what='', rel=4e-3, amax=2.0 ** -6):
    """got: bf16-rounded kernel output, ref: fp32 oracle.  bf16 rounding alone gives a relative Frobenius error of
    about 2^-9/sqrt(3) = 1.1e-3 and a max error of 2^-9 * max|ref|; anything structural is O(1)."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what + ': non-finite output'
    r = rel_err(got, ref)
    m = float((got - ref).abs().max())
    scale = float(ref.abs().max()) + 1e-30
    assert r < rel, '%s: relative error %.3e (limit %.1e), max abs %.3e / %.3e' % (what, r, rel, m, scale)
    assert m <= amax * scale, '%s: max abs error %.3e > %.3e' % (what, m, amax * scale)


def assert_f32_close(got, ref, what='', rel=2e-3):
    """fp32 results computed from bf16 operands (weight gradients, losses): only summation order differs."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what + ': non-finite output'
    r = rel_err(got, ref)
    assert r < rel, '%s: relative error %.3e (limit %.1e)' % (what, r, rel)


def bf16r(t):
    return t.to(BF16).float()


def nhwc(t):
    """[N,C,H,W] fp32 -> [N,H,W,C] bf16 on the device"""
    return t.permute(0, 2, 3, 1).contiguous().to(BF16).to(DEV)


def nchw(t):
    """[N,H,W,C] device bf16 -> [N,C,H,W] fp32 cpu"""
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------------------------------
# numpy restatement of the packed images (include/rumpy_amd.h, rumpy_pack_item)
# ------------------------------------------------------------------------------------------------------------------
def pack_ref(w, shuffle):
    """w: [Co,Ci,3,3] fp32 numpy.  Returns (fwd, dgrad) float arrays in packed order (values, before bf16 rounding)."""
    co_n, ci_n = w.shape[0], w.shape[1]
    ctn, chn = co_n // 64, ci_n // 64
    fwd = np.zeros((ctn, chn, 4, 18, 64, 8), np.float32)
    dgr = np.zeros((chn, ctn, 4, 18, 64, 8), np.float32)
    lane = np.arange(64)
    r, g = lane & 15, lane >> 4
    e = np.arange(8)
    for s in range(18):
        half, tap = s & 1, s >> 1
        ky, kx = tap // 3, tap % 3
        for wave in range(4):
            c = 16 * wave + r                                    # channel quarter `wave`, MFMA row r: [64]
            cc = (32 * half + 8 * g)[:, None] + e[None, :]       # [64,8]
            for ct in range(ctn):
                co = (4 * c + ct) if shuffle else (64 * ct + c)
                for ch in range(chn):
                    fwd[ct, ch, wave, s] = w[co[:, None], 64 * ch + cc, ky, kx]
            for ctp in range(chn):
                ci = 64 * ctp + c
                for chp in range(ctn):
                    co2 = (4 * cc + chp) if shuffle else (64 * chp + cc)
                    dgr[ctp, chp, wave, s] = w[co2, ci[:, None], 2 - ky, 2 - kx]
    return fwd, dgr


def pack_bias_ref(b, shuffle):
    n = b.shape[0]
    i = np.arange(n)
    ct, c = i >> 6, i & 63
    return b[4 * c + ct] if shuffle else b.copy()


class PackedConv:
    """Device-side packed images of one conv produced by the HIP pack kernel."""

    def __init__(self, w, b, kind=0, shuffle=False):
        self.w = w.float().contiguous().to(DEV)
        self.b = b.float().contiguous().to(DEV)
        self.cout, self.cin, self.kind, self.shuffle = w.shape[0], w.shape[1], kind, shuffle
        if kind == 0:
            n = self.cout * self.cin * 9
            self.w_fwd = torch.zeros(n, dtype=BF16, device=DEV)
            self.w_dgrad = torch.zeros(n, dtype=BF16, device=DEV)
            self.b_packed = torch.zeros(self.cout, dtype=torch.float32, device=DEV)
        else:
            self.w_fwd = torch.zeros(18 * 64 * 8, dtype=BF16, device=DEV)
            self.w_dgrad = torch.zeros(4 * 2 * 64 * 8, dtype=BF16, device=DEV)
            self.b_packed = None
        it = L.PackItem(w=self.w.data_ptr(), b=self.b.data_ptr(), w_fwd=self.w_fwd.data_ptr(), w_dgrad=self.w_dgrad.data_ptr(),
                        b_packed=(self.b_packed.data_ptr() if self.b_packed is not None else None), cout=self.cout,
                        cin=self.cin, kind=kind, shuffle=1 if shuffle else 0)
        self.items = to_dev_bytes((L.PackItem * 1)(it))
        L.check(L.lib().rumpy_pack_weights(self.items.data_ptr(), 1, stream()), 'pack')
        torch.cuda.synchronize()


def hip_conv(x, pc, N, H, W, dgrad=False, relu=False, scale=1.0, mask=None, res1=None, res2=None, pool=False,
             in_mode=0, out_mode=0, use_bias=True, grid_x=0):
    """x: device bf16 tensor in the kernel's layout.  Returns (out, pool|None)."""
    if dgrad:
        w, cin_chunks, cout_tiles, b = pc.w_dgrad, pc.cout // 64, pc.cin // 64, None
    else:
        w, cin_chunks, cout_tiles, b = pc.w_fwd, pc.cin // 64, pc.cout // 64, (pc.b_packed if use_bias else None)
    if out_mode == 1:
        out = torch.full((N, 2 * H, 2 * W, 64), float('nan'), dtype=BF16, device=DEV)
    else:
        out = torch.full((N, H, W, 64 * cout_tiles), float('nan'), dtype=BF16, device=DEV)
    tiles = int(L.lib().rumpy_conv_pool_tiles(H, W, cin_chunks))
    pl = torch.full((N, tiles, 64 * cout_tiles), float('nan'), dtype=torch.float32, device=DEV) if pool else None
    p = lambda t: None if t is None else t.data_ptr()
    a = L.ConvArgs(x=x.data_ptr(), w=w.data_ptr(), bias=p(b), out=out.data_ptr(), mask=p(mask), res1=p(res1), res2=p(res2),
                   pool=p(pl), N=N, H=H, W=W, cin_chunks=cin_chunks, cout_tiles=cout_tiles, in_mode=in_mode, out_mode=out_mode,
                   relu=1 if relu else 0, scale=float(scale), grid_x=grid_x)
    L.call('rumpy_conv3x3', a, stream())
    torch.cuda.synchronize()
    return out, pl


def hip_wgrad(jobs_spec, mt, reduce_spec, gw, gb, variant=0, shares=None):
    """jobs_spec: list of dict(x, dy, n0, n1, H, W, x_cstride, x_coff, dy_mode, dy_cstride, dy_coff);
    reduce_spec: list of dict(first_job, njobs, co_count, co_mode, co_off, ci_total, ci_off, write_bias, scale);
    shares: None (one job per workgroup, rumpy_wgrad_grouped) or the list of first-job offsets of rumpy_wgrad_shares."""
    lib = L.lib()
    sf = int(lib.rumpy_wgrad_slab_floats(mt))
    slabs = torch.full((len(jobs_spec) * sf,), float('nan'), dtype=torch.float32, device=DEV)
    jobs = []
    for k, j in enumerate(jobs_spec):
        tiles = (j['n1'] - j['n0']) * ((j['H'] + 7) // 8) * ((j['W'] + 15) // 16)
        jobs.append(L.WgradJob(x=j['x'].data_ptr(), dy=j['dy'].data_ptr(), slab=slabs.data_ptr() + 4 * k * sf, n0=j['n0'], n1=j['n1'],
                               t0=j.get('t0', 0), t1=j.get('t1', tiles),
                               H=j['H'], W=j['W'], x_cstride=j['x_cstride'], x_coff=j['x_coff'], dy_mode=j['dy_mode'],
                               dy_cstride=j['dy_cstride'], dy_coff=j['dy_coff'], mt=mt))
    jd = to_dev_bytes((L.WgradJob * len(jobs))(*jobs))
    if shares is None:
        L.check(lib.rumpy_wgrad_grouped(jd.data_ptr(), len(jobs), mt, variant, stream()), 'wgrad')
    else:
        fd = torch.tensor(shares, dtype=torch.int32, device=DEV)
        L.check(lib.rumpy_wgrad_shares(jd.data_ptr(), fd.data_ptr(), len(shares) - 1, stream()), 'wgrad shares')
    items = []
    for r in reduce_spec:
        items.append(L.ReduceItem(slab=slabs.data_ptr() + 4 * r['first_job'] * sf, slab_stride=sf, njobs=r['njobs'], mt=mt,
                                  co_count=r['co_count'], co_mode=r['co_mode'], co_off=r['co_off'], ci_total=r['ci_total'],
                                  ci_off=r['ci_off'], write_bias=r['write_bias'], scale=r['scale'], gw=gw.data_ptr(), gb=gb.data_ptr()))
    idv = to_dev_bytes((L.ReduceItem * len(items))(*items))
    L.check(lib.rumpy_wgrad_reduce(idv.data_ptr(), len(items), stream()), 'reduce')
    torch.cuda.synchronize()
    return slabs

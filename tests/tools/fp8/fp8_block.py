"""fp8 go / no-go, hardware side: the residual-block kernel with both sweeps on the block-scaled fp8 MFMA
(tests/tools/fp8/conv_block_fp8.hip) - checked against a torch emulation of its arithmetic, then timed next to conv_block_kernel.

    make -C tests/tools/fp8 && python tests/tools/fp8/fp8_block.py          (on the GPU box)

Emulation = what the kernel computes: x and the intermediate activation rounded to OCP e4m3 after division by a power-of-two per-tensor
scale, the filters likewise, exact products, fp32 accumulation; bias, ReLU, res_scale and the residual add in fp32; output rounded to bf16."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from rumpy_amd import _lib as L                                   # noqa: E402
from gpu_utils import BF16, DEV, PackedConv, stream     # noqa: E402


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

F8 = torch.float8_e4m3fn


class Fp8Args(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w1', C.c_void_p), ('b1', C.c_void_p), ('w2', C.c_void_p), ('b2', C.c_void_p), ('out', C.c_void_p),
                ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('scale2', C.c_float), ('x_scale', C.c_float), ('t_scale', C.c_float),
                ('sa1', C.c_int32), ('sa2', C.c_int32)]


def q8(t, scale):
    """real values -> the e4m3 values the hardware conversion gives for value / scale (round to nearest even), as float32"""
    return (t.float() / scale).to(F8).float()


def pow2_scale(t, target=224.0):
    """power-of-two scale that puts max |t| near `target` (e4m3 tops out at 448)"""
    return float(2.0 ** np.ceil(np.log2(float(t.abs().max()) / target)))


def pack_filter_fp8(w, scale):
    """w [64, 64, 3, 3] fp32 -> the kernel's filter image [q 4][mfma 5][lane 64][32 bytes] (conv_block_fp8.hip: P[ky], Q01, Q2)"""
    w8 = (w.float() / scale).to(F8).view(torch.uint8).numpy()          # [cout, cin, ky, kx] bytes
    img = np.zeros((4, 5, 64, 32), dtype=np.uint8)
    for q in range(4):
        for lane in range(64):
            r, g = lane & 15, lane >> 4
            co = 16 * q + r
            for ky in range(3):
                img[q, ky, lane, :16] = w8[co, 16 * g:16 * g + 16, ky, 0]
                img[q, ky, lane, 16:] = w8[co, 16 * g:16 * g + 16, ky, 1]
            img[q, 3, lane, :] = w8[co, 32 * (g & 1):32 * (g & 1) + 32, g >> 1, 2]
            img[q, 4, lane, :16] = w8[co, 16 * g:16 * g + 16, 2, 2]
    return torch.from_numpy(img).to(DEV)


def emulate(x, w1, b1, w2, b2, rs, sx, st, sw1, sw2):
    """x [N,H,W,64] bf16 (values) -> out [N,H,W,64] fp32 before the bf16 rounding, and the quantised intermediate"""
    xn = x.float().permute(0, 3, 1, 2)
    t = torch.nn.functional.conv2d(q8(xn, sx).double(), q8(w1, sw1).double(), padding=1).float() * (sx * sw1) + b1.view(1, -1, 1, 1)
    t = torch.relu(t)
    tq = q8(t, st)
    y = torch.nn.functional.conv2d(tq.double(), q8(w2, sw2).double(), padding=1).float() * (st * sw2) + b2.view(1, -1, 1, 1)
    return (xn + rs * y).permute(0, 2, 3, 1), tq.permute(0, 2, 3, 1)


def main():
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libfp8.so'))
    lib.fp8_block.restype = C.c_int
    lib.fp8_block.argtypes = [C.POINTER(Fp8Args), C.c_void_p]
    gen = np.random.default_rng(0)
    rs = 0.1

    def mk():
        return (torch.from_numpy(gen.uniform(-0.06, 0.06, (64, 64, 3, 3)).astype(np.float32)), torch.from_numpy(gen.uniform(-0.1, 0.1, 64).astype(np.float32)))

    # ---- numerics: the kernel against the emulation of its arithmetic, and both against fp32 / the bf16 kernel ----
    for (N, H, W) in ((2, 13, 48), (3, 20, 37), (1, 5, 9)):
        (w1, b1), (w2, b2) = mk(), mk()
        x = torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(BF16)
        sx, sw1, sw2 = pow2_scale(x), pow2_scale(w1), pow2_scale(w2)
        tref = torch.relu(torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w1, b1, padding=1))
        st = pow2_scale(tref)
        want, tq = emulate(x, w1, b1, w2, b2, rs, sx, st, sw1, sw2)
        exact = (x.float().permute(0, 3, 1, 2) + rs * torch.nn.functional.conv2d(tref, w2, b2, padding=1)).permute(0, 2, 3, 1)
        xd = x.to(DEV)
        out = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
        f1, f2 = pack_filter_fp8(w1, sw1), pack_filter_fp8(w2, sw2)
        b1d, b2d = b1.to(DEV), b2.to(DEV)
        a = Fp8Args(x=xd.data_ptr(), w1=f1.data_ptr(), b1=b1d.data_ptr(), w2=f2.data_ptr(), b2=b2d.data_ptr(), out=out.data_ptr(), N=N, H=H, W=W,
                    scale2=rs, x_scale=sx, t_scale=st, sa1=127 + int(np.log2(sw1)), sa2=127 + int(np.log2(sw2)))
        assert lib.fp8_block(C.byref(a), stream()) == 0
        torch.cuda.synchronize()
        got = out.float().cpu()
        d = (got - want.to(BF16).float()).abs()
        # a T value that sits on an e4m3 rounding boundary can fall to the other side under a different fp32 summation order: a few isolated
        # elements differ by one fp8 step of one T value times a weight; everything else is the bf16 rounding of the same number
        frac_off = float((d > 2.0 ** -6 * want.abs().clamp(min=1.0)).float().mean())
        rel_emul = float((got - want).norm() / want.norm())
        # the size of the fp8 effect itself, on the block's residual branch y = out - x
        ybr = lambda o: o - x.float()
        rel_fp8 = float((ybr(got) - ybr(exact)).norm() / ybr(exact).norm())
        pa, pb = PackedConv(w1, b1), PackedConv(w2, b2)
        o16 = torch.full((N, H, W, 64), float('nan'), dtype=BF16, device=DEV)
        L.call('rumpy_conv_block', L.BlockArgs(x=xd.data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(),
                                               b2=pb.b_packed.data_ptr(), out=o16.data_ptr(), N=N, H=H, W=W, relu1=1, scale1=1.0, scale2=rs), stream())
        torch.cuda.synchronize()
        rel_bf16 = float((ybr(o16.float().cpu()) - ybr(exact)).norm() / ybr(exact).norm())
        print('%2d x %2d x %2d: kernel vs emulation rel %.2e, elements beyond one bf16 step %.4f %%; residual branch vs fp32: fp8 kernel %.3e, bf16 kernel %.3e'
              % (N, H, W, rel_emul, 100 * frac_off, rel_fp8, rel_bf16))
        assert torch.isfinite(got).all() and rel_emul < 2e-3 and frac_off < 2e-3

    # ---- timing: 16 blocks of the headline shape, inference form (the activation between the convs is not stored), same box, alternating ----
    N, H, W, nb = 32, 48, 48, 16
    pcs = [(mk(), mk()) for _ in range(nb)]
    packed = [(PackedConv(*a_), PackedConv(*b_)) for a_, b_ in pcs]
    f8 = [(pack_filter_fp8(a_[0], 2.0 ** -11), a_[1].to(DEV), pack_filter_fp8(b_[0], 2.0 ** -11), b_[1].to(DEV)) for a_, b_ in pcs]
    bufs = [torch.randn(N, H, W, 64, device=DEV).to(BF16) for _ in range(nb + 1)]
    ts = [torch.empty(N, H, W, 64, dtype=BF16, device=DEV) for _ in range(nb)]

    def run_bf16(store_t):
        for b, (pa, pb) in enumerate(packed):
            L.call('rumpy_conv_block', L.BlockArgs(x=bufs[b].data_ptr(), w1=pa.w_fwd.data_ptr(), b1=pa.b_packed.data_ptr(), w2=pb.w_fwd.data_ptr(),
                                                   b2=pb.b_packed.data_ptr(), t=ts[b].data_ptr() if store_t else None, out=bufs[b + 1].data_ptr(), N=N, H=H,
                                                   W=W, relu1=1, scale1=1.0, scale2=0.1), stream())

    def run_fp8():
        for b, (w1, b1, w2, b2) in enumerate(f8):
            a = Fp8Args(x=bufs[b].data_ptr(), w1=w1.data_ptr(), b1=b1.data_ptr(), w2=w2.data_ptr(), b2=b2.data_ptr(), out=bufs[b + 1].data_ptr(), N=N, H=H,
                        W=W, scale2=0.1, x_scale=2.0 ** -5, t_scale=2.0 ** -5, sa1=127 - 11, sa2=127 - 11)
            lib.fp8_block(C.byref(a), stream())
    for rep in range(3):
        us_t = time_fn(lambda: run_bf16(True), iters=20)
        us_b = time_fn(lambda: run_bf16(False), iters=20)
        us_8 = time_fn(run_fp8, iters=20)
        print('16 residual blocks 32x48x48, us per block: bf16 training form (T stored) %6.2f | bf16 inference form %6.2f | fp8 sweeps, inference form %6.2f  (%+.1f %%)'
              % (us_t / nb, us_b / nb, us_8 / nb, 100 * (us_8 / us_b - 1)))


if __name__ == '__main__':
    main()

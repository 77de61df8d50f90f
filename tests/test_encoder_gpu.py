"""-m gpu parity of the degradation-encoder kernels: rumpy_enc_conv, rumpy_enc_pool and the LeakyReLU
epilogue of rumpy_head_fwd against plain torch fp32 convolutions on the SAME bf16-rounded operands (so the only difference is
the fp32 summation order: tolerance 2e-3 relative to the output scale before the final bf16 rounding of the result)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from rumpy_amd import _lib as L

DEV = torch.device('cuda:0')
BF16 = torch.bfloat16


def _stream():
    return torch.cuda.current_stream(DEV).cuda_stream


def _pack(w, b):
    """fp32 OIHW -> MFMA fragment image (kind 0) through the library's own packing kernel"""
    cout, cin = w.shape[:2]
    wd, bd = w.to(DEV).contiguous(), b.to(DEV).contiguous()
    wf = torch.empty(cout * cin * 9, dtype=BF16, device=DEV)
    bp = torch.empty(cout, dtype=torch.float32, device=DEV)
    item = L.PackItem(w=wd.data_ptr(), b=bd.data_ptr(), w_fwd=wf.data_ptr(), w_dgrad=None, b_packed=bp.data_ptr(), cout=cout, cin=cin, kind=0, shuffle=0)
    tab = torch.from_numpy(np.frombuffer(bytes((L.PackItem * 1)(item)), dtype=np.uint8).copy()).to(DEV)
    L.check(L.lib().rumpy_pack_weights(tab.data_ptr(), 1, _stream()), 'pack')
    torch.cuda.synchronize()
    return wf, bp


@pytest.mark.parametrize('N,H,W,cin,cout,stride', [
    (2, 48, 48, 64, 64, 1), (2, 48, 48, 64, 128, 2), (3, 24, 24, 128, 128, 1), (3, 24, 24, 128, 256, 2), (4, 12, 12, 256, 256, 1),
    (1, 37, 53, 64, 128, 2), (1, 19, 27, 128, 128, 1), (2, 5, 3, 64, 64, 2), (1, 1, 1, 64, 64, 1),       # ragged / odd / degenerate sizes
])
def test_enc_conv_against_torch(N, H, W, cin, cout, stride):
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + cin + stride)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    xb, wb = x.to(BF16).float(), w.to(BF16).float()
    ref = F.leaky_relu(F.conv2d(xb, wb, b, stride=stride, padding=1), 0.1)
    wf, bp = _pack(w, b)
    xd = xb.permute(0, 2, 3, 1).contiguous().to(DEV, BF16)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    assert ref.shape[2:] == (Ho, Wo)
    out = torch.full((N, Ho, Wo, cout), float('nan'), dtype=BF16, device=DEV)
    L.call('rumpy_enc_conv', L.EncConvArgs(x=xd.data_ptr(), w=wf.data_ptr(), bias=bp.data_ptr(), out=out.data_ptr(), N=N, H=H, W=W,
                                           cin=cin, cout=cout, stride=stride, neg_slope=0.1), _stream())
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    scale = float(ref.abs().max())
    # bf16 result rounding (2^-9 relative) + summation order
    assert float((got - ref).abs().max()) <= 2e-3 * scale + 2 ** -8 * scale, float((got - ref).abs().max())
    # pooling of that map
    pooled = torch.zeros(N, cout, device=DEV)
    L.check(L.lib().rumpy_enc_pool(out.data_ptr(), pooled.data_ptr(), N, Ho * Wo, cout, 0, _stream()), 'pool')
    torch.cuda.synchronize()
    assert torch.allclose(pooled.cpu(), out.float().cpu().mean((1, 2)), atol=1e-5, rtol=1e-5)


def test_enc_conv_refuses_unsupported_shapes():
    x = torch.zeros(1, 4, 4, 64, dtype=BF16, device=DEV)
    for bad in (dict(cin=32, cout=64, stride=1), dict(cin=64, cout=96, stride=1), dict(cin=64, cout=64, stride=3)):
        a = L.EncConvArgs(x=x.data_ptr(), w=x.data_ptr(), bias=x.data_ptr(), out=x.data_ptr(), N=1, H=4, W=4, neg_slope=0.1, **bad)
        assert L.lib().rumpy_enc_conv(a, _stream()) != 0


def test_head_conv_leaky_epilogue():
    g = torch.Generator().manual_seed(11)
    N, H, W = 2, 20, 28
    x = torch.rand(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.3
    b = torch.randn(64, generator=g) * 0.1
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    outs = {}
    for name, sm1 in (('none', 0.0), ('leaky', -0.9), ('relu', -1.0)):
        o = torch.empty(N, H, W, 64, dtype=BF16, device=DEV)
        L.call('rumpy_head_fwd', L.HeadFwdArgs(x=xd.data_ptr(), w=wd.data_ptr(), b=bd.data_ptr(), out=o.data_ptr(), N=N, C=3, H=H, W=W, cout=64,
                                               neg_slope_m1=sm1), _stream())
        outs[name] = o.float().cpu().permute(0, 3, 1, 2)
    torch.cuda.synchronize()
    ref = F.conv2d(x, w, b, padding=1)
    for name, r in (('none', ref), ('leaky', F.leaky_relu(ref, 0.1)), ('relu', F.relu(ref))):
        assert float((outs[name] - r).abs().max()) <= 2 ** -8 * float(r.abs().max()) + 1e-5, name

"""A/B of the encoder's plain 3x3 conv kernel (rumpy_enc_conv) against the SR path's rumpy_conv3x3 on the encoder's stride-1 shapes.
    python tests/tools/enc_conv_ab.py [N]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rumpy_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
s = torch.cuda.current_stream(dev).cuda_stream
for (cin, cout, hw) in ((64, 64, 48), (64, 128, 48), (256, 128, 24), (256, 256, 12), (256, 256, 24)):
    g = torch.Generator().manual_seed(cin + cout)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev)
    b = torch.zeros(cout, device=dev)
    wf = torch.empty(cout * cin * 9, dtype=BF16, device=dev)
    bp = torch.empty(cout, device=dev)
    item = L.PackItem(w=w.data_ptr(), b=b.data_ptr(), w_fwd=wf.data_ptr(), w_dgrad=None, b_packed=bp.data_ptr(), cout=cout, cin=cin, kind=0, shuffle=0)
    tab = torch.from_numpy(np.frombuffer(bytes((L.PackItem * 1)(item)), dtype=np.uint8).copy()).to(dev)
    L.check(L.lib().rumpy_pack_weights(tab.data_ptr(), 1, s), 'pack')
    x = torch.randn(N, hw, hw, cin, generator=g).to(dev, BF16)
    o1 = torch.empty(N, hw, hw, cout, dtype=BF16, device=dev)
    o2 = torch.empty_like(o1)
    a1 = L.EncConvArgs(x=x.data_ptr(), w=wf.data_ptr(), bias=bp.data_ptr(), out=o1.data_ptr(), N=N, H=hw, W=hw, cin=cin, cout=cout, stride=1, neg_slope=1.0)
    a2 = L.ConvArgs(x=x.data_ptr(), w=wf.data_ptr(), bias=bp.data_ptr(), out=o2.data_ptr(), mask=None, res1=None, res2=None, pool=None, N=N, H=hw, W=hw,
                    cin_chunks=cin // 64, cout_tiles=cout // 64, in_mode=0, out_mode=0, relu=0, scale=1.0, grid_x=0, fmt=0)
    res = []
    for name, args in (('rumpy_enc_conv', a1), ('rumpy_conv3x3', a2)):
        for _ in range(3):
            L.call(name, args, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.call(name, args, s)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    flop = 2.0 * N * hw * hw * cin * cout * 9
    print('%3d -> %3d @%2d N=%d: enc_conv %7.1f us (%5.0f TF/s)   conv3x3 %7.1f us (%5.0f TF/s)   max |diff| %.3g' %
          (cin, cout, hw, N, res[0], flop / res[0] / 1e6, res[1], flop / res[1] / 1e6, float((o1.float() - o2.float()).abs().max())))

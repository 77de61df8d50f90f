"""CPU emulation of the HIP path's roundings (bf16 MFMA operands, fp32 accumulation, bf16 or split trunk storage) on top of
the oracle's modules - a design tool: predicts the forward self-PSNR of a storage scheme before a kernel is written.

    python tests/tools/precision_sim.py [rcan|edsr] [seed] [lr_hw]

trunk modes: 'bf16'  = every block / group / body output stored as bf16 (round 1);
             'split' = trunk stored as bf16 hi + bf16 lo (conv operand = hi, residual operand = hi + lo);
             'fp32'  = trunk kept in fp32 (conv operand = bf16(trunk)).
"""
import sys
import os

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import sr_oracle as O  # noqa: E402


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def store(t, mode):
    if mode == 'bf16':
        return rb(t)
    if mode == 'split':
        hi = rb(t)
        return hi + rb(t - hi)
    return t


def conv(x, m):
    """bf16 operands, fp32 accumulate"""
    return F.conv2d(rb(x), rb(m.weight), m.bias, padding=1)


def sim_forward(net, x, mode, inner='bf16'):
    """x fp32 NCHW; emulates engine.py's forward plan.  inner: storage of non-trunk tensors (t1)"""
    with torch.no_grad():
        a0 = store(F.conv2d(x, net.head[0].weight, net.head[0].bias, padding=1), mode)     # head is exact fp32
        cur = a0
        body = list(net.body)
        for m in body[:-1]:
            if isinstance(m, O.ScaledResidualBlock):
                t1 = rb(F.relu(conv(cur, m.body[0])))
                cur = store(cur + m.res_scale * conv(t1, m.body[2]), mode)
            else:   # AttentionGroup
                gin = cur
                for b in list(m.body)[:-1]:
                    t1 = rb(F.relu(conv(cur, b.body[0])))
                    t2 = conv(t1, b.body[2])
                    gate = b.body[3].conv_du(t2.mean((2, 3), keepdim=True))
                    cur = store(cur + gate * t2, mode)
                cur = store(conv(cur, m.body[-1]) + gin, mode)
        r = store(conv(cur, body[-1]) + a0, mode)
        u = r
        for m in net.tail[0]:
            u = rb(conv(u, m)) if isinstance(m, torch.nn.Conv2d) else m(u)
        return conv(u, net.tail[1])


def self_psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 100.0 if mse == 0 else 10 * np.log10(1.0 / mse)


if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else 'rcan'
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else (402 if name == 'rcan' else 401)
    hw = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    torch.set_num_threads(8)
    net = O.build_oracle(name, scale=4)
    net.load_state_dict(O.seeded_state_dict(net, seed))
    x, _ = O.synthetic_batch(1235, 1, lr_hw=hw, scale=4)
    with torch.no_grad():
        ref = net(x)
    print('%s seed %d: output rms %.3f' % (name, seed, float(ref.pow(2).mean().sqrt())))
    for mode in ('bf16', 'split', 'fp32'):
        out = sim_forward(net, x, mode)
        print('  trunk %-5s: self-PSNR %.2f dB, max abs %.2e' % (mode, self_psnr(out, ref), float((out - ref).abs().max())))

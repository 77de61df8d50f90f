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


# ---------------------------------------------------------------------------------------------------------------------------------
# training step: the same storage roundings in the forward AND the backward direction (every stored activation gradient is bf16)
class _Round(torch.autograd.Function):
    """bf16 rounding of a stored tensor: the value in the forward pass, its gradient in the backward pass"""

    @staticmethod
    def forward(ctx, t, fwd, bwd):
        ctx.bwd = bwd
        return rb(t) if fwd else t

    @staticmethod
    def backward(ctx, g):
        return (rb(g) if ctx.bwd else g), None, None


def R(t, fwd=True, bwd=True):
    return _Round.apply(t, fwd, bwd)


def convg(x, m):
    """forward: bf16 operands; backward: the incoming gradient is a stored bf16 tensor, weight gradient fp32 from bf16 operands"""
    return F.conv2d(x, R(m.weight, True, False), m.bias, padding=1)


def sim_train_grads(net, x, y, trunk_bwd=True):
    """L1 train step of engine.py's plan with its roundings; returns {name: grad}.  trunk_bwd=False keeps the trunk GRADIENT in fp32"""
    net.zero_grad()
    a0 = R(F.conv2d(x, net.head[0].weight, net.head[0].bias, padding=1), True, trunk_bwd)
    cur = a0
    body = list(net.body)
    for m in body[:-1]:
        if isinstance(m, O.ScaledResidualBlock):
            t1 = R(F.relu(convg(cur, m.body[0])))
            cur = R(cur + m.res_scale * convg(t1, m.body[2]), True, trunk_bwd)
        else:
            gin = cur
            for b in list(m.body)[:-1]:
                t1 = R(F.relu(convg(cur, b.body[0])))
                t2 = R(convg(t1, b.body[2]), False, True)          # forward: fp32 accumulators; backward: d_t2 stored bf16
                gate = b.body[3].conv_du(t2.mean((2, 3), keepdim=True))
                cur = R(cur + gate * t2, True, trunk_bwd)
            cur = R(convg(cur, m.body[-1]) + gin, True, trunk_bwd)
    r = R(convg(cur, body[-1]) + a0)
    u = r
    for m in net.tail[0]:
        u = R(convg(u, m)) if isinstance(m, torch.nn.Conv2d) else m(u)
    out = convg(u, net.tail[1])
    loss = (out - y).abs().mean()
    loss.backward()
    return {k: p.grad.clone() for k, p in net.named_parameters()}, float(loss)


def grad_report(name='rcan', seed=522, N=2, hw=48, **kw):
    net = O.build_oracle(name, scale=4, **kw)
    net.load_state_dict(O.seeded_state_dict(net, seed))
    x, y = O.synthetic_batch(seed + 1000, N, lr_hw=hw, scale=4)
    net.zero_grad()
    (net(x) - y).abs().mean().backward()
    ref = {k: p.grad.clone() for k, p in net.named_parameters()}
    for tb in (True, False):
        g, _ = sim_train_grads(net, x, y, trunk_bwd=tb)
        rows = sorted(((float((g[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)), k) for k in ref), reverse=True)
        allg, allr = torch.cat([g[k].reshape(-1) for k in ref]), torch.cat([ref[k].reshape(-1) for k in ref])
        print('%s N=%d %dx%d, gradient trunk %s: whole-gradient rel %.3e, median %.3e, worst %s' % (
            name, N, hw, hw, 'bf16' if tb else 'fp32', float((allg - allr).norm() / allr.norm()), float(np.median([r[0] for r in rows])),
            ', '.join('%s %.2e' % (k, r) for r, k in rows[:3])))


# ---------------------------------------------------------------------------------------------------------------------------------
# fp8 (BASELINE config 5 names "fp8 MFMA conv"): what e4m3 operands would cost the evaluation PSNR of the G18 model
def fp8_report():
    """python -c "import sys; sys.path.insert(0, 'tests/tools'); import precision_sim as P; P.fp8_report()" """
    global rb

    def f8(t):      # e4m3 with a per-tensor power-of-two scale (amax -> below 240)
        a = float(t.abs().max())
        if a == 0:
            return t
        sc = 2.0 ** np.floor(np.log2(240.0 / a))
        return (t * sc).to(torch.float8_e4m3fn).float() / sc

    def h16(t):
        return t.to(torch.float16).float()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'golden', 'g18_rcan_psnr.npz'))
    to_t = lambda a: torch.from_numpy(a.transpose(2, 0, 1).astype(np.float32) / 255.)[None]
    lt, ht = to_t(g['lr']), to_t(g['hr'])
    net = O.build_oracle('rcan', scale=4)
    net.load_state_dict(O.interpolating_state_dict(net, 502))
    with torch.no_grad():
        ref = net(lt)
    hy = O.rgb_to_ycbcr_jpg(O.clip01(ht.numpy())[0])
    yp = lambda o: O.psnr(O.rgb_to_ycbcr_jpg(O.clip01(o.numpy())[0])[0], hy[0])
    keep, rb = rb, f8
    o = sim_forward(net, lt, 'bf16')
    rb = keep
    print('G18 RCAN, e4m3 storage + MFMA operands everywhere (per-tensor scale): self-PSNR %.2f dB, dPSNR %+.3f dB (reference %.2f dB)'
          % (self_psnr(o, ref), yp(o) - yp(ref), yp(ref)))
    with torch.no_grad():
        c8 = lambda x, m: F.conv2d(f8(x), f8(m.weight), m.bias, padding=1)
        c16 = lambda x, m: F.conv2d(h16(x), h16(m.weight), m.bias, padding=1)
        a0 = h16(F.conv2d(lt, net.head[0].weight, net.head[0].bias, padding=1))
        cur = a0
        body = list(net.body)
        for m in body[:-1]:
            gin = cur
            for b in list(m.body)[:-1]:
                t1 = f8(F.relu(c8(cur, b.body[0])))
                t2 = c8(t1, b.body[2])
                cur = h16(cur + b.body[3].conv_du(t2.mean((2, 3), keepdim=True)) * t2)
            cur = h16(c16(cur, m.body[-1]) + gin)
        u = h16(c16(cur, body[-1]) + a0)
        for m in net.tail[0]:
            u = h16(c16(u, m)) if isinstance(m, torch.nn.Conv2d) else m(u)
        o = c16(u, net.tail[1])
    print('G18 RCAN, e4m3 MFMA operands inside the RCABs only, fp16 trunk / upsampler / tail: self-PSNR %.2f dB, dPSNR %+.3f dB'
          % (self_psnr(o, ref), yp(o) - yp(ref)))

"""fp8 go / no-go, accuracy side (CPU, torch): what e4m3 MFMA operands inside the residual blocks / RCABs do to a
TRAINING step.  The emulated arithmetic is the one tests/tools/fp8/conv_block_fp8.hip runs on the hardware (that kernel agrees with this
emulation to the bf16 rounding of its output, fp8_block.py): operands rounded to OCP e4m3 after a power-of-two scale, exact products, fp32
accumulation.  Everything outside the block convs - trunk, head, upsampler, tail, weight gradients, channel attention - keeps the bf16
roundings of the product path (tests/tools/precision_sim.py), so the difference between the 'bf16' and the 'fp8' rows is the fp8 operands alone.

    python tests/tools/fp8/fp8_train_sim.py grads      one-step gradient error against the fp32 oracle (EDSR-baseline, RCAN 10 x 20)
    python tests/tools/fp8/fp8_train_sim.py traj       loss trajectory on the learnable task of tests/tools/trajectory.py

Scale granularities: 'tensor' = one power-of-two scale per tensor (delayed-scaling style; what the timing kernel used);
'block' = MX: one e8m0 scale per 32 consecutive channels of a pixel (activations, gradients) / of a filter tap row (weights)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
from oracle import sr_oracle as O      # noqa: E402
import precision_sim as P              # noqa: E402

F8 = torch.float8_e4m3fn
F8G = torch.float8_e5m2
MODE = {'gran': 'tensor', 'grad_fmt': 'e5m2'}


def _q(t, fmt, amax_dims, top):
    """round to fp8 with power-of-two scales: amax over `amax_dims` (None = whole tensor) is placed below `top`"""
    a = t.abs().amax(dim=amax_dims, keepdim=True) if amax_dims is not None else t.abs().max()
    sc = torch.exp2(torch.floor(torch.log2(top / a.clamp(min=1e-30))))
    return (t * sc).to(fmt).float() / sc


def q_act(t):        # [N, C, H, W]
    if MODE['gran'] == 'tensor':
        return _q(t, F8, None, 448.0)
    n, c, h, w = t.shape
    return _q(t.reshape(n, c // 32, 32, h, w), F8, 2, 448.0).reshape(n, c, h, w)


def q_grad(t):
    fmt, top = (F8G, 57344.0) if MODE['grad_fmt'] == 'e5m2' else (F8, 448.0)
    if MODE['gran'] == 'tensor':
        return _q(t, fmt, None, top)
    n, c, h, w = t.shape
    return _q(t.reshape(n, c // 32, 32, h, w), fmt, 2, top).reshape(n, c, h, w)


def q_w(w, along_cin=True):      # [Cout, Cin, 3, 3]: blocks of 32 along the contraction axis (cin forward, cout in the data gradient)
    if MODE['gran'] == 'tensor':
        return _q(w, F8, None, 448.0)
    co, ci, kh, kw = w.shape
    if along_cin:
        return _q(w.reshape(co, ci // 32, 32, kh, kw), F8, 2, 448.0).reshape(w.shape)
    return _q(w.reshape(co // 32, 32, ci, kh, kw), F8, 1, 448.0).reshape(w.shape)


class Conv8(torch.autograd.Function):
    """3x3 conv with e4m3 MFMA operands forward and in the data gradient; weight gradient from the bf16-stored operands (product path)"""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.conv2d(q_act(x), q_w(w), b, padding=1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = F.conv_transpose2d(q_grad(g), q_w(w, along_cin=False), padding=1)
        gw = torch.nn.grad.conv2d_weight(P.rb(x), w.shape, P.rb(g), padding=1)
        return gx, gw, g.sum((0, 2, 3))


def conv8(x, m):
    return Conv8.apply(x, m.weight, m.bias)


def train_grads(net, x, y, fp8):
    """precision_sim.sim_train_grads with the two convs of every residual block / RCAB on fp8 operands when `fp8`"""
    R, convg = P.R, P.convg
    cb = conv8 if fp8 else convg
    net.zero_grad()
    a0 = R(F.conv2d(x, net.head[0].weight, net.head[0].bias, padding=1))
    cur = a0
    body = list(net.body)
    for m in body[:-1]:
        if isinstance(m, O.ScaledResidualBlock):
            t1 = R(F.relu(cb(cur, m.body[0])))
            cur = R(cur + m.res_scale * cb(t1, m.body[2]))
        else:
            gin = cur
            for b in list(m.body)[:-1]:
                t1 = R(F.relu(cb(cur, b.body[0])))
                t2 = R(cb(t1, b.body[2]), False, True)
                gate = b.body[3].conv_du(t2.mean((2, 3), keepdim=True))
                cur = R(cur + gate * t2)
            cur = R(convg(cur, m.body[-1]) + gin)
    r = R(convg(cur, body[-1]) + a0)
    u = r
    for m in net.tail[0]:
        u = R(convg(u, m)) if isinstance(m, torch.nn.Conv2d) else m(u)
    out = convg(u, net.tail[1])
    loss = (out - y).abs().mean()
    loss.backward()
    return {k: p.grad.clone() for k, p in net.named_parameters()}, float(loss)


def report(name, seed, N, hw, **kw):
    net = O.build_oracle(name, scale=4, **kw)
    net.load_state_dict(O.seeded_state_dict(net, seed))
    x, y = O.synthetic_batch(seed + 1000, N, lr_hw=hw, scale=4)
    net.zero_grad()
    (net(x) - y).abs().mean().backward()
    ref = {k: p.grad.clone() for k, p in net.named_parameters()}
    conv_keys = [k for k in ref if 'conv_du' not in k and ref[k].dim() == 4]
    rows = [('bf16 (product path)', False, None, None)] + [('fp8 e4m3 fwd / %s grad, scale per %s' % (gf, gr), True, gr, gf)
                                                           for gr in ('tensor', 'block') for gf in ('e5m2', 'e4m3')]
    for label, fp8, gr, gf in rows:
        if fp8:
            MODE['gran'], MODE['grad_fmt'] = gr, gf
        g, loss = train_grads(net, x, y, fp8)
        rel = sorted(((float((g[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)), k) for k in conv_keys), reverse=True)
        allg, allr = torch.cat([g[k].reshape(-1) for k in ref]), torch.cat([ref[k].reshape(-1) for k in ref])
        cos = float((allg @ allr) / (allg.norm() * allr.norm()))
        print('%-14s N=%d %dx%d  %-46s whole gradient rel %.3e (cos %.5f), 3x3-conv tensors: median %.3e, worst %.3e (%s), beyond 3e-2: %d of %d'
              % (name + ' ' + 'x'.join(str(v) for v in kw.values()), N, hw, hw, label, float((allg - allr).norm() / allr.norm()), cos,
                 float(np.median([r[0] for r in rel])), rel[0][0], rel[0][1], sum(1 for r in rel if r[0] > 3e-2), len(rel)), flush=True)


def trajectory(name='rcan', steps=60):
    """fp32 oracle / bf16 product-path emulation / fp8 emulations on the learnable task of tests/tools/trajectory.py, same batches, Adam lr 2e-4"""
    kw = {'edsr': dict(scale=2, num_blocks=4, res_scale=0.1), 'rcan': dict(scale=2, n_resgroups=2, n_resblocks=3, reduction=16)}[name]
    gen = torch.Generator().manual_seed(3)
    base = F.interpolate(torch.rand(8, 3, 12, 12, generator=gen), size=(48, 48), mode='bicubic', align_corners=False).clamp(0, 1)
    lr_img = F.avg_pool2d(base, 2)
    batches = []
    for s in range(steps):
        idx = torch.randperm(8, generator=gen)[:4]
        batches.append((lr_img[idx].contiguous(), base[idx].contiguous()))
    torch.manual_seed(8)
    net0 = O.build_oracle(name, **kw)
    sd0 = {k: v.clone() for k, v in net0.state_dict().items()}
    curves = {}
    for label, how in (('fp32', None), ('bf16', (False, None, None)), ('fp8 tensor e5m2', (True, 'tensor', 'e5m2')), ('fp8 block e5m2', (True, 'block', 'e5m2')),
                       ('fp8 block e4m3', (True, 'block', 'e4m3'))):
        net = O.build_oracle(name, **kw)
        net.load_state_dict(sd0)
        opt = torch.optim.Adam(net.parameters(), lr=2e-4)
        losses = []
        for x, y in batches:
            if how is None:
                opt.zero_grad()
                loss = (net(x) - y).abs().mean()
                loss.backward()
                losses.append(float(loss))
            else:
                if how[0]:
                    MODE['gran'], MODE['grad_fmt'] = how[1], how[2]
                g, l = train_grads(net, x, y, how[0])
                for k, p in net.named_parameters():
                    p.grad = g[k]
                losses.append(l)
            opt.step()
        curves[label] = np.array(losses)
    ref = curves['fp32']
    print('%s, %d steps: loss at steps 0 / %d / %d; largest relative difference to the fp32 run over all steps' % (name, steps, steps // 2, steps - 1))
    for label, c in curves.items():
        print('  %-18s %.5f  %.5f  %.5f   max |dloss| / loss %.4f' % (label, c[0], c[steps // 2], c[-1], float(np.max(np.abs(c - ref) / ref))))


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'grads'
    torch.set_num_threads(8)
    if what == 'grads':
        report('edsr', 521, 2, 48)
        report('rcan', 522, 1, 48)
    elif what == 'small':
        report('rcan', 522, 2, 24, n_resgroups=2, n_resblocks=4)
    else:
        trajectory(sys.argv[2] if len(sys.argv) > 2 else 'rcan', int(sys.argv[3]) if len(sys.argv) > 3 else 60)

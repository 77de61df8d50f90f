"""Winograd F(2x2, 3x3) go / no-go for the 64 -> 64 sweeps of the residual-block kernels, accuracy side (CPU, torch):
the one-step gradient of EDSR-baseline (16 blocks) and RCAN 10 x 20 with both convs of every residual block / RCAB evaluated as
    Y = A^T [ (G g G^T) . (B^T d B) ] A      per 2 x 2 output tile (4 x 4 input patch, stride 2),
transforms in fp32, the TRANSFORMED operands rounded to bf16 (what the matrix pipe would read), exact products, fp32 accumulation and output
transform - forward and in the data gradient (transposed, flipped filter); weight gradients and everything outside the block convs on the
product path's bf16 roundings (tests/tools/precision_sim.py), like tests/tools/fp8/fp8_train_sim.py.

    python tests/tools/winograd_sim.py

The structural side of the decision is in DESIGN.md 4.2 item 11: per 2 x 2 outputs the matrix pipe does 16 instead of 36 MACs per channel pair
(7.3 -> 3.3 us of pipe time per launch), but the transformed input of a strip is FOUR bf16 values per pixel and channel - the 10 x 50-pixel
input tile and the 8 x 50 intermediate tile of one workgroup become 256 KB + 205 KB of LDS (160 KB exist) - and the transforms are 8 + 6 VALU
operations per pixel and channel on a kernel whose epilogues are already issue-bound."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools', 'fp8'))
from oracle import sr_oracle as O      # noqa: E402
import precision_sim as P              # noqa: E402
import fp8_train_sim as S              # noqa: E402

BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
rb = lambda t: t.to(torch.bfloat16).float()


def winograd_conv(x, w, b):
    """3x3 same-padding conv of [N, C, H, W] (H, W even) by F(2x2, 3x3) with bf16 transformed operands"""
    N, C, H, W = x.shape
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)                    # [N, C, H/2, W/2, 4, 4]
    V = rb(torch.einsum('ai,ncthij,bj->ncthab', BT, d, BT))
    U = rb(torch.einsum('ai,ocij,bj->ocab', G, w, G))
    M = torch.einsum('ocab,ncthab->nothab', U.double(), V.double()).float()
    Y = torch.einsum('ia,nothab,jb->nothij', AT, M, AT)                            # [N, O, H/2, W/2, 2, 2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], H, W)
    return y if b is None else y + b.view(1, -1, 1, 1)


class ConvW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return winograd_conv(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = winograd_conv(g, w.transpose(0, 1).flip(2, 3), None)
        gw = torch.nn.grad.conv2d_weight(P.rb(x), w.shape, P.rb(g), padding=1)
        return gx, gw, g.sum((0, 2, 3))


def main():
    torch.set_num_threads(8)
    # self-check of the transform algebra in fp64-exact operands
    x, w = torch.randn(1, 8, 12, 12), torch.randn(8, 8, 3, 3)
    global rb
    keep, rb = rb, (lambda t: t)
    assert float((winograd_conv(x, w, None) - F.conv2d(x, w, padding=1)).abs().max()) < 1e-4
    rb = keep
    S.conv8 = lambda xx, m: ConvW.apply(xx, m.weight, m.bias)
    for name, seed, N in (('edsr', 521, 2), ('rcan', 522, 1)):
        net = O.build_oracle(name, scale=4)
        net.load_state_dict(O.seeded_state_dict(net, seed))
        x, y = O.synthetic_batch(seed + 1000, N, lr_hw=48, scale=4)
        net.zero_grad()
        (net(x) - y).abs().mean().backward()
        ref = {k: p.grad.clone() for k, p in net.named_parameters()}
        conv_keys = [k for k in ref if 'conv_du' not in k and ref[k].dim() == 4]
        for label, wino in (('bf16 direct (product path)', False), ('Winograd F(2x2,3x3), bf16 transformed operands', True)):
            g, loss = S.train_grads(net, x, y, wino)
            rel = sorted(((float((g[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)), k) for k in conv_keys), reverse=True)
            allg, allr = torch.cat([g[k].reshape(-1) for k in ref]), torch.cat([ref[k].reshape(-1) for k in ref])
            print('%-5s N=%d 48x48  %-48s whole gradient rel %.3e, 3x3-conv tensors: median %.3e, worst %.3e (%s), beyond 3e-2: %d of %d'
                  % (name, N, label, float((allg - allr).norm() / allr.norm()), float(np.median([r[0] for r in rel])), rel[0][0], rel[0][1],
                     sum(1 for r in rel if r[0] > 3e-2), len(rel)), flush=True)


if __name__ == '__main__':
    main()

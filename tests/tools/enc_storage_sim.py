"""CPU simulation of the encoder trunk's 16-bit storage points: gradient error against the fp32 graph for bf16 / fp16 / fp32
filters, conv outputs and stage outputs (forward) and bf16 / fp16 gradients - what decided the fp16 forward storage of round 3.
    python tests/tools/enc_storage_sim.py"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from oracle import sr_oracle as O, contrastive_oracle as CO
torch.set_num_threads(4)
class Pt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, fd, bd):
        ctx.bd=bd
        return t.to(fd).float() if fd is not None else t
    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.bd).float() if ctx.bd is not None else g), None, None
B,H_,F32=torch.bfloat16, torch.float16, None
def trunk(enc, x, wf, zf, af, gb):
    a=x
    for i in range(6):
        conv,bn=enc.E[3*i],enc.E[3*i+1]
        w=conv.weight if i==0 else Pt.apply(conv.weight, wf, None)
        z=Pt.apply(F.conv2d(a,w,conv.bias,stride=conv.stride,padding=1), zf, gb)
        a=Pt.apply(F.leaky_relu(bn(z),0.1), af, gb if i<5 else None)
    return a.mean((2,3))
def grads(enc, x, r1, r2, cfg):
    enc.zero_grad(); enc.train()
    fea = enc.E(x).squeeze(-1).squeeze(-1) if cfg is None else trunk(enc,x,*cfg)
    q=enc.mlp(fea)
    ((fea*r1).sum()+(q*r2).sum()).backward()
    return {k:p.grad.clone() for k,p in enc.named_parameters()}
ZB={'E.%d.bias'%i for i in (0,3,6,9,12,15)}
for N,hw,seed in ((8,32,308),(3,48,303)):
    enc=O.OracleEncoder(); enc.load_state_dict(O.seeded_encoder_state(O.OracleEncoder(), seed))
    x=CO.contrastive_batch(310+N,N,1,hw=hw)[:,0].contiguous()
    g=torch.Generator().manual_seed(N); r1,r2=torch.randn(N,256,generator=g),torch.randn(N,256,generator=g)
    ref=grads(enc,x,r1,r2,None)
    for name,cfg in (('bf16 all (today)',(B,B,B,B)),('fp16 w,z,a; bf16 grads',(H_,H_,H_,B)),('fp16 w,a; fp32 z; bf16 grads',(H_,F32,H_,B)),
                     ('bf16 w; fp16 z,a',(B,H_,H_,B)),('fp16 w,z,a; fp16 grads',(H_,H_,H_,H_)),('fp32 fwd; bf16 grads',(F32,F32,F32,B)),('bf16 w,a; fp32 z',(B,F32,B,B))):
        gr=grads(enc,x,r1,r2,cfg)
        num=sum(float((gr[k]-ref[k]).pow(2).sum()) for k in ref if k not in ZB); den=sum(float(ref[k].pow(2).sum()) for k in ref if k not in ZB)
        worst=max((float((gr[k]-ref[k]).norm()/(ref[k].norm()+1e-30)),k) for k in ref if k not in ZB)
        print('N=%d %dx%d %-34s whole %.3e worst %.3e %s'%(N,hw,hw,name,(num/den)**.5,worst[0],worst[1]))

"""debug: per-parameter gradient errors of a wide RCAN step against the oracle (GPU box)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sr_oracle as O
from tests.test_network_gpu import _pair

kw = dict(scale=2, n_feats=int(sys.argv[1]) if len(sys.argv) > 1 else 128, n_resgroups=1, n_resblocks=2, reduction=16)
kw['n_resgroups'] = 2
h, oh = _pair('rcan', int(sys.argv[2]) if len(sys.argv) > 2 else 1519, **kw)
x, y = O.synthetic_batch(1600, 2, lr_hw=16, scale=2)
loss, out = h.run_train(x=x, y=y)
oloss, oout = oh.run_train(x, y)
print('loss', float(loss), float(oloss))
for (k, p), (k2, q) in zip(h.net.named_parameters(), oh.net.named_parameters()):
    g, r = p.grad.detach().float().cpu().double().reshape(-1), q.grad.double().reshape(-1)
    rel = float((g - r).norm() / (r.norm() + 1e-30))
    cos = float((g @ r) / (g.norm() * r.norm() + 1e-30))
    print('%-45s rel %.3e cos %.6f |r| %.3e' % (k, rel, cos, float(r.norm())))

# ---- channel-attention internals of every RCAB against the oracle (hooks) ----
plan = h.net.engine.plan_for(2, 16, 16, True)
byptr = {t.data_ptr(): t for t in plan.keep}
fused = [a for op, a in plan.fwd if op == 'rumpy_ca_fwd_fused']
bfused = [a for op, a in plan.bwd if op == 'rumpy_ca_bwd_fused']
print(len(fused), 'ca_fwd_fused ops,', len(bfused), 'ca_bwd_fused ops; op names bwd:', sorted(set(op for op, _ in plan.bwd)))
omods = [m for n_, m in oh.net.named_modules() if n_.endswith('body.3')]
rec = []
for m in omods:
    m.avg_pool.register_forward_hook(lambda mod, i, o: rec.append(('mean', o.detach().reshape(o.shape[0], -1))))
    m.conv_du[1].register_forward_hook(lambda mod, i, o: rec.append(('hidden', o.detach().reshape(o.shape[0], -1).clone())))
    m.conv_du[3].register_forward_hook(lambda mod, i, o: rec.append(('gate', o.detach().reshape(o.shape[0], -1))))
oh.net.load_state_dict(O.seeded_state_dict(oh.net, int(sys.argv[2]) if len(sys.argv) > 2 else 1519))
h.net.load_state_dict(oh.net.state_dict())
with torch.no_grad():
    oh.net(x)
loss, out = h.run_train(x=x, y=y)
for i, a in enumerate(fused):
    for j, nm in enumerate(('mean', 'hidden', 'gate')):
        ref = rec[3 * i + j][1]
        got = byptr[getattr(a, nm)].float().cpu()
        print(i, nm, 'max abs err %.3e  max |ref| %.3e' % (float((got - ref).abs().max()), float(ref.abs().max())), 'min |hidden pre|' if nm == 'hidden' else '')

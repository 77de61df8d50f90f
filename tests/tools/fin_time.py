"""Where rumpy_finish_reduce's time goes (GPU box): python tests/tools/fin_time.py [edsr|rcan]
Times the launch with all roles, with the 64-channel convs' items only, and with the tail / head roles only (inputs: the slabs of a real step)."""
import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sr_oracle as O
from rumpy_amd import _lib as L
from rumpy_amd.shared_framework.models import define_model

name = sys.argv[1] if len(sys.argv) > 1 else 'edsr'
kw = dict(scale=4) if name == 'edsr' else dict(scale=4, n_resgroups=10, n_resblocks=20, reduction=16)
x, y = O.synthetic_batch(670, 32, lr_hw=48, scale=4)
h = define_model(name, model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, lr=1e-3, **kw)
net = h.net
for _ in range(2):
    h.run_train(x=x.cuda(), y=y.cuda(), keep_on_device=True)
eng = net.engine
plan = eng.plan_for(32, 48, 48, True)
stream = torch.cuda.current_stream().cuda_stream
gs = 1.0 / plan.out.numel()


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


n = len(plan.reduce_keep)
print('%s: %d items, jobs per item %s' % (name, n, sorted(set(plan.reduce_host[i].njobs for i in plan.reduce_keep))))
print('all roles       %.1f us' % timed(lambda: eng._reduce(plan, stream, plan.reduce_dev_notail, n, gs, tail=True, head=True)))
print('items only      %.1f us' % timed(lambda: eng._reduce(plan, stream, plan.reduce_dev_notail, n, gs, tail=False, head=False)))
print('tail only       %.1f us' % timed(lambda: eng._reduce(plan, stream, plan.reduce_dev_notail, 0, gs, tail=True, head=False)))
print('head only       %.1f us' % timed(lambda: eng._reduce(plan, stream, plan.reduce_dev_notail, 0, gs, tail=False, head=True)))
opt = h.optimizer
print('adam + re-pack  %.1f us' % timed(lambda: opt.step()))

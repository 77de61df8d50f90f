"""The weight-gradient launch of the 64-channel layers (wgrad_dma_kernel<4>, the shares of a real training plan) timed alone, back to back:
   python tests/tools/wgrad_time.py [edsr|rcan] [reps]        RUMPY_AMD_LIB selects a variant build (WGRAD_ABL timing builds: results wrong by design)"""
import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sr_oracle as O
from rumpy_amd.shared_framework.models import define_model

name = sys.argv[1] if len(sys.argv) > 1 else 'edsr'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
kw = dict(scale=4) if name == 'edsr' else dict(scale=4, n_resgroups=10, n_resblocks=20, reduction=16)
nb = 32 if name == 'edsr' else 16
x, y = O.synthetic_batch(670, nb, lr_hw=48, scale=4)
h = define_model(name, model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, lr=1e-3, **kw)
for _ in range(2):
    h.run_train(x=x.cuda(), y=y.cuda(), keep_on_device=True)
eng = h.net.engine
plan = eng.plan_for(nb, 48, 48, True)
stream = torch.cuda.current_stream().cuda_stream
ts = []
for rnd in range(3):
    for _ in range(3):
        eng._wgrad4(plan, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng._wgrad4(plan, stream)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / reps * 1e3)
print('%s wgrad4 alone: %s us per launch (lib %s)' % (name, ' '.join('%.1f' % t for t in ts), os.environ.get('RUMPY_AMD_LIB', 'product')))

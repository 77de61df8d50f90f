"""Loss trajectory of the HIP path against the CPU oracle over many steps (GPU box): python tests/tools/trajectory.py [edsr|rcan] [steps]"""
import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sr_oracle as O
from rumpy_amd.shared_framework.models import define_model
name = sys.argv[1] if len(sys.argv) > 1 else 'edsr'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
kw = {'edsr': dict(scale=2, num_blocks=4, res_scale=0.1), 'rcan': dict(scale=2, n_resgroups=2, n_resblocks=3, reduction=16)}[name]
torch.manual_seed(8)
h = define_model(name, model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, lr=2e-4, **kw)
onet = O.build_oracle(name, **kw)
onet.load_state_dict({k: v.cpu() for k, v in h.net.state_dict().items()})
oh = O.OracleHandler(onet, lr=2e-4)
# a learnable target: HR = smooth image, LR = its 2x average pooling (the nets must learn to upsample)
gen = torch.Generator().manual_seed(3)
base = torch.nn.functional.interpolate(torch.rand(8, 3, 12, 12, generator=gen), size=(48, 48), mode='bicubic', align_corners=False).clamp(0, 1)
lr_img = torch.nn.functional.avg_pool2d(base, 2)
worst = 0.0
for s in range(steps):
    idx = torch.randperm(8, generator=gen)[:4]
    x, y = lr_img[idx].contiguous(), base[idx].contiguous()
    l, _ = h.run_train(x=x, y=y)
    ol, _ = oh.run_train(x, y)
    worst = max(worst, abs(float(l) - float(ol)) / float(ol))
    if s % 10 == 0 or s == steps - 1:
        print('step %3d  hip %.5f  oracle %.5f' % (s, float(l), float(ol)))
print('largest relative loss difference over %d steps: %.3f' % (steps, worst))

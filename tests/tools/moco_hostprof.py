"""cProfile of the MoCo step's host side (GPU box): python tests/tools/moco_hostprof.py"""
import cProfile
import os
import pstats
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import contrastive_oracle as CO  # noqa: E402  (inputs only)
from rumpy_amd.shared_framework.models import define_model  # noqa: E402

h = define_model('mococontrastive', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, model_name='default', crop_count=2, lr=1e-4)
xs = [CO.contrastive_batch(10 + i, 32, 2, hw=48).view(32, 6, 48, 48).cuda() for i in range(4)]
for i in range(10):
    h.run_train(x=xs[i % 4], y=None)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    h.run_train(x=xs[i % 4], y=None)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)

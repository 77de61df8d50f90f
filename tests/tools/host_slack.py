"""How close the host is to being the bottleneck of the EDSR-baseline step: time the Python loop alone (no final sync) against the synced loop.
    python tests/tools/host_slack.py [steps]"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import SCHED, synthetic_batch  # noqa: E402
from rumpy_amd.shared_framework.models import define_model  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(8)
h = define_model('edsr', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False, scale=4, lr=1e-4,
                 scheduler='cosine_annealing_warm_restarts', scheduler_params=SCHED)
pool = [tuple(t.cuda() for t in synthetic_batch(1234 + i, 32)) for i in range(8)]
for i in range(50):
    h.run_train(x=pool[i % 8][0], y=pool[i % 8][1], keep_on_device=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    h.run_train(x=pool[i % 8][0], y=pool[i % 8][1], keep_on_device=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('per step: host loop %.3f ms, with the final sync %.3f ms (the loop itself waits for each step\'s forward pass through the loss read-back)'
      % ((t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3))
# host work alone: the same loop with the GPU made irrelevant is not available; instead: cProfile-free estimate from a loop over an idle GPU
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    h.run_train(x=pool[i % 8][0], y=pool[i % 8][1], keep_on_device=True)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(12)

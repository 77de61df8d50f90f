"""Per-step GPU and host times of the first steps of a fresh process (the driver times steps 6..25): where does a short run lose 7 %?
usage: python3 tests/tools/step_ramp.py [steps] [spin_ms_before]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from rumpy_amd.shared_framework.models import define_model

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dev = torch.device('cuda', 0)
torch.manual_seed(8)
h = define_model('edsr', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False, scale=4,
                 lr=1e-4, scheduler='cosine_annealing_warm_restarts', scheduler_params=bench.SCHED)
pool = []
for i in range(8):
    x, y = bench.synthetic_batch(1234 + i, 32, lr_hw=48)
    pool.append((x.to(dev), y.to(dev)))
spin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
if spin_ms > 0:          # keep the GPU busy with unrelated work first: separates clock ramp from first-use effects of the step itself
    kind = sys.argv[3] if len(sys.argv) > 3 else 'mm'
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    big, big2 = torch.empty(1 << 29, device=dev, dtype=torch.float32), torch.empty(1 << 29, device=dev, dtype=torch.float32)      # 2 GB each
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < spin_ms * 1e-3:
        for _ in range(4):
            if kind in ('mm', 'both'):
                for _ in range(5):
                    a @ a
            if kind in ('copy', 'both'):
                big2.copy_(big)
        torch.cuda.synchronize()
    del big, big2
torch.cuda.synchronize()
import copy
sd0 = copy.deepcopy(h.net.state_dict())


def run(tag):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    host = []
    wall0 = time.perf_counter()
    for i in range(steps):
        t0 = time.perf_counter()
        ev[i][0].record()
        h.run_train(x=pool[i % 8][0], y=pool[i % 8][1], keep_on_device=True)
        ev[i][1].record()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - wall0
    gpu = [a.elapsed_time(b) for a, b in ev]
    gaps = [ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(steps - 1)]
    print(tag, 'wall %.1f ms for %d steps' % (1e3 * wall, steps))
    for lo, hi in ((0, 1), (1, 2), (2, 5), (5, 10), (10, 25), (25, 50), (50, 100), (100, steps)):
        if lo >= steps:
            break
        hi = min(hi, steps)
        n = hi - lo
        print('steps %3d..%3d: gpu %.4f ms  gap to next %.4f ms  host call %.4f ms' % (lo, hi - 1, sum(gpu[lo:hi]) / n, sum(gaps[lo:min(hi, steps - 1)]) / max(1, min(hi, steps - 1) - lo),
                                                                                      1e3 * sum(host[lo:hi]) / n))


run('fresh process:')
run('same weights continue:')
h.net.load_state_dict(sd0)            # back to the initial weights (Adam moments kept): is the ramp a property of the VALUES?
torch.cuda.synchronize()
run('initial weights reloaded:')

#!/usr/bin/env python3
"""Headline benchmark: 48x48 LR patches/sec for a full EDSR-baseline x4 training step (forward + L1 + backward +
Adam + per-batch LR schedule) on N MI355X, bf16 MFMA operands / fp32 accumulate, through the handler API.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): EDSR-baseline (64 feats x 16 blocks) x4, per-GPU batch 32 of
uniform[0,1) synthetic [32,3,48,48] / [32,3,192,192] pairs (numpy default_rng(1234+i), pool of 8 device-resident
batches), weights = handler default init under torch.manual_seed(8), Adam lr 1e-4, cosine warm restarts (T_0 40000).
One JSON line on rank 0; `roofline` prices the dominant kernel (3x3 conv 64->64, forward and data-gradient launches)
from HIP-event timings taken in this process; `cpu_baseline` times the CPU oracle on the host cores.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PATCH_TRAIN = 27.407e9      # EDSR-baseline x4 @48x48, fwd + wgrad + dgrad
HBM_PEAK_GBPS = 8000.0               # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PMC_TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
MFMA_BF16_PEAK_TFLOPS = 2500.0       # MI355X dense bf16 (MI355X_MICROARCH.md)
MFMA_FP8_PEAK_TFLOPS = 5000.0        # MI355X dense fp8 (--precision fp8: the block kernels' sweeps run on v_mfma_scale_f32_16x16x128_f8f6f4)
SCHED = {'t_mult': 1, 'restart_period': 40000, 'lr_min': 1e-7}


def synthetic_batch(seed, n, lr_hw=48, scale=4, channels=3):
    """"DIV2K-shaped" synthetic batch - x uniform[0,1) fp32 [n,3,48,48], y uniform[0,1) fp32 [n,3,192,192] from
    numpy.random.default_rng(seed) (the same stream the tests draw their inputs from)."""
    import numpy as np
    import torch
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.random((n, channels, lr_hw, lr_hw), dtype=np.float32))
    y = torch.from_numpy(rng.random((n, channels, lr_hw * scale, lr_hw * scale), dtype=np.float32))
    return x, y


def bench_moco(args):
    """--model moco (information): one MoCo training step of the degradation encoder (define_model('mococontrastive'),
    crop_count 2: N query + N key crops of 48x48) per step on one GPU.  Same line format; `roofline` prices the trunk's 64 -> 64 3x3 conv
    (the largest launch of the step) from HIP-event timings of eager launches taken in this process."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback exists for the product path)')
    if args.gpus != 1 or int(os.environ.get('WORLD_SIZE', '1')) != 1:
        raise SystemExit('--model moco is a one-GPU line')
    from rumpy_amd import _lib as L
    from rumpy_amd.shared_framework.models import define_model
    dev = torch.device('cuda', 0)
    N = args.batch
    torch.manual_seed(8)
    h = define_model('mococontrastive', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, model_name='default', crop_count=2, lr=1e-4)
    rng = np.random.default_rng(1234)
    pool = [torch.from_numpy(rng.random((N, 6, 48, 48), dtype=np.float32)).to(dev) for _ in range(8)]
    for i in range(args.warmup):
        h.run_train(x=pool[i % 8], y=None)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, _ = h.run_train(x=pool[i % 8], y=None)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    # dominant launch: conv 64 -> 64 on [N, 48, 48], with the plan's own buffers and filter image
    enc = h.net.encoder_q
    plan = enc._train_plans[(N, 48, 48, 0)]
    im0 = enc._training_images(dev)[0]
    wf, bp = im0[0], im0[2]
    from rumpy_amd.regression.models.contrastive_learning.encoding_models import _conv_plain, TRAIN_FMT, TRAIN_Z32
    from rumpy_amd import _lib as L
    s = torch.cuda.current_stream(dev).cuda_stream
    if TRAIN_Z32:      # as the step launches it since round 4: filter + rounding-residual image, fp32 conv output (twice the MFMAs per algorithmic FLOP)
        launch = lambda: L.call('rumpy_enc_conv', L.EncConvArgs(x=plan['a'][0].data_ptr(), w=wf.data_ptr(), bias=bp.data_ptr(), out=plan['z'][1].data_ptr(), N=N, H=48, W=48,
                                                                cin=64, cout=64, stride=1, neg_slope=1.0, fmt=TRAIN_FMT, w_lo=im0[3].data_ptr(), out_fmt=L.FMT_F32), s)
    else:
        launch = lambda: _conv_plain(plan['a'][0].data_ptr(), wf.data_ptr(), bp.data_ptr(), plan['z'][1].data_ptr(), N, 48, 48, 64, 64, s, fmt=TRAIN_FMT)
    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        launch()
    e1.record()
    torch.cuda.synchronize(dev)
    avg_s = e0.elapsed_time(e1) * 1e-3 / 50
    flop = 2.0 * N * 48 * 48 * 64 * 64 * 9
    alg_bytes = N * 48 * 48 * 64 * (2.0 + (4.0 if TRAIN_Z32 else 2.0))
    tflops, gbps = flop / avg_s / 1e12, alg_bytes / avg_s / 1e9
    roofline = {'bound': 'mfma', 'achieved': round(tflops, 2), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                'traffic': None, 'kernel': ('enc_conv_kernel via rumpy_enc_conv (3x3 conv 64 -> 64 of the encoder trunk on the filter + its rounding-residual image, fp32 output; the largest launch of the step)'
                                            if TRAIN_Z32 else 'conv3x3_strip_kernel via rumpy_conv3x3 (3x3 conv 64 -> 64 of the encoder trunk, the largest launch of the step)'),
                'avg_launch_us': round(avg_s * 1e6, 3), 'launches_timed': 50, 'algorithmic_gflop_per_launch': round(flop / 1e9, 3),
                'algorithmic_mb_per_launch': round(alg_bytes / 1e6, 3),
                'hbm': {'achieved': round(gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(gbps / HBM_PEAK_GBPS, 4)}}
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import contrastive_oracle as CO            # the checker, timed as the CPU baseline: the ONLY use of oracle/ here
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        try:                                             # container CPU quota (cgroup v2)
            q, per = open('/sys/fs/cgroup/cpu.max').read().split()
            if q != 'max':
                usable = max(1, min(usable, int(int(q) / int(per))))
        except Exception:
            pass
        cores = max(1, min(usable, args.cpu_threads))
        torch.set_num_threads(cores)
        oh = CO.OracleContrastiveHandler('mococontrastive', crop_count=2, lr=1e-4)
        xc = pool[0][:args.cpu_batch].cpu()
        oh.run_train(xc)
        best, total, timed = 1e30, 0.0, 0
        while timed < 3 or (total < 10.0 and timed < 200):
            t2 = time.perf_counter()
            oh.run_train(xc)
            dt = time.perf_counter() - t2
            best, total, timed = min(best, dt), total + dt, timed + 1
        cpu = {'value': round(2 * args.cpu_batch / best, 1), 'unit': 'LR crops/s', 'cores': torch.get_num_threads(), 'kind': 'port',
               'sample': '1 warm-up + %d timed MoCo steps of %d + %d 48x48 crops (%.1f s of CPU work; value = best step), torch CPU fp32 oracle'
                         % (timed, args.cpu_batch, args.cpu_batch, total)}
    # 2.752 GFLOP per (query, key) pair and step: encoder forward 0.688 x (3 for the query: forward + data + weight gradient, 1 for the key)
    line = {'metric': '48px LR crops/sec (MoCo train step) DASR encoder bf16', 'value': round(2 * N * args.steps / elapsed, 1), 'unit': 'LR crops/s',
            'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic uniform[0,1) crops, random-init weights (seed 8)',
            'config': {'workload': 'MoCo step of the degradation encoder (query + key encoder, 8192-entry queue), %d + %d crops of 48x48 per step' % (N, N),
                       'global_batch': N, 'parallelism': 'dp1', 'optimizer': 'Adam lr 1e-4', 'loss': float(loss),
                       'train_tflops': round(N * args.steps / elapsed * 2.752e9 / 1e12, 2)},
            'roofline': roofline, 'cpu_baseline': cpu}
    print(json.dumps(line), flush=True)


def source_sha16(rel_paths):
    """sha256[:16] over the given source files (relative to the repo root): what a committed PMC measurement is keyed on"""
    import hashlib
    h = hashlib.sha256()
    for rel in rel_paths:
        with open(os.path.join(ROOT, rel), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (tests/tools/pmc_step.sh writes the
    entries: separate rocprofv3 --pmc passes over this very command, 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes).
    An entry counts only while the kernel sources it was measured on are unchanged (sha256 of the files): otherwise (None, why) - the
    line then says `traffic: null` instead of quoting a number of other code."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            e = json.load(f).get(key)
    except Exception as ex:      # noqa: BLE001
        return None, 'no profiles/pmc_traffic.json (%s)' % type(ex).__name__
    if e is None:
        return None, 'no PMC entry for %s' % key
    now = source_sha16(e['sources'])
    if now != e['sha16']:
        return None, 'PMC entry for %s was measured on other kernel sources (sha16 %s, now %s): re-run tests/tools/pmc_step.sh' % (key, e['sha16'], now)
    return float(e['bytes_per_launch']), e['source']


def _usable_cores(cap):
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    try:                                             # container CPU quota (cgroup v2): "<quota> <period>" or "max <period>"
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            usable = max(1, min(usable, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(usable, cap)), usable


def bench_eval(args):
    """--mode eval (information; north_star: "training/inference path"): the evaluation forward pass of a whole image as eval_sisr makes
    it (run_eval of ONE fp32 NCHW image, rumpy/SISR/models/interface.py:103-124; EDSR / RCAN are not chopped), default a 510 x 339 LR image
    (a DIV2K x4 validation image).  Plans are IEEE fp16 with the upsampler filters as image + rounding-residual image (two
    conv launches per stage); the line prices that against the one-launch form (RUMPY_EVAL_UP_RESIDUAL=0) in the same process.
    value = LR megapixels per second with the image resident in HBM and the output left there (keep_on_device)."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback exists for the product path)')
    if args.gpus != 1 or int(os.environ.get('WORLD_SIZE', '1')) != 1:
        raise SystemExit('--mode eval is a one-GPU line')
    import ctypes
    from rumpy_amd import _lib as L
    from rumpy_amd.shared_framework.models import define_model
    dev = torch.device('cuda', 0)
    H, W = args.eval_size
    name = {'edsr256': 'edsr'}.get(args.model, args.model)
    extra = {'edsr256': dict(num_features=256, num_blocks=32, res_scale=0.1)}.get(args.model, {})
    if args.model not in ('edsr', 'rcan', 'edsr256'):
        raise SystemExit('--mode eval takes --model edsr | rcan | edsr256')
    fwd_flop_per_px = {'edsr': 9.138e9, 'rcan': 73.35e9, 'edsr256': 231.6e9}[args.model] / 2304.0      # forward FLOPs per 48 x 48 patch
    rng = np.random.default_rng(4321)
    pool = [torch.from_numpy(rng.random((1, 3, H, W), dtype=np.float32)).to(dev) for _ in range(4)]

    def build(up_residual):
        os.environ['RUMPY_EVAL_UP_RESIDUAL'] = '1' if up_residual else '0'
        torch.manual_seed(8)
        h = define_model(name, model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=True, checkpoint_load=False, loss_masking=False,
                         scale=4, **extra)
        h.defer_eval_status = True        # this loop keeps every image on the device and examines the status words itself (check_eval below)
        return h

    def run(h, steps, warmup):
        for i in range(warmup):
            h.run_eval(x=pool[i % 4], keep_on_device=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            out, _, _ = h.run_eval(x=pool[i % 4], keep_on_device=True)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps, out
    steps, warmup = min(args.steps, 100), min(args.warmup, 10)
    h = build(True)
    sec, out = run(h, steps, warmup)
    eng = h.net.engine
    eng.check_eval()
    plan = eng.plan_for(1, H, W, False, eng.eval_fmt)
    ops = [op for op, _ in plan.fwd]
    blocks = [a for op, a in plan.fwd if op in ('rumpy_conv_block', 'rumpy_rcab_fwd', 'rumpy_rcab2_fwd')]
    roofline = None
    if blocks:
        lib = L.lib()
        lib.rumpy_probe_begin(5, len(blocks) * 5 + 8)
        for i in range(5):
            h.run_eval(x=pool[i % 4], keep_on_device=True)
        torch.cuda.synchronize(dev)
        eng.check_eval()
        tot = ctypes.c_double(0.0)
        n_launch = lib.rumpy_probe_end(ctypes.byref(tot))
        if n_launch > 0:
            avg_s = tot.value * 1e-3 / n_launch
            flop = 2 * 2.0 * H * W * 64 * 576                    # two 64 -> 64 3x3 convs per launch
            alg_bytes = 2.0 * H * W * 64 * 2                      # evaluation: block input and output only, fp16
            tflops, gbps = flop / avg_s / 1e12, alg_bytes / avg_s / 1e9
            roofline = {'bound': 'mfma', 'achieved': round(tflops, 2), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(tflops / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None,
                        'kernel': ('rcab2_kernel' if 'rumpy_rcab2_fwd' in ops else 'rcab_kernel' if 'rumpy_rcab_fwd' in ops else 'conv_block_kernel') + ' forward form, fp16, column tiles (one residual block per launch)',
                        'avg_launch_us': round(avg_s * 1e6, 3), 'launches_timed': n_launch, 'launches_per_image': len(blocks),
                        'algorithmic_gflop_per_launch': round(flop / 1e9, 3), 'algorithmic_mb_per_launch': round(alg_bytes / 1e6, 3),
                        'hbm': {'achieved': round(gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(gbps / HBM_PEAK_GBPS, 4)}}
    h1 = build(False)
    sec1, _ = run(h1, steps, warmup)
    h1.net.engine.check_eval()
    os.environ.pop('RUMPY_EVAL_UP_RESIDUAL', None)
    os.environ['RUMPY_BLOCK_W48'] = '1'              # A/B: two launches per block on images wider than one strip (the round-2 path)
    h2 = build(True)
    sec2, _ = run(h2, steps, warmup)
    h2.net.engine.check_eval()
    os.environ.pop('RUMPY_BLOCK_W48', None)
    del h1, h2
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import sr_oracle as O                  # the checker, timed as the CPU baseline: the ONLY use of oracle/ in this function
        cores, usable = _usable_cores(args.cpu_threads)
        torch.set_num_threads(cores)
        torch.manual_seed(8)
        onet = O.build_oracle(name, scale=4, **extra)
        oh = O.OracleHandler(onet, lr=1e-4, eval_mode=True)
        ch, cw = max(16, H // 3), max(16, W // 3)             # bounded sample: a ninth of the image, same per-pixel work
        xc = pool[0][:, :, :ch, :cw].cpu()
        oh.run_eval(xc)
        best, total, timed = 1e30, 0.0, 0
        while timed < 3 or (total < 10.0 and timed < 40):
            t2 = time.perf_counter()
            oh.run_eval(xc)
            dt = time.perf_counter() - t2
            best, total, timed = min(best, dt), total + dt, timed + 1
        cpu = {'value': round(ch * cw / best / 1e6, 4), 'unit': 'LR Mpix/s', 'cores': torch.get_num_threads(), 'kind': 'port',
               'sample': '1 warm-up + %d timed forward passes of a %d x %d LR crop (%.1f s of CPU work; value = best pass), torch CPU fp32 oracle; '
                         'host reports %d logical CPUs, %d usable under the cgroup quota' % (timed, cw, ch, total, os.cpu_count() or 0, usable)}
    line = {'metric': 'LR megapixels/sec (whole-image evaluation forward) %s x4 fp16' % args.model.upper(), 'value': round(H * W / sec / 1e6, 3),
            'unit': 'LR Mpix/s', 'n_gpus': 1, 'steps': steps, 'warmup': warmup, 'ms_per_step': round(1e3 * sec, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic uniform[0,1) image, random-init weights (seed 8)',
            'config': {'workload': '%s x4 evaluation forward of one %d x %d LR image (%d x %d output), fp16 evaluation plans, image resident in HBM'
                                   % ({'edsr': 'EDSR-baseline', 'rcan': 'RCAN 10 x 20', 'edsr256': 'EDSR 256 x 32'}[args.model], W, H, 4 * W, 4 * H),
                       'fwd_tflops': round(H * W * fwd_flop_per_px / sec / 1e12, 2),
                       'fwd_mfma_frac': round(H * W * fwd_flop_per_px / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                       'launches_per_image': len(ops) + 1, 'one_launch_blocks': len(blocks),
                       'ms_upsampler_one_launch_form': round(1e3 * sec1, 4),           # RUMPY_EVAL_UP_RESIDUAL=0: what the doubled upsampler launches cost
                       'ms_two_launches_per_block': round(1e3 * sec2, 4)},             # RUMPY_BLOCK_W48=1: the round-2 path for W > 48
            'roofline': roofline, 'cpu_baseline': cpu}
    print(json.dumps(line), flush=True)


def count_gpus_without_hip(topology=None, environ=None):
    """GPUs this process would see, counted WITHOUT loading anything that initialises HIP / HSA (the launcher parent must stay a process that
    has never touched the GPU): the KFD topology's nodes with simd_count > 0 (CPU agents have 0), cut down by the ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES
    lists the runtime honours (each list: entries up to the first invalid one; an entry is an index into what the previous filter left, or a
    GPU-<uuid>).  None when the topology is not readable (the ranks then find out themselves and fail with their own message)."""
    topology = topology or os.environ.get('RUMPY_KFD_TOPOLOGY', '/sys/class/kfd/kfd/topology/nodes')
    environ = os.environ if environ is None else environ
    try:
        nodes = sorted(os.listdir(topology), key=lambda d: (len(d), d))
    except OSError:
        return None
    have = 0
    for node in nodes:
        try:
            with open(os.path.join(topology, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue          # (a node this cgroup may not read is a device it may not use either)
        if int(props.get('simd_count', '0')) > 0:
            have += 1
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        if var not in environ:
            continue
        kept = 0
        for entry in environ[var].split(','):
            entry = entry.strip()
            if entry.startswith('GPU-') or (entry.isdigit() and int(entry) < have):
                kept += 1
            else:
                break
        have = kept
    return have


def launch_ranks(n, argv):
    """`python bench.py --gpus N` called plainly (no RANK in the environment): this process - which imports nothing that touches the GPU -
    starts the N rank processes itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would set
    them), relays rank 0's JSON line and exits with the worst return code.  Children are fresh interpreters (subprocess, never exec).
    Supervision (what torchrun's agent does): every child is polled; when one exits non-zero the others - which would wait for it in a
    collective for ever - are killed (exactly the PIDs started here) and that code is returned; the same after RUMPY_BENCH_TIMEOUT seconds
    (default 3600) with code 124."""
    import socket
    import subprocess
    import threading
    one_device = os.environ.get('RUMPY_BENCH_ONE_DEVICE') == '1'
    if not one_device:
        have = count_gpus_without_hip()
        if have is not None and have < n:
            sys.stderr.write('bench.py: --gpus %d, but this node exposes %d GPU(s)\n' % (n, have))
            return 2
    with socket.socket() as sk:              # a free rendezvous port
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(os.environ.get('MASTER_PORT', port)), HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.setdefault('OMP_NUM_THREADS', '8')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get('RUMPY_BENCH_TIMEOUT', '3600'))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            rc = failed[0]
            sys.stderr.write('bench.py: rank %d exited with code %d; ending the other ranks\n' % (codes.index(failed[0]), rc))
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            rc = 124
            sys.stderr.write('bench.py: ranks still running after RUMPY_BENCH_TIMEOUT; ending them\n')
            break
        time.sleep(0.1)
    for p in procs:
        if p.poll() is None:                 # a peer of a failed rank waits in a collective: end exactly that process
            p.kill()
        p.wait()
    reader.join(timeout=5)
    sys.stdout.write((out0[0] if out0 else '') or '')
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--batch', type=int, default=32, help='LR patches per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--probe-steps', type=int, default=5)
    ap.add_argument('--no-as-called', action='store_true', help='skip the informational host-tensor leg (profiles: its copies run beside the kernels of 23 more steps)')
    ap.add_argument('--settle-ms', type=float, default=0.0,
                    help='A/B tooling only (default 0 = off): hold the GPU under load for this long BEFORE the W warm-up steps.  `value` is always the '
                         'contract\'s region of the process as the flags describe it; with the default flags that is a fresh process (ADVICE r4).')
    ap.add_argument('--settled-probe-ms', type=float, default=120.0,
                    help='AFTER the contract\'s timed region (one GPU): keep the GPU under load for this long, then time the same W + K steps again and '
                         'report them as the extra field `settled` - a fresh process starts with the core clock at its idle level and the SMU needs '
                         '30-50 ms of load to raise it (profiles/r04_clock_ramp.txt: MFMA-bound kernels 15-27 %% slower over the first 25 steps).  '
                         'Information beside `value`, never `value`.  0 = off')
    ap.add_argument('--cpu-threads', type=int, default=32)
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--model', choices=('edsr', 'rcan', 'qrcan', 'blindqrcan', 'edsr256', 'moco'), default='edsr',
                    help='edsr = the headline workload (BASELINE.json metric); rcan = RCAN x4 10x20 (BASELINE config 3), qrcan = the same with a meta-attention q-layer (5 metadata entries) in every block, blindqrcan = frozen contrastive degradation encoder + QRCAN (BASELINE config 5 in bf16; q-layers as in the reference test config), edsr256 = EDSR at the reference\'s shipped width (div2k/edsr.toml: 256 features x 32 blocks); all for information')
    ap.add_argument('--device-patches', action='store_true',
                    help='draw every batch on the fly from a device-resident uint8 image cache (SURVEY.md 8f.1) instead of the pre-generated pool')
    ap.add_argument('--lr-size', type=int, default=48, help='LR patch side of the training step (48 = the headline metric; the reference\'s shipped '
                                                            'div2k configs crop 64: --lr-size 64 --batch 8 is div2k/rcan.toml)')
    ap.add_argument('--mode', choices=('train', 'eval'), default='train', help='eval = whole-image evaluation forward (information)')
    ap.add_argument('--precision', choices=('bf16', 'fp8'), default='bf16',
                    help="training arithmetic of the one-launch residual-block / RCAB kernels: bf16 (default, the headline metric) or the fp8 opt-in "
                         "(block-scaled fp8 MFMA; BASELINE config 5; its own accuracy class, DESIGN.md 2.2) - for information, never the headline")
    ap.add_argument('--eval-size', type=int, nargs=2, default=(339, 510), metavar=('H', 'W'), help='LR image of --mode eval')
    ap.add_argument('--allreduce-form', choices=('auto', 'inline', 'early', 'side'), default='auto',
                    help='data-parallel runs: how the gradient all-reduce is issued (auto = early half for >= 4 M gradient elements, else one '
                         'inline collective after the backward pass); reported as distributed.allreduce_form')
    args = ap.parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.allreduce_form != 'auto':
        for k in ('RUMPY_DP_EARLY', 'RUMPY_DP_LATE', 'RUMPY_DP_INLINE'):
            os.environ.pop(k, None)
        os.environ.update({'inline': {'RUMPY_DP_LATE': '1', 'RUMPY_DP_INLINE': '1'}, 'early': {'RUMPY_DP_EARLY': '1'},
                           'side': {'RUMPY_DP_LATE': '1', 'RUMPY_DP_INLINE': '0'}}[args.allreduce_form])
    if args.model == 'moco':
        return bench_moco(args)
    if args.mode == 'eval':
        return bench_eval(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # test hook (1-GPU boxes): RUMPY_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo, so that the N > 1 code path (broadcast,
    # two-phase weight gradients, early all-reduce on the side stream, max-over-ranks timing) can run where RCCL cannot
    one_device = os.environ.get('RUMPY_BENCH_ONE_DEVICE') == '1'
    if one_device:
        local_rank = 0
    # second test hook: RUMPY_DP_FORCE=1 under torch.distributed.run with ONE rank keeps the whole data-parallel path on (RCCL communicator,
    # broadcast, early all-reduce on the side stream) - the only way to run RCCL itself on a 1-GPU box
    dp = world > 1 or (os.environ.get('RUMPY_DP_FORCE') == '1' and 'RANK' in os.environ)
    if dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        if one_device:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    if dp and world != args.gpus and os.environ.get('RUMPY_DP_FORCE') != '1':
        raise SystemExit('--gpus %d, but WORLD_SIZE is %d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback exists for the product path)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from rumpy_amd import _lib as L
    from rumpy_amd.parallel import broadcast_parameters
    from rumpy_amd.shared_framework.models import define_model

    N = args.batch
    P = args.lr_size
    flop_per_patch = {'edsr': FLOP_PER_PATCH_TRAIN, 'edsr256': 694.7e9}.get(args.model, 220.04e9) * (P * P / 2304.0)      # SURVEY.md 8(d), per 48 x 48 patch
    torch.manual_seed(8)                                    # reference default seed (net_train.py:20)
    BLIND = dict(style='standard', include_q_layer=True, selective_meta_blocks=[True] + [False] * 9, num_q_layers_inner_residual=1)
    extra = {'qrcan': dict(style='standard', include_q_layer=True, metadata=['m%d' % i for i in range(5)]),
             'blindqrcan': dict(block_encoder_loading=True, **BLIND), 'edsr256': dict(num_features=256, num_blocks=32, res_scale=0.1)}.get(args.model, {})      # encoder weights: random init (no checkpoint offline)
    fp8 = args.precision == 'fp8'
    if fp8:
        extra = dict(extra, precision='fp8')
    mfma_peak = MFMA_FP8_PEAK_TFLOPS if fp8 else MFMA_BF16_PEAK_TFLOPS      # the dominant kernel's matrix instruction (re-derived below from what the plan really launches)
    fp8_live = fp8
    h = define_model({'blindqrcan': 'contrastiveblindqrcan', 'edsr256': 'edsr'}.get(args.model, args.model), model_save_dir=tempfile.mkdtemp(), device=local_rank,
                     eval_mode=False, checkpoint_load=False, loss_masking=False, scale=4, lr=1e-4, scheduler='cosine_annealing_warm_restarts',
                     scheduler_params=SCHED, **extra)
    meta_pool = [torch.rand(N, 5, 1, 1, generator=torch.Generator().manual_seed(77 + i)).to(dev) for i in range(8)] if args.model == 'qrcan' else None
    if dp:
        broadcast_parameters(h.net)
        h.set_multi_gpu()
    pool = []
    for i in range(8):
        x, y = synthetic_batch(1234 + i + 100 * rank, N, lr_hw=P)
        pool.append((x.to(dev), y.to(dev)))

    src = None
    if args.device_patches:
        # random crop + flips + transpose + uint8 -> float/255 on the GPU, random numbers drawn like the reference
        import random
        from rumpy_amd.sr_tools.device_patches import DevicePatchSource
        gen = np.random.default_rng(99 + rank)
        lrs = [gen.integers(0, 256, (192, 256, 3), dtype=np.uint8) for _ in range(64)]
        hrs = [gen.integers(0, 256, (768, 1024, 3), dtype=np.uint8) for _ in range(64)]
        src = DevicePatchSource(lrs, hrs, 4, P, device=dev)
        random.seed(8 + rank)

    def step(i):
        if src is not None:
            x, y = src.sample([(i * N + k) % len(src) for k in range(N)], rng=random)
        else:
            x, y = pool[i % len(pool)]
        if meta_pool is not None:
            return h.run_train(x=x, y=y, keep_on_device=True, extra_channels=meta_pool[i % len(meta_pool)])
        return h.run_train(x=x, y=y, keep_on_device=True)

    def fence():
        torch.cuda.synchronize(dev)
        if dp:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_region():
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize fences; MAX over the ranks"""
        for i in range(args.warmup):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss, _ = step(i)
        fence()
        own = time.perf_counter() - t0
        worst = own
        if dp:
            t = torch.tensor([own], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst = float(t.item())
        return worst, own, loss

    # The driver runs ONE plain command per N.  When no all-reduce form is forced, a data-parallel run first tries both forms as part of its
    # WARM-UP (a short region each, inside one process group: one inline collective behind the backward pass / the early half on the side stream
    # under the remaining weight gradients), keeps the faster one, and then runs the contract's region - W warm-up steps, exactly K
    # timed steps - ONCE, in that form.  `value` is that region; the trial figures are reported under `distributed.forms` (`value`
    # is never the better of two timed regions).  Every rank prints its own time to stderr.
    forms_ms = forms_loss = None
    settle = settled = None
    if args.settle_ms > 0:
        t_load = time.perf_counter()
        n_settle = 0
        if not dp:
            while time.perf_counter() - t_load < 1e-3 * args.settle_ms:
                step(n_settle)
                n_settle += 1
        else:
            for n_settle in range(1, 1 + max(1, int(args.settle_ms / 1.2))):      # the same count on every rank: the step contains the all-reduce
                step(n_settle)
        settle = {'steps_before_warmup': n_settle, 'ms': round(1e3 * (time.perf_counter() - t_load), 1),
                  'what': '--settle-ms (A/B tooling): training steps that kept the GPU loaded in front of the W warm-up steps'}
    hipnet0 = getattr(h.net, 'hip_generator', h.net)
    if dp and args.allreduce_form == 'auto' and hasattr(hipnet0, 'engine_forward') and not getattr(hipnet0, 'use_graph', False) \
            and not any(os.environ.get(k) for k in ('RUMPY_DP_EARLY', 'RUMPY_DP_LATE')):
        forms_ms, forms_loss, best = {}, {}, None
        k_try, w_try = min(args.steps, 20), min(args.warmup, 5)
        keep = (args.steps, args.warmup)
        args.steps, args.warmup = k_try, w_try
        for form in ('inline', 'early'):
            h.set_allreduce_form(form)
            e, own, l = timed_region()
            forms_ms[h.data_parallel.form] = round(1e3 * e / k_try, 4)
            forms_loss[h.data_parallel.form] = float(l)
            sys.stderr.write('bench.py: rank %d, warm-up trial of all-reduce form %s: %.4f ms per step over %d steps (max over ranks %.4f)\n'
                             % (rank, h.data_parallel.form, 1e3 * own / k_try, k_try, 1e3 * e / k_try))
            if best is None or e < best[0]:
                best = (e, form)
        args.steps, args.warmup = keep
        h.set_allreduce_form(best[1])       # (max over ranks, all-reduced: every rank picks the same form)
    elapsed, own, loss = timed_region()
    if dp:
        sys.stderr.write('bench.py: rank %d: %.4f ms per step (max over ranks %.4f)\n' % (rank, 1e3 * own / args.steps, 1e3 * elapsed / args.steps))
    if not dp and args.settled_probe_ms > 0:
        t_load = time.perf_counter()
        n_more = 0
        while time.perf_counter() - t_load < 1e-3 * args.settled_probe_ms:
            step(n_more)
            n_more += 1
        e2, _, _ = timed_region()
        settled = {'value': round(N * args.steps / e2, 2), 'ms_per_step': round(1e3 * e2 / args.steps, 4), 'steps': args.steps, 'warmup': args.warmup,
                   'steps_before': args.warmup + args.steps + n_more,
                   'what': 'the same W + K region timed again behind the contract\'s region and %d more steps (%.0f ms under load): the core clock has '
                           'left its idle level by then (profiles/r04_clock_ramp.txt).  Information; `value` is the first region.' % (n_more, args.settled_probe_ms)}
    ms_per_step = 1e3 * elapsed / args.steps
    value = N * world * args.steps / elapsed

    # ---- dominant-kernel timing with HIP events (same process, same steps, right after the timed region); for the block kernels the
    # events are attached to the dispatches themselves (hipExtLaunchKernelGGL inside the library): kernel begin / end timestamps, no
    # marker packets in the stream - the figure agrees with the rocprofv3 --kernel-trace --stats average ----
    roofline = None
    if rank != 0:
        # the probe steps contain the gradient all-reduce: EVERY rank has to run them (rank 0 alone would wait for its peers forever)
        for i in range(args.probe_steps):
            step(i)
        torch.cuda.synchronize(dev)
    if rank == 0:
        import ctypes
        lib = L.lib()
        # dominant kernel: the residual-block kernel (two 64->64 convs per launch, conv_block.hip) when the engine uses it,
        # otherwise the single-layer strip kernel
        hipnet = getattr(h.net, 'hip_generator', h.net)
        plan = hipnet.engine.plan_for(N, P, P, True)
        ops = plan.fwd + plan.bwd
        blocks = [a for name, a in ops if name == 'rumpy_conv_block']
        chains = [a for name, a in ops if name == 'rumpy_res_chain']      # runs of residual blocks as one persistent launch (conv_chain.hip); probe id 5 too
        rcabs = [a for name, a in ops if name in ('rumpy_rcab_fwd', 'rumpy_rcab_bwd', 'rumpy_rcab2_fwd', 'rumpy_rcab2_bwd')]      # share probe id 5 with the block kernel
        rcab2 = any(name == 'rumpy_rcab2_fwd' for name, _ in ops)
        use_block = len(blocks) + len(rcabs) + len(chains) > 0
        #  an fp8 line only if every one-launch block of the plan really runs the fp8 kernels (patches wider than 48 pixels do not)
        fp8_live = fp8 and use_block and not chains and all(getattr(a, 'w1_f8', None) for a in blocks + rcabs)
        if fp8 and not fp8_live:
            mfma_peak = MFMA_BF16_PEAK_TFLOPS
        wide_convs = [a for name, a in ops if name == 'rumpy_conv3x3' and a.cin_chunks == 4] if getattr(hipnet.engine, 'wide', False) else []
        hipnet.use_graph = False          # the probe records events around eager launches (a graph replay has none)
        lib.rumpy_probe_begin(5 if use_block else (3 if wide_convs else 1), max(80, len(blocks) + len(rcabs), len(wide_convs) + 8) * args.probe_steps + 8)
        for i in range(args.probe_steps):
            step(i)
        torch.cuda.synchronize(dev)
        tot = ctypes.c_double(0.0)
        n_launch = lib.rumpy_probe_end(ctypes.byref(tot))
        if n_launch > 0:
            avg_s = tot.value * 1e-3 / n_launch
            layer_flop = 2.0 * N * P * P * 64 * 576      # algorithmic FLOPs of one 64->64 3x3 layer
            tensor_bytes = N * P * P * 64 * 2
            # algorithmic bytes: every [N,P,P,64] bf16 tensor a launch must touch once, from the engine's launch plan
            if wide_convs:
                # every rumpy_conv3x3 launch of the step (body 256 -> 256 forward + data gradient, upsampler forward): mean flops / mean duration
                flop = sum(2.0 * a.N * a.H * a.W * 256 * 64 * a.cout_tiles * 9 for a in wide_convs) / len(wide_convs)
                tensor_bytes = 1
                tensors = [a.N * a.H * a.W * 2 * (256 + 64 * a.cout_tiles + 256 * sum(1 for f in ('mask', 'res1', 'res2') if getattr(a, f))) for a in wide_convs]
                kname = 'conv3x3_kernel<4> (Cin = 256 form of the 3x3 conv with fused epilogues: all launches of an EDSR 256 x 32 step)'
            elif use_block:
                flop = 2 * layer_flop                      # the halo-row recompute of the first conv is overhead, not counted
                # (a mask that travels as bytes - maskbits - is 1/16 of a tensor and not counted)
                tensors = [2 + sum(1 for f in ('t', 'res2') if getattr(a, f)) + (1 if (a.mask and not a.maskbits) else 0) for a in blocks] + \
                          [2 + sum(1 for f in (('u_in', 'x_out', 't', 'res2') if rcab2 else ('t', 't2', 't2_in', 'res2')) if getattr(a, f)) +
                           (1 if (not rcab2 and a.mask and not a.maskbits) else 0) for a in rcabs]
                kname = 'conv_block_kernel (residual block: two 3x3 convs 64->64 per launch, fwd + data-gradient launches)'
                if chains and not blocks and not rcabs:
                    # one launch = a whole run of blocks: flops and algorithmic tensors of all of them (per block: x | g read once at the chain's head
                    # only - the strip stays in LDS - so a block touches its T store, its OUT store, + res2 where given; the first block also its input)
                    nb = [a.nblocks for a in chains]
                    # ... + the single conv at the chain's outer end where the launch carries it (EDSR's body-end conv: forward it stores its output and
                    # reads the skip operand; backward it stores the chain's first input, which the launch then does NOT read)
                    ne = sum(1 for a in chains if a.edge_w)
                    flop = (2 * layer_flop * sum(nb) + layer_flop * ne) / len(nb)
                    tensors = [1 + sum(1 + (1 if b.t else 0) + (1 if b.res2 else 0) for b in a._blocks_host) + (0 if not a.edge_w else (2 if a.edge_res else 1))
                               for a in chains]
                    kname = ('block_chain_kernel (a run of %d residual blocks%s per persistent launch, halo rows handed over through the XCD L2; fwd + data-gradient '
                             'launches)' % (nb[0], ' + the body-end conv' if ne else ''))
                if fp8 and blocks and all(a.w1_f8 for a in blocks):
                    kname = 'conv_block_fp8_kernel (residual block per launch, both sweeps on the block-scaled fp8 MFMA; fwd + data-gradient launches)'
                if rcabs:
                    kname = 'rcab_kernel (residual channel-attention block per launch: two 3x3 convs 64->64 + attention gate; fwd + bwd launches)'
                    if rcab2:
                        kname = 'rcab2_kernel (residual channel-attention block per launch, gate applied by the consuming launch: two 3x3 convs 64->64; fwd + bwd launches)'
                    if fp8 and all(a.w1_f8 for a in rcabs):
                        kname = 'rcab_fp8_kernel (residual channel-attention block per launch, both sweeps on the block-scaled fp8 MFMA; fwd + bwd launches)'
            else:
                flop = layer_flop
                tensors = [2 + sum(1 for f in ('mask', 'res1', 'res2') if getattr(a, f))
                           for name, a in ops
                           if name == 'rumpy_conv3x3' and a.cin_chunks == 1 and a.cout_tiles == 1 and a.out_mode == 0]
                kname = 'conv3x3_strip_kernel (3x3 conv 64->64, fwd + dgrad launches)'
            alg_bytes = tensor_bytes * sum(tensors) / max(1, len(tensors))
            t_mfma, t_hbm = flop / (mfma_peak * 1e12), alg_bytes / (HBM_PEAK_GBPS * 1e9)
            tflops, gbps = flop / avg_s / 1e12, alg_bytes / avg_s / 1e9
            common = {'kernel': kname,
                      'avg_launch_us': round(avg_s * 1e6, 3), 'launches_timed': n_launch, 'launches_per_step': len(tensors),
                      'algorithmic_gflop_per_launch': round(flop / 1e9, 3), 'algorithmic_mb_per_launch': round(alg_bytes / 1e6, 3),
                      'mfma': {'achieved': round(tflops, 2), 'peak': mfma_peak, 'unit': 'TFLOP/s',
                               'frac': round(tflops / mfma_peak, 4)},
                      'hbm': {'achieved': round(gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(gbps / HBM_PEAK_GBPS, 4)}}
            # the launch sits at the ridge (C = 64): the binding roof is whichever limit takes longer for one launch
            if t_hbm >= t_mfma and not fp8:      # (the fp8 line keeps the matrix-pipe view beside the bf16 line's; `hbm` is next to it)
                roofline = {'bound': 'hbm', 'achieved': round(gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                            'frac': round(gbps / HBM_PEAK_GBPS, 4), 'traffic': None}
            else:
                roofline = {'bound': 'mfma', 'achieved': round(tflops, 2), 'peak': mfma_peak, 'unit': 'TFLOP/s',
                            'frac': round(tflops / mfma_peak, 4), 'traffic': None}
            roofline.update(common)
            # HBM bytes per launch from the PMC passes committed under profiles/ (not re-measured here; null when the kernel changed since)
            kind = 'conv3x3_cin256' if wide_convs else 'conv3x3_strip' if not use_block else ('rcab2_kernel' if rcab2 else 'rcab_kernel') if rcabs else 'block_chain_kernel' if (chains and not blocks) else 'conv_block_kernel'
            if use_block and kname.startswith(('conv_block_fp8_kernel', 'rcab_fp8_kernel')):      # the fp8 kernels have PMC entries of their own
                kind = 'rcab_fp8_kernel' if rcabs else 'conv_block_fp8_kernel'
            roofline['traffic'], roofline['traffic_source'] = pmc_traffic('%s:N%d:P%d' % (kind, N, P))

    # ---- informational: the step exactly as the reference's caller makes it (SISRInterface.train_batch, interface.py:97-101): batch on the HOST,
    # output returned to the HOST (keep_on_device=False): 15 MB up + 14 MB down over PCIe per step.  Never `value`. ----
    as_called = None
    if rank == 0 and world == 1 and src is None and meta_pool is None and not args.no_as_called:
        host_pool = [(x.cpu(), y.cpu()) for x, y in pool[:4]]
        for i in range(3):
            h.run_train(x=host_pool[i % 4][0], y=host_pool[i % 4][1])
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        n_ac = 20
        for i in range(n_ac):
            h.run_train(x=host_pool[i % 4][0], y=host_pool[i % 4][1])
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t1
        as_called = {'value': round(N * n_ac / dt, 1), 'unit': 'LR patches/s', 'ms_per_step': round(1e3 * dt / n_ac, 3), 'steps': n_ac,
                     'what': 'run_train(x, y) with pageable host tensors in and the output copied back to the host (keep_on_device=False), '
                             'as SISRInterface.train_batch calls it; PCIe-inclusive, informational'}

    # ---- CPU baseline: the oracle (torch-CPU fp32 restatement of the reference) on the host cores, rank 0, N=1 only ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import sr_oracle as O                  # the checker, timed as the CPU baseline: the ONLY use of oracle/ in this file
        torch.manual_seed(8)
        onet = O.build_oracle({'blindqrcan': 'contrastiveblindqrcan', 'edsr256': 'edsr'}.get(args.model, args.model), scale=4,
                              **{'qrcan': dict(style='standard', include_q_layer=True, num_metadata=5), 'blindqrcan': BLIND,
                                 'edsr256': dict(num_features=256, num_blocks=32, res_scale=0.1)}.get(args.model, {}))
        cpu_meta = torch.rand(args.cpu_batch, 5, 1, 1) if args.model == 'qrcan' else None
        oh = O.OracleHandler(onet, lr=1e-4, scheduler='cosine_annealing_warm_restarts', scheduler_params=SCHED)
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        try:                                             # container CPU quota (cgroup v2): "<quota> <period>" or "max <period>"
            q, per = open('/sys/fs/cgroup/cpu.max').read().split()
            if q != 'max':
                usable = max(1, min(usable, int(int(q) / int(per))))
        except Exception:
            pass
        cores = max(1, min(usable, args.cpu_threads))   # torch CPU convs stop scaling (and thrash) far below 256 threads
        torch.set_num_threads(cores)
        nb = args.cpu_batch                              # bounded sample: a few patches, same per-patch work
        xb, yb = synthetic_batch(1234, nb, lr_hw=P)
        t1 = time.perf_counter()
        oh.run_train(xb, yb, extra_channels=cpu_meta)
        warm = time.perf_counter() - t1
        # bounded sample: at least 3 timed steps, then until about 12 s of CPU work (40 steps at most)
        best, total, timed = 1e30, 0.0, 0
        while timed < 3 or (total < 12.0 and timed < 40):
            t2 = time.perf_counter()
            oh.run_train(xb, yb, extra_channels=cpu_meta)
            dt = time.perf_counter() - t2
            best, total, timed = min(best, dt), total + dt, timed + 1
        cpu = {'value': round(nb / best, 3), 'unit': 'LR patches/s', 'cores': torch.get_num_threads(), 'kind': 'port',
               'sample': '1 warm-up + %d timed %s x4 train steps of %d %dx%d patches (%.1f s of CPU work; value = best step, mean step %.3f s), '
                         'torch CPU fp32 oracle; host reports %d logical CPUs, %d usable under the cgroup quota'
                         % (timed, {'edsr': 'EDSR-baseline'}.get(args.model, args.model.upper()), nb, P, P, total, total / timed, os.cpu_count() or 0, usable),
               's_per_step': round(best, 3), 'warmup_s': round(warm, 3)}

    if rank == 0:
        if fp8 and not fp8_live:
            sys.stderr.write('bench.py: --precision fp8, but this plan has no fp8 launch (patches wider than 48 pixels?): the line is labelled bf16\n')
        line = {'metric': '%dpx LR patches/sec (train step) %s x4 %s' % (P, args.model.upper(), 'fp8 opt-in' if fp8_live else 'bf16'), 'value': round(value, 2), 'unit': 'LR patches/s',
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': 'fp8 (e4m3 forward / e5m2 gradient MFMA operands of the residual-block launches, fp32 accumulation; bf16 storage and bf16 MFMA elsewhere)' if fp8_live else 'bf16',
                'data': ('synthetic uint8 images in HBM, patches cropped/flipped/converted on the GPU every step (device patch pipeline), '
                         'random-init weights (seed 8)') if args.device_patches else
                        'synthetic uniform[0,1) DIV2K-shaped patches, random-init weights (seed 8)',
                'config': {'workload': ('EDSR-baseline x4 (64 feats x 16 blocks)' if args.model == 'edsr' else 'EDSR x4 (256 feats x 32 blocks, the reference\'s div2k/edsr.toml)' if args.model == 'edsr256' else 'RCAN x4 (10 groups x 20 RCABs, 64 feats)' + {'qrcan': ' + meta-attention q-layers, 5 metadata entries', 'blindqrcan': ' + frozen contrastive degradation encoder (256-vector) driving q-layers in group 0 block 0'}.get(args.model, '')) +
                                       ' train step, %dx%d LR patches, batch %d per GPU' % (P, P, N),
                           'global_batch': N * world, 'parallelism': 'dp%d' % world, 'optimizer': 'Adam lr 1e-4 + cosine warm restarts per batch',
                           'loss': float(loss), 'train_tflops': round(value * flop_per_patch / 1e12, 2),
                           'train_mfma_frac': round(value * flop_per_patch / 1e12 / (MFMA_BF16_PEAK_TFLOPS * world), 4),      # (whole step, against the bf16 peak: most of a step's launches are bf16 in either precision)
                           'train_mfma_frac_peak_tflops': MFMA_BF16_PEAK_TFLOPS,
                           'precision': args.precision},
                'roofline': roofline, 'cpu_baseline': cpu, 'as_called': as_called, 'settled': settled, 'clock_settle': settle,
                # what the collectives really ran on (the driver's scaling run can check that RCCL saw N ranks on N devices)
                'distributed': {'world_size': dist.get_world_size() if dp else 1, 'backend': dist.get_backend() if dp else None,
                                'device_count': torch.cuda.device_count(), 'ranks_on_one_device': bool(one_device),
                                'allreduce_form': getattr(getattr(h, 'data_parallel', None), 'form', None) if dp else None,
                                'forms_loss': forms_loss, 'forms': forms_ms,       # ms per step of every form timed in this run (--allreduce-form auto: inline, then early); `value` is the faster
                                'grad_allreduce_mb': round(getattr(h.net, 'hip_generator', h.net).flat_g.numel() * 4 / 1e6, 2) if dp else 0.0}}
        print(json.dumps(line), flush=True)
    if dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""which blocks of BASELINE config 5's generator (blind QRCAN 10 x 20, precision='fp8') carry the whole-gradient error of the
joint-loss step?  The fp32 oracle's gradient is taken ONCE per case; the HIP step is then repeated with chosen blocks kept on the bf16 kernels
(RUMPY_FP8_BF16_BLOCKS, engine.py::_fp8_bf16_blocks) and the class numbers of tests/test_fp8_gpu.py::_fp8_class_check are printed per policy.

  python tests/tools/fp8_block_ablation.py [supmoco|moco|frozen] [--singles] [--budget SECONDS] [--policies 'first:20;every:20;...']

Output -> profiles/r06_fp8_block_ablation.txt (copied by hand from gpurun_out/)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import sr_oracle as O  # noqa: E402
import test_fp8_gpu as T  # noqa: E402

NBLK = 200


def stats(named_h, named_o):
    """whole-gradient relative error, cosine, worst 3x3 tensor, and the share of the squared error carried by each block's own tensors"""
    num = den = dot = gg = 0.0
    worst = (0.0, None)
    per = {}
    for (k, p), (_, q) in zip(named_h, named_o):
        g, r = p.grad.detach().float().cpu().double().reshape(-1), q.grad.double().reshape(-1)
        e = float((g - r).pow(2).sum())
        num += e; den += float(r.pow(2).sum()); dot += float(g @ r); gg += float(g.pow(2).sum())
        parts = k.split('.')
        key = 'body.%s.body.%s' % (parts[1], parts[3]) if (parts[0] == 'body' and len(parts) > 4 and parts[2] == 'body') else '.'.join(parts[:2])
        per[key] = per.get(key, 0.0) + e
        if p.dim() == 4 and p.shape[-1] == 3 and float(r.norm()) > 0:
            rel = float((g - r).norm() / r.norm())
            if rel > worst[0]:
                worst = (rel, k)
    return (num / den) ** 0.5, dot / (gg * den) ** 0.5, worst, {k: v / num for k, v in per.items()}


def on_e4m3_grid(sd):
    """the 3x3 filters of the 200 RCABs rounded to the values their fp8 images hold (per-tensor power-of-two scale of rumpy_fp8_pack)"""
    out = {}
    for k, v in sd.items():
        p = k.split('.')
        if len(p) == 7 and p[0] == 'body' and p[2] == 'body' and p[4] == 'body' and p[5] in ('0', '2') and p[6] == 'weight' and v.dim() == 4 and v.shape[-1] == 3:
            scale = 2.0 ** (T.exponent_for(float(v.abs().max())) - 127)
            v = T.q8(v, scale) * scale
        out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('case', nargs='?', default='supmoco', choices=['supmoco', 'moco', 'frozen'])
    ap.add_argument('--singles', action='store_true', help='every block alone in bf16 (200 steps)')
    ap.add_argument('--budget', type=float, default=900.0, help='seconds; the loop stops when they are spent')
    ap.add_argument('--policies', default=None)
    ap.add_argument('--wq', action='store_true', help="diagnostic: the 400 block filters of BOTH nets pre-rounded to the e4m3 grid of rumpy_fp8_pack's scale "
                    "(the fp8 filter images are then exact: what is left is the activations' and gradients' rounding)")
    ap.add_argument('--random', type=int, default=0, help='K random 20-block subsets (how much does the number move with WHICH blocks?)')
    a = ap.parse_args()
    t0 = time.time()
    if a.case == 'frozen':
        onet = O.build_oracle('contrastiveblindqrcan', **T.BLIND_FULL)
        sd = O.seeded_pipeline_state(onet, 4105)
        if a.wq:
            sd = {k: v for k, v in sd.items()}
            g = on_e4m3_grid({k[2:]: v for k, v in sd.items() if k.startswith('G.')})
            sd.update({'G.' + k: v for k, v in g.items()})
        onet.load_state_dict(sd)
        oh = O.OracleHandler(onet, lr=1e-4)
        x, y = O.synthetic_batch(4106, 2, lr_hw=48, scale=4)
        oh.run_train(x, y)

        def step():
            h = T._handler('contrastiveblindqrcan', precision='fp8', metadata_list=None, block_encoder_loading=True, lr=1e-4, **T.BLIND_FULL)
            h.net.load_state_dict(sd)
            h.run_train(x=x, y=y)
            return h
    else:
        case = T.joint_case(*{'supmoco': ('supmoco', 3, 'pre_q'), 'moco': ('moco', 2, 'none')}[a.case])
        if a.wq:
            case['state_hook'] = on_e4m3_grid
        oh = T.joint_oracle(case)

        def step():
            h = T.joint_handler(case)
            h.run_train(x=case['x'], y=case['y'], **case['kw'])
            return h
    print('# case %s: oracle step %.1f s' % (a.case, time.time() - t0), flush=True)

    def run(policy, tag=None):
        os.environ['RUMPY_FP8_BF16_BLOCKS'] = policy
        t1 = time.time()
        h = step()
        eng = h.net.hip_generator.engine if hasattr(h.net, 'hip_generator') else h.net.G.engine
        kept = len(eng.f8_bf16_blocks)
        w, c, worst, per = stats(h.net.G.named_parameters(), oh.net.G.named_parameters())
        print('%-28s bf16 blocks %3d  whole %.4e  cosine %.5f  worst 3x3 %.3e (%s)  [%.1f s]'
              % (tag or policy, kept, w, c, worst[0], worst[1], time.time() - t1), flush=True)
        del h
        torch.cuda.empty_cache()
        return w, per

    base, per = run('none')
    top = sorted(per.items(), key=lambda kv: -kv[1])[:12]
    print('# share of the squared whole-gradient error by the tensors of (all fp8): ' + ', '.join('%s %.1f%%' % (k, 100 * v) for k, v in top), flush=True)
    by_group = {}
    for k, v in per.items():
        p = k.split('.')
        g = p[1] if p[0] == 'body' else k
        by_group[g] = by_group.get(g, 0.0) + v
    print('# ... by residual group: ' + ', '.join('%s %.1f%%' % (k, 100 * v) for k, v in sorted(by_group.items(), key=lambda kv: -kv[1])), flush=True)
    run(','.join(str(b) for b in range(NBLK)), 'all 200 in bf16')
    pols = a.policies.split(';') if a.policies else (
        ['every:20', 'first:10', 'first:20', 'last:10', 'last:20', 'every:10', 'first:40', 'last:40']
        + [','.join(str(20 * g + b) for b in range(20)) for g in range(10)])
    names = {','.join(str(20 * g + b) for b in range(20)): 'group %d (20 blocks)' % g for g in range(10)}
    for p in pols:
        if time.time() - t0 > a.budget:
            print('# budget spent', flush=True)
            return
        run(p, names.get(p))
    if a.random:
        import random
        rng = random.Random(6)
        vals = []
        for i in range(a.random):
            sub = sorted(rng.sample(range(NBLK), 20))
            w, _ = run(','.join(str(b) for b in sub), 'random 20 #%d' % i)
            vals.append(w)
        print('# %d random 20-block subsets: min %.3e  mean %.3e  max %.3e' % (len(vals), min(vals), sum(vals) / len(vals), max(vals)), flush=True)
    if a.singles:
        res = []
        for b in range(NBLK):
            if time.time() - t0 > a.budget:
                print('# budget spent at block %d' % b, flush=True)
                break
            w, _ = run(str(b), 'block %d alone' % b)
            res.append((base - w, b))
        res.sort(reverse=True)
        print('# ranking (drop of the whole-gradient error with block k alone in bf16): ' + ', '.join('%d: %.2e' % (b, d) for d, b in res[:40]), flush=True)


if __name__ == '__main__':
    main()

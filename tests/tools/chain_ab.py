"""Body-chain kernel (tests/tools/csrc/conv_body_chain.hip, experimental library) against the per-block launches of a training plan (GPU box):
    python tests/tools/chain_ab.py [reps]
The forward and backward runs of consecutive rumpy_conv_block launches of an EDSR plan are re-issued as ONE rumpy_body_chain launch each:
every buffer the launches write must be bitwise the same; then both forms are timed.  Timing variants of the kernel: make -C tests/tools/csrc
clean all EXTRA=-DCHAIN_ABL=<v> (1 = no hand-off at all, 3 = also no T stores; >= 9 adds phase stamps, printed here)."""
import ctypes as C, os, sys, tempfile, numpy as np, torch
os.environ['RUMPY_NO_CHAIN'] = '1'      # the plan's per-block launches are what this tool re-issues as a chain
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import sr_oracle as O
from rumpy_amd.shared_framework.models import define_model
import gpu_utils as G

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
kw = dict(scale=4, num_blocks=16, res_scale=0.1)
x, y = O.synthetic_batch(670, 32, lr_hw=48, scale=4)
xd, yd = x.cuda(), y.cuda()
h = define_model('edsr', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, lr=1e-3, **kw)
h.net.load_state_dict(O.seeded_state_dict(O.build_oracle('edsr', **kw), 826))
net = h.net
for _ in range(2):
    net.fused_l1_forward_backward(xd, yd)
eng = net.engine
plan = eng.plan_for(32, 48, 48, True)
stream = torch.cuda.current_stream().cuda_stream
N, H, W = plan.N, plan.H, plan.W
strips = N * ((H + 5) // 6)
flags = torch.zeros(int(G.exp_lib().rumpy_body_chain_flag_bytes(N, H)), dtype=torch.uint8, device='cuda')
epoch = torch.zeros(1, dtype=torch.int32, device='cuda')
status = torch.zeros(1, dtype=torch.int32, device='cuda')
by_ptr = {t.data_ptr(): t for t in plan.keep}


def runs(ops):
    """maximal runs of consecutive block launches, each reading the previous one's output"""
    out, i = [], 0
    while i < len(ops):
        j = i
        if ops[i][0] == 'rumpy_conv_block' and ops[i][1].res_mode == 0:
            j = i + 1
            while j < len(ops) and ops[j][0] == 'rumpy_conv_block' and ops[j][1].x == ops[j - 1][1].out:
                j += 1
        if j - i >= 2:
            out.append(ops[i:j])
        i = max(j, i + 1)
    return out


def written(blocks):
    return [by_ptr[p] for _, a in blocks for p in (a.t, a.out, a.maskbits if a.relu1 else None) if p and p in by_ptr]


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for lab, ops, backward in (('forward', plan.fwd, 0), ('backward', plan.bwd, 1)):
    for blocks in runs(ops):
        tab = (G.ChainBlock * len(blocks))(*[G.ChainBlock(x=a.x, w1=a.w1, b1=a.b1, w2=a.w2, b2=a.b2, res2=a.res2, t=a.t, out=a.out, maskbits=a.maskbits,
                                                           scale1=a.scale1, scale2=a.scale2) for _, a in blocks])
        dev = torch.from_numpy(np.frombuffer(bytes(tab), dtype=np.uint8).copy()).cuda()
        args = G.BodyChainArgs(blocks=dev.data_ptr(), nblocks=len(blocks), N=N, H=H, W=W, backward=backward, fmt=0, flags=flags.data_ptr(),
                               epoch=epoch.data_ptr(), status=status.data_ptr())
        per_block = lambda: eng._run(blocks, stream)
        chain = lambda: G.exp_call('rumpy_body_chain', args, stream)
        per_block()
        torch.cuda.synchronize()
        ref = [t.clone() for t in written(blocks)]
        for t in written(blocks):
            t.zero_()
        bad = 0
        for _ in range(10):
            chain()
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(a.view(torch.uint8), b.view(torch.uint8)) else 1 for a, b in zip(ref, written(blocks)))
        print('%s: %d blocks, %d buffers compared x 10 launches: %d differ; status 0x%x' % (lab, len(blocks), len(ref), bad, int(status.item())))
        print('   one launch per block %.1f us, chain %.1f us' % (timed(per_block), timed(chain)))
        if flags.numel() > strips * 8:          # stamp build
            raw = flags.cpu().numpy()[strips * 8:].view(np.uint64).reshape(strips, 8, 16).astype(np.float64) / 100.0
            names = ['start', 'x rows', 'sweep a', 'acked', 'flag seen', 'halo loaded', 'halo in lds', 'conv1 swept', 'T done', 'conv2 swept', 'OUT done', 'stored']
            rel = raw[:, :, :12] - raw[:, :, 0:1]
            for rh in (0, 1):
                print('   row half %d, us from the start of block %d: ' % (rh, len(blocks) // 2)
                      + '  '.join('%s %.2f' % (nm, rel[:, 4 * rh:4 * rh + 4, i].mean()) for i, nm in enumerate(names)))
            print('   shader clock %.2f GHz, block starts spread over %.1f us' % ((raw[:, :, 13] - raw[:, :, 12]).mean() / rel[:, :, 11].mean() / 10.0,
                                                                                  raw[:, :, 0].max() - raw[:, :, 0].min()))

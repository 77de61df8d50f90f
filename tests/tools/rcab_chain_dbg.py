import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gpu_utils import BF16, DEV, PackedConv, RcabChainArgs, RcabChainBlock, exp_call, exp_lib, stream, to_dev_bytes      # (the kernel lives in tests/tools/csrc since round 6)
from rumpy_amd import _lib as L
which = sys.argv[1]
N, H, W, nblk, cr = [int(v) for v in sys.argv[2:7]]
gen = np.random.default_rng(1)
f32 = lambda lo, hi, *s: torch.from_numpy(gen.uniform(lo, hi, s).astype(np.float32)).to(DEV)
convs = [(PackedConv(f32(-0.05, 0.05, 64, 64, 3, 3).cpu(), f32(-0.1, 0.1, 64).cpu()), PackedConv(f32(-0.05, 0.05, 64, 64, 3, 3).cpu(), f32(-0.1, 0.1, 64).cpu())) for _ in range(nblk)]
mlps = [(f32(-0.3, 0.3, cr, 64), f32(-0.3, 0.3, cr), f32(-0.3, 0.3, 64, cr), f32(-0.3, 0.3, 64)) for _ in range(nblk)]
rnd = lambda: torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(BF16)
x0 = rnd()
t1s, t2s, ys = [rnd() for _ in range(nblk)], [rnd() for _ in range(nblk)], [rnd() for _ in range(nblk)]
dt2s, dt1s, dxs = [rnd() for _ in range(nblk)], [rnd() for _ in range(nblk)], [rnd() for _ in range(nblk)]
mbs = [torch.zeros(N, H, W, 8, dtype=torch.uint8, device=DEV) for _ in range(nblk)]
means, hids, gates = ([torch.rand(*s, device=DEV) for _ in range(nblk)] for s in ((N, 64), (N, cr), (N, 64)))
dzs = [torch.zeros(N, 64, device=DEV) for _ in range(nblk)]
status = torch.zeros(1, dtype=torch.int32, device=DEV)
work = torch.zeros(int(exp_lib().rumpy_rcab_chain_work_bytes(N, H)), dtype=torch.uint8, device=DEV)
xchg = torch.zeros(N * ((H + 5) // 6) * 512, dtype=torch.uint8, device=DEV)
rec = lambda **kw: RcabChainBlock(**{k: (v.data_ptr() if torch.is_tensor(v) else v) for k, v in kw.items()})
if which == 'fwd':
    recs = [rec(x=x0 if b == 0 else ys[b - 1], w1=pa.w_fwd, b1=pa.b_packed, w2=pb.w_fwd, b2=pb.b_packed, t=t1s[b], t2=t2s[b], out=ys[b], maskbits=mbs[b],
                ca_w1=mlps[b][0], ca_b1=mlps[b][1], ca_w2=mlps[b][2], ca_b2=mlps[b][3], mean=means[b], hidden=hids[b], gate=gates[b]) for b, (pa, pb) in enumerate(convs)]
else:
    recs = [rec(x=x0 if k == 0 else dxs[nblk - k], w1=convs[b][1].w_dgrad, w2=convs[b][0].w_dgrad, t=dt1s[b], t2=dt2s[b], t2_in=t2s[b], out=dxs[b], maskbits=mbs[b],
                ca_w1=mlps[b][0], ca_b1=mlps[b][1], ca_w2=mlps[b][2], ca_b2=mlps[b][3], hidden=hids[b], gate=gates[b], dz=dzs[b]) for k, b in enumerate(reversed(range(nblk)))]
tab = to_dev_bytes((RcabChainBlock * nblk)(*recs))
a = RcabChainArgs(blocks=tab.data_ptr(), nblocks=nblk, N=N, H=H, W=W, cr=cr, backward=0 if which == 'fwd' else 1, work=work.data_ptr(), work_bytes=work.numel(),
                    xchg=xchg.data_ptr(), xchg_bytes=xchg.numel(), status=status.data_ptr(), fake_xcc=0, force_sc1=1)
print('launch', which, N, H, W, nblk, cr, flush=True)
exp_call('rumpy_rcab_chain', a, stream())
torch.cuda.synchronize()
print('ok status', hex(int(status.item())), 'finite', all(bool(torch.isfinite(t.float()).all()) for t in (ys if which == 'fwd' else dxs)), flush=True)
reps = 20
for _ in range(3):
    exp_call('rumpy_rcab_chain', a, stream())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    exp_call('rumpy_rcab_chain', a, stream())
e1.record()
torch.cuda.synchronize()
print('%s chain: %.1f us per launch = %.2f us per block' % (which, e0.elapsed_time(e1) / reps * 1e3, e0.elapsed_time(e1) / reps * 1e3 / nblk), flush=True)
if 'RCC_STAMPS' in os.environ.get('RUMPY_EXP_LIB', ''):
    import ctypes
    fn = getattr(ctypes.CDLL(os.environ['RUMPY_EXP_LIB']), 'rumpy_debug_rcc_stamps')
    fn.argtypes = [ctypes.c_void_p]
    nwg = N * ((H + 5) // 6)
    buf = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device=DEV)
    assert fn(buf.data_ptr()) == 0
    exp_call('rumpy_rcab_chain', a, stream())
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(nwg, 8, 16).astype(np.float64)
    names = {'fwd': ['start', '-', 'sweep a', '-', '-', 'conv1 swept', 'T written', 'conv2 swept', 'pool gathered', 'gate ready', 'OUT written', 'stored'],
             'bwd': ['start', 'ds reduced', 'halo in', 'ds gathered', 'd_t2 tile', 'conv2^T swept', 'T written', 'conv1^T swept', '-', '-', 'dx written', 'stored']}[which]
    rel = (raw - raw[:, :, :1]) * 0.01
    for rh in (0, 1):
        print('   row half %d, us from the start of block %d: ' % (rh, nblk // 2) + '  '.join('%s %.2f' % (nm, rel[:, 4 * rh:4 * rh + 4, i].mean()) for i, nm in enumerate(names) if nm != '-'))
    print('   block starts spread over %.1f us' % ((raw[:, :, 0].max() - raw[:, :, 0].min()) * 0.01))

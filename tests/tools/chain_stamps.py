"""phase stamps of the middle block of the product's residual-block chain (conv_chain.hip built with -DCHAIN_STAMPS: RUMPY_AMD_LIB=build_abl/CHAIN_STAMPS/librumpy_amd.so),
forward and data-gradient launch of an EDSR-baseline training plan at 32 x 48 x 48"""
import ctypes, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sr_oracle as O
from rumpy_amd.shared_framework.models import define_model
fn = getattr(ctypes.CDLL(os.environ['RUMPY_AMD_LIB']), 'rumpy_debug_chain_stamps')
fn.argtypes = [ctypes.c_void_p]
x, y = O.synthetic_batch(670, 32, lr_hw=48, scale=4)
h = define_model('edsr', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, lr=1e-3, scale=4)
xd, yd = x.cuda(), y.cuda()
for _ in range(3):
    h.net.fused_l1_forward_backward(xd, yd)
eng = h.net.engine
plan = eng.plan_for(32, 48, 48, True)
stream = torch.cuda.current_stream().cuda_stream
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device='cuda')
assert fn(buf.data_ptr()) == 0
names = ['start', 'x rows', 'sweep a', 'acked', 'flag seen', 'halo in lds', 'conv1 swept', 'T done', 'conv2 swept', 'OUT done', 'stored']
for lab, ops in (('forward', plan.fwd), ('backward', plan.bwd)):
    chain = [op for op in ops if op[0] == 'rumpy_res_chain']
    for _ in range(3):
        eng._run(chain, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng._run(chain, stream)
    e1.record()
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(256, 8, 16).astype(np.float64)
    rel = (raw - raw[:, :, :1]) * 0.01
    print('%s chain (stamp build): %.1f us per launch = %.2f us per block' % (lab, e0.elapsed_time(e1) / 20 * 1e3, e0.elapsed_time(e1) / 20 * 1e3 / chain[0][1].nblocks))
    for rh in (0, 1):
        print('   row half %d, us from the start of block %d: ' % (rh, chain[0][1].nblocks // 2) + '  '.join('%s %.2f' % (nm, rel[:, 4 * rh:4 * rh + 4, i].mean()) for i, nm in enumerate(names)))
    print('   block starts spread over %.1f us' % ((raw[:, :, 0].max() - raw[:, :, 0].min()) * 0.01))

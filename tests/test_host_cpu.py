"""CPU-only tests (-m "not gpu"): host logic of the drop-in boundary, the C-ABI library surface, data-parallel
gradient averaging over gloo.  No kernel is executed here (there is no GPU and no CPU fallback)."""
import ctypes
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np
import pytest
import torch

from oracle import sr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'rumpy_amd.h')


def _lib_or_skip():
    from rumpy_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib


def _handler(name, eval_mode=False, **kw):
    from rumpy_amd.shared_framework.models import define_model
    return define_model(name, model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=eval_mode,
                        checkpoint_load=False, loss_masking=False, metadata_list=None, **kw)


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    _lib = _lib_or_skip()
    with open(HEADER) as f:
        src = f.read()
    product = set(re.findall(r'\b(rumpy_[a-z0-9_]+)\s*\(', src))
    assert not [s for s in product if s.startswith(('rumpy_debug', 'rumpy_probe'))]       # the drop-in boundary carries no measurement hooks
    with open(os.path.join(os.path.dirname(HEADER), 'rumpy_amd_debug.h')) as f:            # ... they are declared beside it
        declared = product | set(re.findall(r'\b(rumpy_[a-z0-9_]+)\s*\(', f.read()))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    h = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared:
        assert hasattr(h, s), s
    lib = _lib.lib()
    assert lib.rumpy_abi_version() == 6
    assert lib.rumpy_wgrad_slab_floats(4) == 64 * 576 + 64 and lib.rumpy_wgrad_slab_floats(1) == 16 * 576 + 16


def test_ctypes_structs_match_the_header_layout():
    """compile a C probe against include/rumpy_amd.h and compare sizeof / offsetof with the ctypes mirrors"""
    _lib = _lib_or_skip()
    pairs = {'rumpy_conv_args': _lib.ConvArgs, 'rumpy_head_fwd_args': _lib.HeadFwdArgs, 'rumpy_enc_conv_args': _lib.EncConvArgs, 'rumpy_rcab_args': _lib.RcabArgs, 'rumpy_rcab2_args': _lib.Rcab2Args, 'rumpy_res_chain_block': _lib.ResChainBlock, 'rumpy_res_chain_args': _lib.ResChainArgs, 'rumpy_enc_bn_args': _lib.EncBnArgs, 'rumpy_enc_bn_bwd_args': _lib.EncBnBwdArgs, 'rumpy_pixel_shuffle_args': _lib.PixelShuffleArgs, 'rumpy_tail_wide_args': _lib.TailWideArgs, 'rumpy_head_wgrad_args': _lib.HeadWgradArgs,
             'rumpy_tail_fwd_args': _lib.TailFwdArgs, 'rumpy_tail_dgrad_args': _lib.TailDgradArgs, 'rumpy_conv4d_tail_args': _lib.Conv4dTailArgs,
             'rumpy_nchw_to_nhwc4_args': _lib.NchwToNhwc4Args, 'rumpy_wgrad_job': _lib.WgradJob, 'rumpy_reduce_item': _lib.ReduceItem,
             'rumpy_pack_item': _lib.PackItem, 'rumpy_ca_mlp_fwd_args': _lib.CaMlpFwdArgs, 'rumpy_ca_scale_args': _lib.CaScaleArgs,
             'rumpy_ca_bwd_reduce_args': _lib.CaBwdReduceArgs, 'rumpy_ca_mlp_bwd_args': _lib.CaMlpBwdArgs,
             'rumpy_ca_bwd_apply_args': _lib.CaBwdApplyArgs, 'rumpy_adam_hyper': _lib.AdamHyper, 'rumpy_adam_args': _lib.AdamArgs,
             'rumpy_sumsq_args': _lib.SumsqArgs, 'rumpy_eval_post_args': _lib.EvalPostArgs, 'rumpy_block_args': _lib.BlockArgs,
             'rumpy_ca_fwd_fused_args': _lib.CaFwdFusedArgs, 'rumpy_ca_bwd_fused_args': _lib.CaBwdFusedArgs,
             'rumpy_q_mlp_item': _lib.QMlpItem, 'rumpy_q_mlpn_item': _lib.QMlpNItem, 'rumpy_ssim_args': _lib.SsimArgs, 'rumpy_patch_item': _lib.PatchItem, 'rumpy_patch_args': _lib.PatchArgs,
             'rumpy_finish_reduce_args': _lib.FinishReduceArgs, 'rumpy_update_item': _lib.UpdateItem, 'rumpy_adam_pack_args': _lib.AdamPackArgs,
             'rumpy_qca_layer': _lib.QcaLayer, 'rumpy_qca_args': _lib.QcaArgs,
             'rumpy_dconv_args': _lib.DconvArgs, 'rumpy_dconv_wgrad_args': _lib.DconvWgradArgs, 'rumpy_mse_args': _lib.MseArgs, 'rumpy_op': _lib.Op,
             'rumpy_sgemm_args': _lib.SgemmArgs, 'rumpy_fp8_pack_item': _lib.Fp8PackItem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rumpy_amd.h"', 'int main(void){']
    for cname, st in pairs.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname, _ in st._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        lines.append('printf("\\n");')
    lines.append('return 0;}')
    d = tempfile.mkdtemp()
    with open(os.path.join(d, 'p.c'), 'w') as f:
        f.write('\n'.join(lines))
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 'p.c'), '-o', os.path.join(d, 'p')])
    out = subprocess.check_output([os.path.join(d, 'p')]).decode().strip().splitlines()
    for line in out:
        parts = line.split()
        st = pairs[parts[0]]
        assert int(parts[1]) == ctypes.sizeof(st), parts[0]
        for (fname, _), off in zip(st._fields_, parts[2:]):
            assert getattr(st, fname).offset == int(off), (parts[0], fname)


def test_argument_validation_runs_without_a_gpu():
    _lib = _lib_or_skip()
    lib = _lib.lib()
    assert lib.rumpy_conv3x3(_lib.ConvArgs(), None) == -1 and b'null' in lib.rumpy_last_error()
    assert lib.rumpy_adam_step(_lib.AdamArgs(), None) == -1
    assert lib.rumpy_wgrad_grouped(None, 0, 4, 0, None) == -1
    assert lib.rumpy_probe_begin(9, 1) == -1


# ------------------------------------------------------------------ plugin boundary
def test_registry_scans_handlers_like_the_reference():
    from rumpy_amd.shared_framework.models import available_models, define_model
    assert available_models['edsr'].endswith('SISR.models.advanced.handlers.EDSRHandler')
    assert available_models['rcan'].endswith('SISR.models.advanced.handlers.RCANHandler')
    with pytest.raises(KeyError):
        define_model('nonexistent')


def test_handlers_expose_the_reference_contract(golden_dir):
    with open(os.path.join(golden_dir, 'g8_params.json')) as f:
        g = json.load(f)
    h = _handler('edsr', scale=4, lr=2e-4, some_unknown_kwarg=1, scheduler='multi_step_lr',
                 scheduler_params={'milestones': [2, 4], 'gamma': 0.5})
    assert h.model_name == 'edsr' and h.colorspace == 'rgb' and h.im_input == 'unmodified'
    assert isinstance(h.criterion, torch.nn.L1Loss) and h.legacy_load and h.curr_epoch == 0 and h.eval_request_loss
    assert [(k, list(v.shape)) for k, v in h.net.state_dict().items()] == [tuple(x) for x in map(tuple, g['edsr_keys'])]
    assert h.print_parameters() == g['edsr_baseline_count'] == 1517571
    assert h.get_learning_rate() == 2e-4 and h.verify_eval() is True
    for _ in range(2):
        h.learning_rate_scheduler.step()
    assert abs(h.get_learning_rate() - 1e-4) < 1e-12
    r = _handler('rcan', eval_mode=True, scale=4)
    assert [(k, list(v.shape)) for k, v in r.net.state_dict().items()] == [tuple(x) for x in map(tuple, g['rcan_keys'])]
    assert r.print_parameters() == g['rcan_count'] == 15592355
    assert r.optimizer is None and r.model_name == 'rcan'
    with pytest.raises(RuntimeError, match='eval mode'):
        r.run_train(x=torch.zeros(1, 3, 8, 8), y=torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match='scheduler not implemented'):
        _handler('edsr', scheduler='bogus', scheduler_params={})


def test_default_init_consumes_rng_like_the_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g8_init_seed8.npz'))
    for name, kw in (('edsr', dict(scale=4)), ('rcan', dict(scale=4, n_resgroups=2, n_resblocks=2))):
        torch.manual_seed(8)
        h = _handler(name, eval_mode=True, **kw)
        sd = h.net.state_dict()
        for k in g.files:
            if k.startswith(name + '/') and not k.endswith('checksum'):
                assert np.array_equal(sd[k.split('/', 1)[1]].numpy(), g[k]), k
        chk = sum(float(v.double().sum()) for v in sd.values())
        assert abs(chk - float(g[name + '/checksum'])) < 1e-9


def test_compute_without_gpu_fails_loudly():
    h = _handler('edsr', scale=4, num_blocks=1)
    x, y = O.synthetic_batch(1, 1, lr_hw=8, scale=4)
    with pytest.raises(RuntimeError, match='no CPU path'):
        h.run_train(x=x, y=y)
    with pytest.raises(RuntimeError, match='no CPU path'):
        h.run_eval(x=x)
    with pytest.raises(RuntimeError, match='no CPU path'):
        h.optimizer.step()
    with pytest.raises(RuntimeError, match='multiple of 64|n_feats = 64'):
        from rumpy_amd.engine import SREngine
        hh = _handler('edsr', scale=4, num_blocks=1, num_features=272)       # beyond the widest kernel (widths up to 256 run embedded in the next kernel width)
        SREngine(hh.net._spec(), torch.device('cpu'))


def test_checkpoint_layout_matches_reference(golden_dir):
    sched = dict(scheduler='cosine_annealing_warm_restarts', scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7})
    for name, kw in (('edsr', dict(scale=4, num_blocks=2)), ('rcan', dict(scale=4, n_resgroups=1, n_resblocks=2))):
        with open(os.path.join(golden_dir, 'g9_checkpoint_%s.json' % name)) as f:
            g = json.load(f)
        h = _handler(name, lr=1e-3, **kw, **sched)
        st = h.save_model('train_model', extract_state_only=True)
        assert sorted(st.keys()) == g['top_keys']
        assert sorted(st['optimizer'].keys()) == g['optimizer_keys']
        assert sorted(st['optimizer']['param_groups'][0].keys()) == g['optimizer_param_group_keys']
        assert sorted(st['optimizer']['state'][0].keys()) == g['optimizer_state_entry_keys']
        assert sorted(st['scheduler_G'].keys()) == g['scheduler_keys']
        assert st['model_name'] == g['model_name'] and st['model_epoch'] == 0
        # file round trip + a reference-shaped (oracle) net and a stock Adam accept it
        h.set_epoch(2)
        h.save_model('train_model')
        loaded = torch.load(os.path.join(h.model_save_dir, 'train_model_2'), map_location='cpu', weights_only=False)
        onet = O.build_oracle(name, **kw)
        onet.load_state_dict(loaded['network'])
        torch.optim.Adam(onet.parameters(), lr=1.0).load_state_dict(loaded['optimizer'])
        h2 = _handler(name, lr=5e-4, **kw, **sched)
        h2.model_save_dir = h.model_save_dir
        h2.load_model('train_model', 2, config_changes={'values_changed': {"root['internal_params']['lr']": {'new_value': 7e-4}}})
        assert h2.curr_epoch == 2 and abs(h2.get_learning_rate() - 7e-4) < 1e-12
        for a, b in zip(h.net.parameters(), h2.net.parameters()):
            assert torch.equal(a, b)
        # legacy prefixes are stripped
        legacy = {('model.module.' + k): v for k, v in loaded['network'].items()}
        assert list(h.legacy_switch(legacy).keys()) == list(loaded['network'].keys())


def test_reference_checkpoint_with_empty_optimizer_state_loads():
    """a reference checkpoint saved before the first step has an empty Adam state"""
    h = _handler('edsr', scale=4, num_blocks=1, lr=1e-3)
    onet = O.build_oracle('edsr', scale=4, num_blocks=1)
    ref_opt = torch.optim.Adam(onet.parameters(), lr=3e-4)
    state = {'network': onet.state_dict(), 'model_name': 'edsr', 'model_epoch': 5, 'optimizer': ref_opt.state_dict(), 'steps': None}
    h.load_model('train_model', 5, preloaded_state=state)
    assert h.curr_epoch == 5 and h.optimizer.step_count == 0 and abs(h.get_learning_rate() - 3e-4) < 1e-12
    assert float(h.optimizer.flat_m.abs().sum()) == 0.0


def test_flat_parameter_views_survive_module_moves():
    h = _handler('edsr', scale=4, num_blocks=1)
    net = h.net
    assert net.head[0].weight.data_ptr() == net.flat_p.data_ptr()
    assert sum(p.numel() for p in net.parameters()) == net.flat_p.numel()
    before = net.flat_p.clone()
    net.double().float()          # goes through nn.Module._apply -> re-flattened
    assert net.head[0].weight.data_ptr() == net.flat_p.data_ptr() and torch.equal(before, net.flat_p)
    assert net.tail[1].bias.grad.data_ptr() == net.flat_g[-3:].data_ptr()


# ------------------------------------------------------------------ data parallel (gloo, world size 2)
_DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from rumpy_amd.parallel import GradientAverager, broadcast_parameters, bucket_bounds
rank = int(os.environ['RANK'])
dist.init_process_group('gloo')
g = torch.arange(10007, dtype=torch.float32) * (rank + 1)
avg = GradientAverager(flat_grad=g, bucket_elems=1000)
assert avg.world_size == 2 and len(avg.buckets) == 11 and avg.buckets[0] == (9007, 10007) and avg.buckets[-1] == (0, 7)
mult = avg.average_sum()                                          # the SUM lands in the buffer; the mean factor goes to the fused Adam (grad_mult)
assert mult == 0.5 and torch.equal(g, torch.arange(10007, dtype=torch.float32) * 3.0), rank
g1 = torch.arange(10007, dtype=torch.float32) * (rank + 1)
assert GradientAverager(flat_grad=g1, bucket_elems=1000).average() is None                          # everybody else: the mean, in place
assert torch.equal(g1, torch.arange(10007, dtype=torch.float32) * 1.5), rank
g3 = torch.arange(10007, dtype=torch.float32) * (rank + 1)        # one bucket: the blocking-form collective from the caller's stream
avg3 = GradientAverager(flat_grad=g3)
assert avg3.inline and len(avg3.buckets) == 1 and avg3.form == 'inline' and avg3.average_sum() == 0.5 and not avg3.pending
assert torch.equal(g3, torch.arange(10007, dtype=torch.float32) * 3.0), rank
# two-phase form (SREngine.backward(on_ready=...)): the upper part is launched early by begin(ptr), average() covers the rest
g2 = torch.arange(10007, dtype=torch.float32) * (rank + 1)
avg2 = GradientAverager(flat_grad=g2, bucket_elems=1000)
avg2.begin(g2.data_ptr() + 4 * 6001)
assert avg2.early_lo == 6001 and len(avg2.pending) == 5          # [9007,10007) ... [6001,7007): LAST parameters first
g2[:6001] += 0.0                                                  # "remaining weight gradients" written while the upper part is in flight
assert avg2.average_sum() == 0.5
assert avg2.early_lo is None and not avg2.pending
assert torch.equal(g2, torch.arange(10007, dtype=torch.float32) * 3.0), rank
avg2.begin(g2.data_ptr())                                         # boundary at the start of the buffer: nothing to split
assert avg2.early_lo is None
# parameters outside the flat buffer (the blind pipeline's trainable encoder): one coalesced all-reduce, mean written back into .grad;
# a parameter without a gradient on one rank enters as zero and receives the mean there too (every replica takes the same step)
from rumpy_amd.parallel import ParameterGradientAverager
ps = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2)), torch.zeros(4)]
flat = torch.zeros(15)
ps[0].grad = flat.view(3, 5); flat += rank + 1.0                   # a view of a flat gradient buffer, like the encoder trunk's
if rank == 0:
    ps[1].grad = torch.full((7,), 4.0)
ps[2].grad = torch.tensor([1.0, -1.0]) * (rank + 1)
pavg = ParameterGradientAverager(ps)
assert pavg.active and len(pavg.params) == 3
for rep in range(2):
    pavg.average()
assert torch.equal(flat, torch.full((15,), 1.5)) and ps[0].grad.data_ptr() == flat.data_ptr()
assert torch.equal(ps[1].grad, torch.full((7,), 2.0))          # (4 + 0) / 2 on both ranks, then (2 + 2) / 2
assert torch.equal(ps[2].grad, torch.tensor([1.5, -1.5]))
class Net: pass
n = Net(); n.flat_p = torch.full((5,), float(rank)); n._packed_version = 1
broadcast_parameters(n, src=0)
assert float(n.flat_p.sum()) == 0.0 and n._packed_version is None
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_gradient_averager_gloo_world_size_2():
    from rumpy_amd.parallel import bucket_bounds
    assert bucket_bounds(10, 4) == [(6, 10), (2, 6), (0, 2)]
    d = tempfile.mkdtemp()
    script = os.path.join(d, 'w.py')
    with open(script, 'w') as f:
        f.write(_DP_WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_device_patch_source_needs_a_gpu():
    """SURVEY.md 8f.1 path: no CPU fallback either."""
    import numpy as np
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from rumpy_amd.sr_tools.device_patches import DevicePatchSource
    lr = np.zeros((8, 8, 3), np.uint8)
    hr = np.zeros((16, 16, 3), np.uint8)
    with pytest.raises(RuntimeError, match='no CPU path'):
        DevicePatchSource([lr], [hr], 2, 4)


def test_patch_item_numpy_mirror_matches_the_c_struct():
    import ctypes as C

    from rumpy_amd import _lib as L
    from rumpy_amd.sr_tools import device_patches as DP
    assert DP.ITEM_BYTES == C.sizeof(L.PatchItem)
    for name, _ in L.PatchItem._fields_:
        assert DP.ITEM_DTYPE.fields[name][1] == getattr(L.PatchItem, name).offset, name


def test_reference_written_checkpoint_loads_into_the_handler(golden_dir):
    """G11: a `train_model_2` file written by the real reference handler (tests/golden/make_golden_checkpoint.py) goes
    through our BaseModel.load_model: weights, Adam moments / step count, scheduler state and epoch arrive unchanged."""
    import os
    import shutil
    import tempfile

    import numpy as np
    import torch

    from rumpy_amd.shared_framework.models import define_model
    src = os.path.join(golden_dir, 'g11_ref_checkpoint')
    tmp = tempfile.mkdtemp()
    shutil.copy(os.path.join(src, 'train_model_2'), tmp)
    h = define_model('edsr', model_save_dir=tmp, device=torch.device('cpu'), eval_mode=False, checkpoint_load=False,
                     loss_masking=False, metadata_list=None, scale=2, num_features=64, num_blocks=1, res_scale=0.1, lr=1e-3,
                     scheduler='cosine_annealing_warm_restarts', scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7})
    state = h.load_model('train_model', 2)
    assert state['model_name'] == 'edsr' and h.curr_epoch == 2
    ref = torch.load(os.path.join(tmp, 'train_model_2'), map_location='cpu', weights_only=False)
    sd = h.net.state_dict()
    assert list(sd.keys()) == list(ref['network'].keys())
    for k in sd:
        assert torch.equal(sd[k].cpu(), ref['network'][k]), k
    # optimizer: per-parameter moments in parameter order, step count, hyper-parameters
    osd = h.optimizer.state_dict()
    assert osd['param_groups'][0]['lr'] == ref['optimizer']['param_groups'][0]['lr']
    for i, st in ref['optimizer']['state'].items():
        mine = osd['state'][i]
        assert torch.equal(mine['exp_avg'].cpu().reshape(-1), st['exp_avg'].reshape(-1)), i
        assert torch.equal(mine['exp_avg_sq'].cpu().reshape(-1), st['exp_avg_sq'].reshape(-1)), i
        assert float(mine['step']) == float(st['step']) == 2.0
    exp = np.load(os.path.join(src, 'expected.npz'))
    assert np.isclose(h.get_learning_rate(), float(exp['lr_at_save']), rtol=1e-12)
    # and the file our save_model writes has the reference's structure (round trip through the reference's own loader is
    # covered by the key/layout fixture G9)
    h.save_model('resaved')
    again = torch.load(os.path.join(tmp, 'resaved_2'), map_location='cpu', weights_only=False)
    assert sorted(again.keys()) == sorted(ref.keys())
    for k in ref['network']:
        assert torch.equal(again['network'][k], ref['network'][k])


def test_ssim_oracle_properties_and_no_cpu_path():
    """oracle/ssim_oracle.py: identical images give 1; the crop (5 px) equals the window radius, so the result cannot depend on
    the filter's boundary mode (the HIP kernel relies on that and never evaluates padded pixels); the product has no CPU path."""
    import numpy as np
    import pytest
    import torch
    from scipy.ndimage import gaussian_filter

    from oracle import ssim_oracle as SO
    gen = np.random.default_rng(5)
    a = gen.uniform(0, 1, (37, 52))
    b = np.clip(a + gen.normal(0, 0.05, a.shape), 0, 1)
    assert SO.ssim_plane(a, a) == 1.0
    v = SO.ssim_plane(a, b)
    assert 0.5 < v < 1.0

    def with_mode(mode):
        f = lambda z: gaussian_filter(z, sigma=1.5, truncate=3.5, mode=mode)
        ux, uy, uxx, uyy, uxy = f(a), f(b), f(a * a), f(b * b), f(a * b)
        S = ((2 * ux * uy + 1e-4) * (2 * (uxy - ux * uy) + 9e-4)) / ((ux ** 2 + uy ** 2 + 1e-4) * ((uxx - ux * ux) + (uyy - uy * uy) + 9e-4))
        return S[5:-5, 5:-5].mean()
    assert abs(with_mode('constant') - v) < 1e-12 and abs(with_mode('nearest') - v) < 1e-12
    if not torch.cuda.is_available():
        from rumpy_amd.sr_tools.metrics import Metrics
        with pytest.raises(RuntimeError, match='no CPU path'):
            Metrics().run_ssim(a[None, None], b[None, None])


def test_qrcan_module_tree_has_the_reference_key_and_creation_order(golden_dir):
    """state_dict keys (= registration order: final_body first, a block's final_body and q_node before its body) against the key list
    the REAL reference QRCAN produced (golden G12), and seed-for-seed identical initial weights against the oracle restatement."""
    from rumpy_amd.SISR.models.attention_manipulators.architectures import QRCAN
    g = np.load(os.path.join(golden_dir, 'g12_qrcan_small_train.npz'))
    kw = dict(scale=2, n_feats=16, n_resgroups=2, n_resblocks=2, reduction=16)
    torch.manual_seed(8)
    net = QRCAN(style='standard', include_q_layer=True, num_metadata=5, **kw)
    assert list(net.state_dict().keys()) == [str(k) for k in g['keys']]
    torch.manual_seed(8)
    onet = O.build_oracle('qrcan', style='standard', include_q_layer=True, num_metadata=5, **kw)
    for (k, a), (k2, b) in zip(net.state_dict().items(), onet.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    # selective placement: only group 1, first block of each group
    sel = QRCAN(style='standard', include_q_layer=True, num_metadata=5, selective_meta_blocks=[False, True], num_q_layers_inner_residual=1, **kw)
    qk = [k for k in sel.state_dict() if 'q_node' in k]
    assert qk and all(k.startswith('body.1.body.0.q_node') for k in qk)
    with pytest.raises(RuntimeError):
        QRCAN(style='modulate', **kw)
    # ParaCALayer's num_layers other than 2: the REAL reference's keys (golden G24) and the oracle's seed-8 weights
    for depth in (1, 3, 6):
        g24 = np.load(os.path.join(golden_dir, 'g24_qrcan_qdepth%d_small_train.npz' % depth))
        torch.manual_seed(8)
        netd = QRCAN(style='standard', include_q_layer=True, num_metadata=5, num_layers_in_q_layer=depth, **kw)
        assert list(netd.state_dict().keys()) == [str(k) for k in g24['keys']]
        torch.manual_seed(8)
        od = O.build_oracle('qrcan', style='standard', include_q_layer=True, num_metadata=5, num_layers_in_q_layer=depth, **kw)
        for (k, a), (k2, b) in zip(netd.state_dict().items(), od.state_dict().items()):
            assert k == k2 and torch.equal(a, b), k
    with pytest.raises(RuntimeError, match='FC layers'):
        QRCAN(style='standard', include_q_layer=True, num_metadata=5, num_layers_in_q_layer=9, **kw)


@pytest.mark.parametrize('style', ['max_concat', 'mini_concat', 'extended_attention', 'softmax'])
def test_styled_qcalayer_module_tree_has_the_reference_keys_and_creation_order(golden_dir, style):
    """the QCALayer styles whose gate MLP reads the attribute vector: state_dict keys against the REAL reference QRCAN's (golden G19) and the
    default initialisation under seed 8 against the reference's per-tensor checksums (same creation order -> same weights)"""
    from rumpy_amd.SISR.models.attention_manipulators.architectures import QRCAN
    g = np.load(os.path.join(golden_dir, 'g19_qrcan_styles_small_train.npz'))
    torch.manual_seed(8)
    net = QRCAN(style=style, include_q_layer=False, num_metadata=5, scale=2, n_feats=16, n_resgroups=2, n_resblocks=2, reduction=16)
    assert list(net.state_dict().keys()) == [str(k) for k in g[style + '.keys']]
    init8 = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in net.state_dict().values()])
    assert np.allclose(init8, g[style + '.init8'], rtol=0, atol=1e-12)


def test_qmodel_metadata_vector_follows_the_reference_selection_rules():
    """QModel.generate_channels / channel_concat_logic (attention_manipulators/__init__.py:84-162) on the host, no GPU needed."""
    import tempfile
    from rumpy_amd.shared_framework.models import available_models, define_model
    assert available_models['qrcan'].endswith('attention_manipulators.handlers.QRCANHandler')
    h = define_model('qrcan', device='cpu', model_save_dir=tempfile.mkdtemp(), eval_mode=True, style='standard', include_q_layer=True,
                     metadata=['noise', 'blur'], n_resgroups=1, n_resblocks=1)
    assert h.num_metadata == 2 and h.model_name == 'qrcan' and h.colorspace == 'augmented_rgb'
    x = torch.zeros(3, 3, 8, 8)
    md = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    ch = h.generate_channels(x, md, [('jpeg',), ('blur',), ('other',), ('noise',)])
    assert tuple(ch.shape) == (3, 2, 1, 1) and torch.equal(ch[:, :, 0, 0], md[:, [1, 3]])
    one = define_model('qrcan', device='cpu', model_save_dir=tempfile.mkdtemp(), eval_mode=True, style='standard', include_q_layer=True,
                       n_resgroups=1, n_resblocks=1)
    assert one.metadata == ['qpi'] and one.num_metadata == 1
    ch1 = one.generate_channels(x, torch.tensor([[0.1], [0.2], [0.3]]), [('qpi',)])
    assert tuple(ch1.shape) == (3, 1, 1, 1) and torch.allclose(ch1.flatten(), torch.tensor([0.1, 0.2, 0.3]))
    with pytest.raises(RuntimeError):
        h.generate_channels(x, None, [('blur',)])
    wide = define_model('qrcan', device='cpu', model_save_dir=tempfile.mkdtemp(), eval_mode=True, style='standard', include_q_layer=True,
                        metadata=['blur_kernel', 'noise'], n_resgroups=1, n_resblocks=1)
    assert wide.num_metadata == 11        # 2 names + the 9 extra entries of a PCA-reduced blur kernel (:46-47)
    assert wide.net.body[0].body[0].q_node.attribute_integrator[0].weight.shape[:2] == (32, 11)
    # the product path has no CPU fallback: running the model without a GPU fails loudly
    with pytest.raises(RuntimeError):
        h.run_eval(x=x, metadata=md, metadata_keys=[('jpeg',), ('blur',), ('other',), ('noise',)])


def test_qrcan_default_style_scale_qpi_matches_the_reference_known_answers(golden_dir):
    """QRCANHandler in its default configuration (style 'modulate', metadata ['qpi']): scale_qpi / generate_channels on the host against
    the vectors the REAL reference handler produced (golden G14)."""
    import tempfile
    from rumpy_amd.shared_framework.models import define_model
    g = np.load(os.path.join(golden_dir, 'g14_qrcan_modulate_small_train.npz'))
    q = torch.from_numpy(g['kat_q'])
    h = define_model('qrcan', device='cpu', model_save_dir=tempfile.mkdtemp(), eval_mode=True, n_resgroups=1, n_resblocks=1, clamp=True,
                     min_mu=-0.1, max_mu=0.9)
    assert h.style == 'modulate' and h.metadata == ['qpi'] and h.num_metadata == 1
    assert np.array_equal(h.scale_qpi(q).numpy(), g['kat_64_clamped'])
    ch = h.generate_channels(torch.zeros(5, 3, 4, 4), q.reshape(5, 1), [('qpi',)])
    assert tuple(ch.shape) == (5, 64, 1, 1) and np.array_equal(ch.numpy(), g['kat_64_clamped'])
    with pytest.raises(RuntimeError):      # q-layers need the 'standard' style
        define_model('qrcan', device='cpu', model_save_dir=tempfile.mkdtemp(), eval_mode=True, n_resgroups=1, n_resblocks=1, include_q_layer=True)


def test_packed_relu_mask_bit_trick_equals_the_float_comparison():
    """block_common.hpp::relu_keep keeps a bf16 half iff its value is > 0, computed on packed pairs with integer operations: checked here
    for all 65536 bit patterns in both halves against the float comparison (NaNs with a clear sign bit count as > 0 - documented)."""
    m = np.arange(65536, dtype=np.uint32)
    pair = (m << 16) | m[::-1]
    t = pair & 0x7fff7fff
    keep = ((((t + 0x7fff7fff) & 0xffffffff) & ~pair & 0x80008000) >> 15) * 0xffff & 0xffffffff
    hi = (m << 16).astype(np.uint32).view(np.float32)
    lo = (m[::-1] << 16).astype(np.uint32).view(np.float32)
    with np.errstate(invalid='ignore'):
        want = np.where(hi > 0, 0xffff0000, 0).astype(np.uint32) | np.where(lo > 0, 0xffff, 0).astype(np.uint32)
    nan = np.isnan(hi) | np.isnan(lo)
    assert np.array_equal(keep[~nan], want[~nan])
    pos_nan_hi = np.isnan(hi) & ((m & 0x8000) == 0)
    assert np.all((keep[pos_nan_hi] & 0xffff0000) == 0xffff0000)


def test_blind_pipeline_host_logic_encoder_checkpoint_and_normalisation():
    """ContrastiveBlindSRPipeline host side (contrastive_blind_sr.py:14-61, 229-239): the encoder state is taken out of a contrastive
    checkpoint (MoCo-style files hold it under 'encoder_q.'), frozen, and the embedding normalisation follows the reference formulas."""
    import tempfile
    from rumpy_amd.shared_framework.models import define_model
    from rumpy_amd.SISR.models.blur_kernel_blind_sr.contrastive_blind_sr import load_encoder_model
    enc = O.OracleEncoder()
    sd = O.seeded_encoder_state(enc, 31)
    d = tempfile.mkdtemp()
    moco = os.path.join(d, 'moco_ckpt')
    torch.save({'model_name': 'mococontrastive', 'network': {**{'encoder_q.' + k: v for k, v in sd.items()},
                                                              **{'encoder_k.' + k: v * 0 for k, v in sd.items()}, 'queue': torch.zeros(3)}}, moco)
    got = load_encoder_model(moco, torch.device('cpu'))
    assert list(got.keys()) == list(sd.keys()) and all(torch.equal(got[k], sd[k]) for k in sd)
    supcon = os.path.join(d, 'supcon_ckpt')
    torch.save({'model_name': 'supcon', 'network': sd}, supcon)
    kw = dict(device='cpu', model_save_dir=d, eval_mode=True, n_resgroups=1, n_resblocks=1, style='standard', include_q_layer=True)
    h = define_model('contrastiveblindqrcan', pre_trained_encoder_weights=supcon, **kw)
    assert h.model_name == 'blind_qrcan' and h.net.G.num_metadata == 256
    assert all(torch.equal(v, sd[k]) for k, v in h.net.E.state_dict().items())
    assert not any(p.requires_grad for p in h.net.E.parameters()) and all(p.requires_grad for p in h.net.G.parameters())
    # only the generator's parameters reach the optimizer (as in the reference), through the fused flat Adam
    ht = define_model('contrastiveblindqrcan', block_encoder_loading=True, **{**kw, 'eval_mode': False})
    assert type(ht.optimizer).__name__ == 'FlatAdam' and len(ht.optimizer.param_groups[0]['params']) == len(list(ht.net.G.parameters()))
    # normalisation of the embedding
    v = torch.tensor([[1.0, 3.0], [2.0, 5.0]])
    hn = define_model('contrastiveblindqrcan', block_encoder_loading=True, encoding_normalization_type='minmax',
                      encoding_normalization_params={'min': 1.0, 'max': 5.0}, **kw)
    assert torch.allclose(hn.net.normalize(v, hn.net.encoding_normalization_params), (v - 1.0) / 4.0)
    hm = define_model('contrastiveblindqrcan', block_encoder_loading=True, encoding_normalization_type='meanstd',
                      encoding_normalization_params={'mean': 2.0, 'std': 0.5}, **kw)
    assert torch.allclose(hm.net.normalize(v, hm.net.encoding_normalization_params), (v - 2.0) / 0.5)
    with pytest.raises(RuntimeError):
        define_model('contrastiveblindqrcan', block_encoder_loading=True, encoding_normalization_type='zscore', **kw)
    with pytest.raises(RuntimeError):      # no CPU fallback for the pipeline either
        hn.run_eval(x=torch.zeros(1, 3, 8, 8))


def test_basic_models_resolve_with_the_reference_keys_and_refuse_the_cpu(golden_dir):
    """define_model('srcnn' | 'vdsr') (BASELINE config 0 and its deeper sibling): handler attributes and state_dict keys as the REAL
    reference handlers have them (fixture G15), default parameter counts, and no CPU compute path."""
    from rumpy_amd.shared_framework.models import define_model
    g = np.load(os.path.join(golden_dir, 'g15_basic_small_train.npz'))
    for name, kw, count in (('srcnn', {}, 57281), ('vdsr', {}, 665921)):
        h = define_model(name, model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, checkpoint_load=False,
                         loss_masking=False, **kw)
        assert sum(p.numel() for p in h.net.parameters()) == count
        assert [h.colorspace, h.im_input, h.model_name, type(h.criterion).__name__] == [str(a) for a in g[name + '.attrs'][:4]]
        assert (h.grad_clip == 0.1) if name == 'vdsr' else (h.grad_clip is None)
        if name == 'srcnn':
            assert list(h.net.state_dict().keys()) == [str(k) for k in g['srcnn.keys']]
        with pytest.raises(RuntimeError):
            h.run_train(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
        with pytest.raises(RuntimeError):
            h.run_eval(torch.zeros(1, 1, 8, 8))


def test_safe_image_save_numpy_path_matches_the_reference_expression(tmp_path):
    """rumpy/sr_tools/visualization.py:31-62 on host arrays: clip(im * 255 / max_val, 0, 255).astype(uint8) (truncation), CHW -> HWC, files readable"""
    from PIL import Image
    from rumpy_amd.sr_tools.visualization import safe_image_save, to_uint8_hwc
    gen = np.random.default_rng(5)
    im = gen.uniform(-0.2, 1.2, (2, 3, 9, 13)).astype(np.float32)
    im[0, 0, 0, 0], im[0, 1, 0, 0], im[0, 2, 0, 0] = 0.999, 254.9999 / 255, 1.0
    u8 = to_uint8_hwc(im)
    ref = np.clip(im.transpose(0, 2, 3, 1) * 255 / 1, 0, 255).astype(np.uint8)
    assert u8.dtype == np.uint8 and np.array_equal(u8, ref) and tuple(u8[0, 0, 0]) == (254, 254, 255)
    safe_image_save(im, str(tmp_path), ['a.png', 'sub/b.png'], config='rgb')
    assert np.array_equal(np.asarray(Image.open(tmp_path / 'a.png')), ref[0]) and np.array_equal(np.asarray(Image.open(tmp_path / 'sub' / 'b.png')), ref[1])
    safe_image_save(torch.from_numpy(im) * 2, str(tmp_path), ['c.png', 'd.png'], config='rgb', max_val=2)
    assert np.array_equal(np.asarray(Image.open(tmp_path / 'c.png')), ref[0])


# ------------------------------------------------------------------ contrastive encoder training: host side
def test_contrastive_handlers_contract_on_cpu():
    """define_model('mococontrastive' | 'supmoco'): reference state_dict keys, the fused optimizer over the query encoder's flat buffer,
    the key encoder frozen, checkpoints the blind pipeline's encoder loader reads, and a loud failure instead of a CPU training step."""
    from oracle import contrastive_oracle as CO
    from rumpy_amd.SISR.models.blur_kernel_blind_sr.contrastive_blind_sr import load_encoder_model
    for name, okind in (('mococontrastive', 'mococontrastive'), ('supmoco', 'supmoco'), ('weakcon', 'weakcon')):
        h = _handler(name, model_name='default', crop_count=3, lr=1e-3)
        onet = CO.OracleContrastiveHandler(okind, crop_count=3).net
        assert list(h.net.state_dict().keys()) == list(onet.state_dict().keys())
        assert [tuple(v.shape) for v in h.net.state_dict().values()] == [tuple(v.shape) for v in onet.state_dict().values()]
        q, k = h.net.encoder_q, h.net.encoder_k
        assert all(p.requires_grad for p in q.parameters()) and not any(p.requires_grad for p in k.parameters())
        assert all(torch.equal(a, b) for a, b in zip(q.parameters(), k.parameters()))
        assert type(h.optimizer).__name__ == 'FlatAdam' and h.optimizer.net is h.net
        assert h.net.flat_p.numel() == 1278784 == sum(p.numel() for p in q.parameters())
        assert q.E[0].weight.data_ptr() == h.net.flat_p.data_ptr() and q.mlp[2].bias.grad.data_ptr() == h.net.flat_g[-256:].data_ptr()
        assert torch.allclose(h.net.queue.norm(dim=0), torch.ones(8192), atol=1e-5) and int(h.net.queue_ptr) == 0
        assert h.model_name == name and h.colorspace == 'rgb' and h.im_input == 'unmodified' and h.eval_request_loss is False
        if not torch.cuda.is_available():
            with pytest.raises(RuntimeError, match='GPU only|no CPU'):
                h.run_train(x=torch.rand(2, 9, 16, 16), y=torch.tensor([[0.8, 0.0, 1.0], [0.0, 0.3, 0.0]]),
                            metadata_keys=[('gaussian_noise_scale',), ('poisson_noise_scale',), ('gray_noise_boolean',)])
        h.save_model('enc')
        enc_sd = load_encoder_model(os.path.join(h.model_save_dir, 'enc_0'), torch.device('cpu'))      # <name>_<epoch>
        assert list(enc_sd.keys()) == list(q.state_dict().keys()) and torch.equal(enc_sd['mlp.2.bias'], q.mlp[2].bias.detach())
    for bad in (dict(model_name='resnet18'), dict(model_name='default', dropdown=8)):
        with pytest.raises(RuntimeError, match='HIP path'):
            _handler('supmoco', **bad)


_KEYS_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from rumpy_amd.regression.models.contrastive_learning.encoding_models import Encoder
from rumpy_amd.regression.models.contrastive_learning.moco import MoCo
from rumpy_amd.regression.models.contrastive_learning.supmoco import SupMoCo
rank = int(os.environ['RANK'])
dist.init_process_group('gloo')
torch.manual_seed(0)
m = MoCo(base_encoder=Encoder, K=64)
keys = torch.full((4, 256), float(rank + 1))
# every rank enqueues the keys of BOTH ranks, in rank order (the enqueue itself is one HIP launch: no CPU path, checked on the GPU)
allk, stride, _ = m._keys_of_all_ranks(torch.stack([keys, keys * 7], dim=1).reshape(8, 256), stride=2)
assert stride == 1 and allk.shape == (8, 256) and torch.equal(allk[:4], torch.ones(4, 256)) and torch.equal(allk[4:], torch.full((4, 256), 2.0))
try:
    m._dequeue_and_enqueue(keys)
    raise SystemExit('a CPU enqueue must fail loudly')
except RuntimeError as e:
    assert 'no CPU path' in str(e)
# replicas start from rank 0's module: the trainable flat buffer in one broadcast, then everything else it holds (key encoder, queue, pointer,
# BatchNorm statistics)
from rumpy_amd.parallel import broadcast_parameters
torch.manual_seed(100 + rank)
m2 = MoCo(base_encoder=Encoder, K=64)
m2.encoder_q.flatten(); m2.encoder_k.flatten()
m2.queue_ptr[0] = 8 * rank
m2.encoder_q.E[1].running_mean += rank
broadcast_parameters(m2)
ref = [torch.zeros_like(t) for t in m2.state_dict().values()]
for t, r in zip(m2.state_dict().values(), ref):
    r.copy_(t); dist.broadcast(r, src=0)
assert all(torch.equal(t, r) for t, r in zip(m2.state_dict().values(), ref)) and int(m2.queue_ptr) == 0
s = SupMoCo(device='cpu', base_encoder=Encoder, K=64, positives_per_class=2)
s.register_classes(5)
allk, stride, alll = s._keys_of_all_ranks(keys, 1, torch.tensor([rank, rank, 3, 4]))
assert allk.shape == (8, 256) and alll.tolist() == [0, 0, 3, 4, 1, 1, 3, 4]
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_moco_queue_takes_the_keys_of_every_rank_gloo_world_size_2():
    d = tempfile.mkdtemp()
    script = os.path.join(d, 'k.py')
    with open(script, 'w') as f:
        f.write(_KEYS_WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29641', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_wide_edsr_and_x3_construct_like_the_reference_and_other_widths_are_refused():
    """div2k/edsr.toml: 256 features x 32 blocks -> the reference's 43,089,923 parameters and key order (G8); scale 3 builds the conv F -> 9F
    upsampler; a width the kernels do not take fails when the engine is built, loudly"""
    h = _handler('edsr', scale=4, num_features=256, num_blocks=32, res_scale=0.1)
    assert sum(p.numel() for p in h.net.parameters()) == 43089923
    o = O.build_oracle('edsr', scale=4, num_features=256, num_blocks=32, res_scale=0.1)
    assert list(h.net.state_dict().keys()) == list(o.state_dict().keys())
    assert h.net.supports_fused_l1 is False and type(h.optimizer).__name__ == 'FlatAdam'
    h3 = _handler('edsr', scale=3, num_blocks=2)
    assert tuple(h3.net.tail[0][0].weight.shape) == (576, 64, 3, 3) and h3.net.supports_fused_l1 is True
    assert [tuple(v.shape) for v in h3.net.state_dict().values()] == [tuple(v.shape) for v in O.build_oracle('edsr', scale=3, num_blocks=2).state_dict().values()]
    from rumpy_amd.engine import SREngine
    # widths between the kernel widths are embedded in the next one (architectures._embed): reference shapes outside, padded filters inside
    h96 = _handler('edsr', scale=2, num_features=96, num_blocks=1)
    o96 = O.build_oracle('edsr', scale=2, num_features=96, num_blocks=1)
    assert [(k, tuple(v.shape)) for k, v in h96.net.state_dict().items()] == [(k, tuple(v.shape)) for k, v in o96.state_dict().items()]
    assert tuple(h96.net.body[0].body[0].weight.shape) == (128, 128, 3, 3) and h96.net.real_numel() == sum(p.numel() for p in o96.parameters())
    sd96 = O.seeded_state_dict(o96, 5)
    h96.net.load_state_dict(sd96)
    assert all(torch.equal(v, sd96[k]) for k, v in h96.net.state_dict().items())
    w = h96.net.body[0].body[0].weight.detach()
    assert float(w[96:].abs().sum()) == 0.0 and float(w[:, 96:].abs().sum()) == 0.0
    assert [tuple(st['exp_avg'].shape) for st in h96.optimizer.state_dict()['state'].values()] == [tuple(v.shape) for v in o96.state_dict().values()]
    for bad in (dict(num_features=320), dict(scale=5)):
        try:
            hb = _handler('edsr', num_blocks=1, **{**dict(scale=2), **bad})
        except (RuntimeError, NotImplementedError):
            continue
        with pytest.raises(RuntimeError, match='n_feats|scale'):
            SREngine(hb.net._spec(), torch.device('cpu'))


def test_bench_launcher_refuses_more_gpus_than_the_node_has_and_relays_the_children(tmp_path):
    """`python bench.py --gpus N` without RANK starts the ranks itself (bench.py::launch_ranks).  More GPUs than the KFD topology shows
    (a fabricated one-GPU topology here; the count never touches HIP) -> non-zero exit with a message before any child starts; with
    RUMPY_BENCH_ONE_DEVICE=1 on this GPU-less container the children start and fail loudly (no GPU), and the parent returns their code
    instead of hanging."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'RUMPY_BENCH_ONE_DEVICE') and
           not k.endswith('_VISIBLE_DEVICES')}            # (this container sets HIP_VISIBLE_DEVICES to the empty list)
    for i, simd in enumerate((0, 1024)):
        (tmp_path / str(i)).mkdir()
        (tmp_path / str(i) / 'properties').write_text('simd_count %d\n' % simd)
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'],
                       env=dict(env, RUMPY_KFD_TOPOLOGY=str(tmp_path)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 2 and b'--gpus 8, but this node exposes 1 GPU' in p.stderr and b'{"metric"' not in p.stdout
    if not torch.cuda.is_available():
        env['RUMPY_BENCH_ONE_DEVICE'] = '1'
        p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode != 0 and b'{"metric"' not in p.stdout


def test_launcher_counts_gpus_from_the_kfd_topology_without_hip(tmp_path):
    """bench.py's launcher parent must never initialise HIP: it counts the KFD topology's nodes with simd_count > 0 and
    applies the *_VISIBLE_DEVICES lists itself.  A fabricated topology (2 CPU agents + 8 GPUs, the shape of an 8 x MI355X node) stands in."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_for_count', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for i in range(10):
        d = tmp_path / str(i)
        d.mkdir()
        (d / 'properties').write_text('cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n' % ((96, 0) if i < 2 else (0, 1024)))
    count = lambda **env: bench.count_gpus_without_hip(str(tmp_path), env)
    assert count() == 8
    assert count(HIP_VISIBLE_DEVICES='0,1,2,3') == 4
    assert count(ROCR_VISIBLE_DEVICES='4,5', HIP_VISIBLE_DEVICES='0,1,2,3') == 2       # HIP's indices refer to what ROCR left: 2 and 3 are invalid
    assert count(HIP_VISIBLE_DEVICES='0,9,1') == 1                                      # the list ends at its first invalid entry
    assert count(CUDA_VISIBLE_DEVICES='') == 0
    assert count(ROCR_VISIBLE_DEVICES='GPU-deadbeef00000000,1') == 2
    assert bench.count_gpus_without_hip(str(tmp_path / 'missing'), {}) is None
    # and the parent of `bench.py --gpus N` has not loaded torch (the module's import alone must not)
    code = ('import sys, importlib.util; spec = importlib.util.spec_from_file_location("b", %r); m = importlib.util.module_from_spec(spec); '
            'spec.loader.exec_module(m); m.count_gpus_without_hip(); print("torch" in sys.modules)' % os.path.join(root, 'bench.py'))
    out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, timeout=120).stdout.decode().strip()
    assert out == 'False', out


def test_launcher_ends_the_other_ranks_when_one_fails(tmp_path):
    """a rank > 0 that dies leaves rank 0 waiting in a collective; the launcher polls every child, kills the rest and returns the
    failing code instead of hanging.  Stand-in children: bench.py's own launch_ranks run on a script whose rank 1 exits 7 and whose rank 0
    would sleep for an hour."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'bench.py')).read()
    child = tmp_path / 'bench.py'
    # the launcher functions verbatim + a main that plays the ranks
    head = src[:src.index('def main():')]
    child.write_text(head + """
def main():
    if 'RANK' not in os.environ:
        sys.exit(launch_ranks(2, []))
    if os.environ['RANK'] == '1':
        sys.exit(7)
    time.sleep(3600)


if __name__ == '__main__':
    main()
""")
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    env['RUMPY_BENCH_ONE_DEVICE'] = '1'
    t0 = time.time()
    p = subprocess.run([sys.executable, str(child)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 7 and time.time() - t0 < 60 and b'rank 1 exited with code 7' in p.stderr
    env['RUMPY_BENCH_TIMEOUT'] = '2'
    child.write_text(child.read_text().replace("sys.exit(7)", "time.sleep(3600)"))
    p = subprocess.run([sys.executable, str(child)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 124


def test_ablation_patch_applies_to_its_base():
    """tests/tools/patches/abl_r03.patch (the *_ABL timing branches of rounds 1-3) is applied by tests/tools/build_abl.sh to the kernel sources of
    the commit it was cut from (ABL_BASE in that script): patch and base must keep matching (the tool had rotted unnoticed)."""
    import re
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir(os.path.join(root, '.git')) or shutil.which('git') is None:
        pytest.skip('no git history here (a snapshot of the tree): build_abl.sh runs where the history is')
    base = re.search(r'^ABL_BASE=(\w+)', open(os.path.join(root, 'tests', 'tools', 'build_abl.sh')).read(), re.M).group(1)
    with tempfile.TemporaryDirectory() as d:
        ar = subprocess.run(['git', '-C', root, 'archive', base, 'rumpy_amd/csrc', 'include'], stdout=subprocess.PIPE, check=True).stdout
        subprocess.run(['tar', '-x', '-C', d], input=ar, check=True)
        subprocess.run(['git', 'init', '-q', d], check=True)
        p = subprocess.run(['git', '-C', d, 'apply', '--check', os.path.join(root, 'tests', 'tools', 'patches', 'abl_r03.patch')],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert p.returncode == 0, p.stdout.decode()[-2000:]


# ------------------------------------------------------------------ round 6: host logic of the watchdog fall-back, the fp8 block policy, the chain geometries
def test_watchdog_codes_name_the_switch_that_applies():
    """codes 0x4ff / 0x500 + b come from the residual-block chain (RUMPY_NO_CHAIN), 0x300 + seq from the RCAB pool exchange (RUMPY_RCAB_FORM / RUMPY_NO_RCAB)"""
    from rumpy_amd.engine import SREngine
    assert 'RUMPY_NO_CHAIN=1' in SREngine.watchdog_text(0x4ff) and 'never published' in SREngine.watchdog_text(0x4ff)
    assert 'RUMPY_NO_CHAIN=1' in SREngine.watchdog_text(0x503) and 'block 3' in SREngine.watchdog_text(0x503)
    t = SREngine.watchdog_text(0x305)
    assert 'RUMPY_RCAB_FORM=lazy' in t and 'RUMPY_NO_RCAB=1' in t and 'RUMPY_NO_CHAIN' not in t


def test_fp8_block_policy_parsing(monkeypatch):
    import types
    from rumpy_amd.engine import SREngine
    me = types.SimpleNamespace(FP8_POLICY_DEFAULT='none')
    pol = lambda: SREngine._fp8_bf16_blocks(me, 200)
    monkeypatch.delenv('RUMPY_FP8_BF16_BLOCKS', raising=False)
    assert pol() == frozenset()
    for spec, want in (('none', set()), ('first:3', {0, 1, 2}), ('last:2', {198, 199}), ('every:50', {0, 50, 100, 150}), ('7,9,500,-1', {7, 9}), ('first:500', set(range(200)))):
        monkeypatch.setenv('RUMPY_FP8_BF16_BLOCKS', spec)
        assert pol() == frozenset(want), spec
    for bad in ('some', 'first:0', 'every:x', 'middle:3', '1;2'):
        monkeypatch.setenv('RUMPY_FP8_BF16_BLOCKS', bad)
        with pytest.raises(RuntimeError, match='RUMPY_FP8_BF16_BLOCKS'):
            pol()


def test_chain_geometries_by_image_width():
    """rumpy_res_chain_strips (host-side, no GPU): 6-row strips up to 48 columns, 4-row strips up to 64 (the reference's shipped crops: 16 x 64 x 64 = 256 strips), 0 beyond;
    the work buffer is sized for the geometry with the more strips"""
    lib = _lib_or_skip().lib()
    assert lib.rumpy_res_chain_strips(32, 48, 48) == 256 and lib.rumpy_res_chain_strips(32, 48, 9) == 256 and lib.rumpy_res_chain_strips(1, 5, 48) == 1
    assert lib.rumpy_res_chain_strips(16, 64, 64) == 256 and lib.rumpy_res_chain_strips(16, 64, 49) == 256 and lib.rumpy_res_chain_strips(3, 13, 60) == 12
    assert lib.rumpy_res_chain_strips(1, 64, 65) == 0 and lib.rumpy_res_chain_strips(1, 64, 0) == 0
    assert lib.rumpy_res_chain_work_bytes(16, 64) >= (32 + 256 + 2 * 256 * 32) * 4      # placement words + one 128-byte flag line per (strip, row half)


def test_weight_gradient_share_cut_covers_balances_and_aligns():
    """rumpy_amd.engine.cut_wgrad_shares (round 6): every unit's ranges cover its tiles once; the most loaded share (tiles + 4 tiles per job) is within 3 % of the
    mean - the launch ends with it; the four output-channel tiles of an upsampler conv get the same ranges on four shares of one XCD (share % 8), before any
    other job; the number of jobs does not grow."""
    from rumpy_amd.engine import cut_wgrad_shares
    JC = 4.0

    def run(units, nsh, jc=JC):
        out = cut_wgrad_shares(units, nsh, jc)
        cost, jobs = [0.0] * nsh, 0
        for (nt, key, q), rs in zip(units, out):
            assert rs[0][0] == 0 and rs[-1][1] == nt and all(a[1] == b[0] for a, b in zip(rs[:-1], rs[1:])) and all(r[1] > r[0] for r in rs)
            for t, t1, k, prio in rs:
                assert 0 <= k < nsh
                cost[k] += t1 - t + JC
                jobs += 1
        return out, cost, jobs

    for plain, up1, up2 in [((576, 33), 576, 2304), ((288, 403), 288, 1152), ((576, 9), 576, 2304), ((576, 3), 576, 2304), ((18, 9), 18, 72)]:
        units = [(plain[0], None, 0)] * plain[1] + [(up1, 'a', q) for q in range(4)] + [(up2, 'b', q) for q in range(4)]
        out, cost, jobs = run(units, 256)
        _, cost0, jobs0 = run([(nt, None, q) for nt, key, q in units], 256, 0.0)       # the cut of rounds 3-5: one sequence, equal tile counts
        mean = sum(cost) / 256
        assert max(cost) <= (1.03 * mean + JC if mean > 40 else mean + 2 * JC + 1), (max(cost), mean)      # (a few tiles per share: no job under 4 tiles)
        assert max(cost) <= max(cost0) and (jobs <= jobs0 + 8 or mean <= 40), (max(cost), max(cost0), jobs, jobs0)
        for key in ('a', 'b'):
            quad = [rs for (nt, k, q), rs in zip(units, out) if k == key]
            for r0, r1, r2, r3 in zip(*quad):
                assert r0[:2] == r1[:2] == r2[:2] == r3[:2] and all(r[3] == 0 for r in (r0, r1, r2, r3))
                assert len(set(r[2] % 8 for r in (r0, r1, r2, r3))) == 1 and len(set(r[2] for r in (r0, r1, r2, r3))) == 4
        for (nt, k, q), rs in zip(units, out):
            if k is None:
                assert all(r[3] == 1 for r in rs)
    # share counts a quad layout does not fit, a conv with other than four output-channel tiles: the plain cut
    units = [(18, None, 0)] * 9 + [(18, 'a', q) for q in range(4)] + [(72, 'b', q) for q in range(3)]
    out, cost, jobs = run(units, 7)
    assert max(cost) <= 1.05 * sum(cost) / 7 + JC and all(r[3] == 1 for rs in out for r in rs)
    out, cost, jobs = run([(2304, 'b', q) for q in range(4)], 256)      # a launch of aligned work only (the upper group of a two-phase plan): all 64 quads
    assert max(cost) == 36 + JC and min(cost) == 36 + JC
    out, cost, jobs = run([(5, None, 0)], 256)       # fewer tiles than shares: a tile per share beats one share with five
    assert jobs == 5 and max(cost) == 1 + JC

"""-m gpu parity of conv_chain.hip (round 5): a run of residual blocks (the EDSR body, forward or data-gradient direction) as ONE persistent launch -
strips claimed per XCD, halo rows handed over through the XCD's L2 (sc0 stores / sc1 loads) or, where a strip had to be claimed from another XCD,
through the memory side.  Reference: rumpy/SISR/models/advanced/common.py ResBlock, architectures.py:198-241 (EDSR.forward).

The chain must be BITWISE the per-block launches (rumpy_conv_block, itself checked against torch fp32 in tests/test_kernels_gpu.py): in every hand-off
form (XCD-local, memory side forced, claim bookkeeping under a faked heavy oversubscription), next to a foreign kernel that really holds CUs, and as the
engine uses it (EDSR training and evaluation plans against RUMPY_NO_CHAIN=1)."""
import numpy as np
import pytest
import torch

from gpu_utils import BF16, DEV, PackedConv, stream, to_dev_bytes
from oracle import sr_oracle as O
from rumpy_amd import _lib as L
from tests.test_network_gpu import _handler, _pair

pytestmark = pytest.mark.gpu


def _chain_case(N, H, W, nblk, backward, fmt, hooks, seed, disturb=False, edge=False, entry='rumpy_res_chain', call=L.call):
    """edge: with the single conv at the chain's outer end (rumpy_res_chain_args.edge_*) - forward: rumpy_conv3x3 (+ bias + residual) behind the last block;
    backward: rumpy_conv3x3 with the data-gradient filter in front of the first, whose output tensor the chain launch then WRITES"""
    gen = np.random.default_rng(seed)
    mk = lambda: PackedConv(torch.from_numpy(gen.uniform(-0.05, 0.05, (64, 64, 3, 3)).astype(np.float32)), torch.from_numpy(gen.uniform(-0.1, 0.1, 64).astype(np.float32)))
    convs = [(mk(), mk()) for _ in range(nblk)]
    pe = mk() if edge else None
    dt = torch.float16 if fmt else BF16
    rnd = lambda: torch.from_numpy(gen.standard_normal((N, H, W, 64)).astype(np.float32)).to(DEV).to(dt)
    x0, extra, eres = rnd(), rnd(), rnd()
    masks = [torch.from_numpy(gen.integers(0, 256, (N, H, W, 8), dtype=np.uint8)).to(DEV) for _ in range(nblk)]
    if fmt:
        items = []
        for pc in [c for pair in convs for c in pair] + ([pe] if edge else []):
            pc.w_h = torch.zeros(64 * 64 * 9, dtype=torch.float16, device=DEV)
            items.append(L.PackItem(w=pc.w.data_ptr(), b=pc.b.data_ptr(), w_fwd=pc.w_h.data_ptr(), w_dgrad=None, b_packed=None, cout=64, cin=64, kind=0, shuffle=0, fmt=L.FMT_F16))
        tab = to_dev_bytes((L.PackItem * len(items))(*items))
        L.check(L.lib().rumpy_pack_weights(tab.data_ptr(), len(items), stream()), 'pack')
    outs = {}
    for form in ('blocks', 'chain'):
        ts = [torch.full((N, H, W, 64), float('nan'), dtype=dt, device=DEV) for _ in range(nblk)] if (backward or not fmt) else [None] * nblk
        ys = [torch.full((N, H, W, 64), float('nan'), dtype=dt, device=DEV) for _ in range(nblk)]
        mbs = masks if backward else [torch.zeros(N, H, W, 8, dtype=torch.uint8, device=DEV) for _ in range(nblk)]
        eo = torch.full((N, H, W, 64), float('nan'), dtype=dt, device=DEV)      # the edge conv's output (backward: = the first block's input)
        first_in = eo if (edge and backward) else x0
        recs = []
        for b, (pa, pb) in enumerate(convs):
            x = first_in if b == 0 else ys[b - 1]
            if backward:
                recs.append(dict(x=x.data_ptr(), w1=pb.w_dgrad.data_ptr(), b1=None, w2=pa.w_dgrad.data_ptr(), b2=None, res2=extra.data_ptr() if b == nblk - 1 else None,
                                 t=ts[b].data_ptr(), out=ys[b].data_ptr(), maskbits=mbs[b].data_ptr(), scale1=0.1, scale2=1.0))
            else:
                w1, w2 = (pa.w_h, pb.w_h) if fmt else (pa.w_fwd, pb.w_fwd)
                recs.append(dict(x=x.data_ptr(), w1=w1.data_ptr(), b1=pa.b_packed.data_ptr(), w2=w2.data_ptr(), b2=pb.b_packed.data_ptr(), res2=None,
                                 t=ts[b].data_ptr() if ts[b] is not None else None, out=ys[b].data_ptr(), maskbits=None if fmt else mbs[b].data_ptr(), scale1=1.0, scale2=0.1))
        ekw = {}
        if edge and backward:
            ekw = dict(edge_w=pe.w_dgrad.data_ptr(), edge_x=x0.data_ptr())
        elif edge:
            ekw = dict(edge_w=(pe.w_h if fmt else pe.w_fwd).data_ptr(), edge_b=pe.b_packed.data_ptr(), edge_res=eres.data_ptr(), edge_out=eo.data_ptr())
        if form == 'blocks':
            if edge and backward:
                L.call('rumpy_conv3x3', L.ConvArgs(x=x0.data_ptr(), w=pe.w_dgrad.data_ptr(), bias=None, out=eo.data_ptr(), N=N, H=H, W=W, cin_chunks=1, cout_tiles=1,
                                                   in_mode=0, out_mode=0, relu=0, scale=1.0, grid_x=0, fmt=fmt), stream())
            for r in recs:
                L.call('rumpy_conv_block', L.BlockArgs(N=N, H=H, W=W, relu1=0 if backward else 1, fmt=fmt, **r), stream())
            if edge and not backward:
                L.call('rumpy_conv3x3', L.ConvArgs(x=ys[-1].data_ptr(), w=(pe.w_h if fmt else pe.w_fwd).data_ptr(), bias=pe.b_packed.data_ptr(), out=eo.data_ptr(),
                                                   res1=eres.data_ptr(), N=N, H=H, W=W, cin_chunks=1, cout_tiles=1, in_mode=0, out_mode=0, relu=0, scale=1.0, grid_x=0,
                                                   fmt=fmt), stream())
        else:
            tab = to_dev_bytes((L.ResChainBlock * nblk)(*[L.ResChainBlock(**r) for r in recs]))
            work = torch.zeros(int(L.lib().rumpy_res_chain_work_bytes(N, H)), dtype=torch.uint8, device=DEV)
            status = torch.zeros(1, dtype=torch.int32, device=DEV)
            a = L.ResChainArgs(blocks=tab.data_ptr(), nblocks=nblk, N=N, H=H, W=W, backward=1 if backward else 0, fmt=fmt, work=work.data_ptr(),
                               work_bytes=work.numel(), status=status.data_ptr(), **hooks, **ekw)
            side = torch.cuda.Stream()
            for rep in range(3):           # (launch epochs: the same work buffer serves launch after launch)
                if disturb:
                    L.check(L.lib().rumpy_debug_occupy(160, 20000.0, side.cuda_stream), 'occupy')
                call(entry, a, stream())
            torch.cuda.synchronize()
            assert int(status.item()) == 0
        torch.cuda.synchronize()
        outs[form] = [t for t in ts if t is not None] + ys + ([] if backward or fmt else mbs) + ([eo] if edge else [])
    for i, (p, q) in enumerate(zip(outs['blocks'], outs['chain'])):
        assert torch.isfinite(p.float()).all() or p.dtype == torch.uint8
        assert torch.equal(p.view(torch.uint8), q.view(torch.uint8)), (i, hooks)


HOOKS = [dict(fake_xcc=0, force_sc1=0), dict(fake_xcc=0, force_sc1=1), dict(fake_xcc=3, force_sc1=1), dict(fake_xcc=1, force_sc1=1), dict(fake_xcc=16, force_sc1=1)]


@pytest.mark.parametrize('hooks', HOOKS)
@pytest.mark.parametrize('N,H,W,nblk,backward,fmt', [(32, 48, 48, 6, 0, 0), (32, 48, 48, 6, 1, 0), (5, 20, 37, 3, 0, 0), (5, 20, 37, 3, 1, 0), (3, 13, 48, 4, 0, 1),
                                                     (1, 5, 9, 2, 0, 0), (7, 31, 24, 5, 1, 0)])
def test_chain_is_bitwise_the_per_block_launches(N, H, W, nblk, backward, fmt, hooks):
    _chain_case(N, H, W, nblk, backward, fmt, hooks, 900 + N + H)


@pytest.mark.parametrize('hooks', HOOKS[:3])
@pytest.mark.parametrize('N,H,W,nblk,backward,fmt', [(32, 48, 48, 6, 0, 0), (32, 48, 48, 6, 1, 0), (5, 20, 37, 3, 0, 0), (5, 20, 37, 3, 1, 0), (3, 13, 48, 4, 0, 1),
                                                     (1, 5, 9, 2, 0, 0), (7, 31, 24, 2, 1, 0), (2, 6, 16, 2, 0, 0), (2, 6, 16, 2, 1, 0)])
def test_chain_with_the_conv_at_its_outer_end_is_bitwise_the_separate_launches(N, H, W, nblk, backward, fmt, hooks):
    """EDSR's body-end conv (+ bias + global skip) behind the last block of the forward chain, its data gradient in front of the first block of the backward
    chain - inside the chain launch (rumpy_res_chain_args.edge_*), against rumpy_conv3x3 in front of / behind per-block launches: every buffer bit for bit"""
    _chain_case(N, H, W, nblk, backward, fmt, hooks, 940 + N + H, edge=True)


# ---- round 6: strips of 4 rows x 64 columns (48 < W <= 64: the reference's shipped 64-pixel crops) - the same kernel, geometry G4; the per-block launches it is
# compared with cut such an image into two column tiles of 32 ----
@pytest.mark.parametrize('hooks', HOOKS[:3])
@pytest.mark.parametrize('N,H,W,nblk,backward,fmt', [(16, 64, 64, 6, 0, 0), (16, 64, 64, 6, 1, 0), (5, 20, 53, 3, 0, 0), (5, 20, 53, 3, 1, 0), (3, 13, 64, 4, 0, 1),
                                                     (1, 5, 49, 2, 0, 0), (7, 30, 60, 5, 1, 0), (2, 4, 64, 2, 0, 0), (2, 4, 64, 2, 1, 0), (9, 23, 57, 16, 0, 0)])
def test_chain_on_four_row_strips_of_64_columns_is_bitwise_the_per_block_launches(N, H, W, nblk, backward, fmt, hooks):
    _chain_case(N, H, W, nblk, backward, fmt, hooks, 2100 + N + H + W)


def test_chain_on_64_column_strips_next_to_a_foreign_kernel(monkeypatch):
    _chain_case(16, 64, 64, 8, 1, 0, dict(fake_xcc=0, force_sc1=0), 79, disturb=True)


@pytest.mark.parametrize('N,hw', [(16, 64), (4, (40, 56))])
def test_edsr_training_at_the_shipped_crop_size_on_the_chain_equals_the_per_block_launches(N, hw, monkeypatch):
    """EDSR x2, 5 blocks, two training steps on 64-pixel crops (16 crops = 256 strips of 4 x 64) and an evaluation image 60 pixels wide: the chain launches against
    RUMPY_NO_CHAIN=1 (column-tiled per-block launches) - losses, outputs, weights and the evaluation image bit for bit; the body-end conv stays its own launch"""
    kw = dict(scale=2, num_blocks=5, res_scale=0.1)
    res = []
    monkeypatch.setenv('RUMPY_CHAIN_ANY_FILL', '1')        # 40 strips (the 4-crop case) and the 10 of the evaluation image are below the fill the engine asks for
    for no_chain in ('0', '1'):
        monkeypatch.setenv('RUMPY_NO_CHAIN', no_chain)
        h, _ = _pair('edsr', 516, sched=False, **kw)
        losses = []
        for step in range(2):
            x, y = O.synthetic_batch(770 + step, N, lr_hw=hw, scale=2)
            loss, out = h.run_train(x=x, y=y)
            losses.append(float(loss))
        xe, _ = O.synthetic_batch(779, 1, lr_hw=(37, 60), scale=2)
        ev, _, _ = h.run_eval(x=xe)
        H, W = (hw, hw) if isinstance(hw, int) else hw
        tp = h.net.engine.plan_for(N, H, W, True)
        chains = [a for op, a in tp.fwd + tp.bwd if op == 'rumpy_res_chain']
        assert (len(chains) == 2 and all(a.nblocks == 5 and not a.edge_w for a in chains)) == (no_chain == '0')
        assert h.net.engine.exchange_status() == 0
        res.append((losses, out.clone(), ev.clone(), h.net.flat_p.detach().cpu().clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])


@pytest.mark.parametrize('backward', [0, 1])
def test_chain_next_to_a_foreign_kernel_that_holds_cus(backward):
    """256 strips on 256 CUs while another queue holds part of the chip (rumpy_debug_occupy, 160 workgroups x 80 KiB of LDS, 20 ms at a time - round 6: the 48
    workgroups of round 5 disturbed nothing, tests/tools/watchdog_probe.py; 160 leave chain workgroups without a CU for up to 20 ms, far below the watchdog): the
    chain's workgroups arrive late and on whatever XCD has room - strips are claimed, oversubscribed XCDs hand their surplus to others (memory-side
    hand-off there), nothing may time out and every buffer must equal the per-block launches'."""
    _chain_case(32, 48, 48, 8, backward, 0, dict(fake_xcc=0, force_sc1=0), 77, disturb=True)


def test_chain_refuses_what_it_cannot_run():
    work = torch.zeros(int(L.lib().rumpy_res_chain_work_bytes(40, 48)), dtype=torch.uint8, device=DEV)
    st = torch.zeros(1, dtype=torch.int32, device=DEV)
    a = L.ResChainArgs(blocks=work.data_ptr(), nblocks=2, N=40, H=48, W=48, work=work.data_ptr(), work_bytes=work.numel(), status=st.data_ptr())
    assert L.lib().rumpy_res_chain(a, None) == -1 and b'co-resident' in L.lib().rumpy_last_error()      # 320 strips > 256 CUs
    a.N, a.W = 2, 65
    assert L.lib().rumpy_res_chain(a, None) == -1 and b'W <= 64' in L.lib().rumpy_last_error()
    assert L.lib().rumpy_res_chain_strips(16, 64, 64) == 256 and L.lib().rumpy_res_chain_strips(32, 48, 48) == 256 and L.lib().rumpy_res_chain_strips(1, 8, 65) == 0
    a.W, a.fake_xcc = 48, 3
    assert L.lib().rumpy_res_chain(a, None) == -1 and b'force_sc1' in L.lib().rumpy_last_error()
    assert L.lib().rumpy_device_xcds() == 8


@pytest.mark.parametrize('mode', ['local', 'sc1'])
def test_edsr_training_and_evaluation_on_the_chain_equal_the_per_block_launches(mode, monkeypatch):
    """EDSR x4, 6 blocks, three training steps at 32 x 48 x 48 and an evaluation image: with the chain (forward + data gradient, one launch each) and with
    RUMPY_NO_CHAIN=1 - losses, outputs, weights and the evaluation image bit for bit; the plans really differ."""
    kw = dict(scale=4, num_blocks=6, res_scale=0.1)
    res = []
    for no_chain in ('0', '1'):
        monkeypatch.setenv('RUMPY_NO_CHAIN', no_chain)
        monkeypatch.setenv('RUMPY_CHAIN_SC1', '1' if mode == 'sc1' else '0')
        h, _ = _pair('edsr', 511, sched=False, **kw)
        losses = []
        for step in range(3):
            x, y = O.synthetic_batch(690 + step, 32, lr_hw=48, scale=4)
            loss, out = h.run_train(x=x, y=y)
            losses.append(float(loss))
        xe, _ = O.synthetic_batch(699, 1, lr_hw=(40, 44), scale=4)
        ev, _, _ = h.run_eval(x=xe)
        eng = h.net.engine
        tp, ep = eng.plan_for(32, 48, 48, True), eng.plan_for(1, 40, 44, False, eng.eval_fmt)
        for ops in (tp.fwd, tp.bwd, ep.fwd):
            names = [op for op, _ in ops]
            assert (names.count('rumpy_res_chain') == 1 and 'rumpy_conv_block' not in names) == (no_chain == '0'), names
            if no_chain == '0':      # ... and the body-end conv (its data gradient) rides in the chain launch
                assert all(bool(a.edge_w) for op, a in ops if op == 'rumpy_res_chain')
        assert eng.exchange_status() == 0
        res.append((losses, out.clone(), ev.clone(), h.net.flat_p.detach().cpu().clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])


@pytest.mark.parametrize('blocks,N', [(2, 32), (32, 8), (3, 1)])
def test_edsr_chains_of_other_lengths_equal_the_per_block_launches(blocks, N, monkeypatch):
    """the shortest run that chains (2 blocks + the body-end conv), the depth of the reference's shipped config (32), a single image: two training steps and an
    evaluation, chain against RUMPY_NO_CHAIN=1, bit for bit"""
    kw = dict(scale=2, num_blocks=blocks, res_scale=0.1)
    res = []
    for no_chain in ('0', '1'):
        monkeypatch.setenv('RUMPY_NO_CHAIN', no_chain)
        h, _ = _pair('edsr', 512, sched=False, **kw)
        losses = []
        for step in range(2):
            x, y = O.synthetic_batch(730 + step, N, lr_hw=48, scale=2)
            loss, out = h.run_train(x=x, y=y)
            losses.append(float(loss))
        xe, _ = O.synthetic_batch(739, 1, lr_hw=(30, 48), scale=2)
        ev, _, _ = h.run_eval(x=xe)
        tp = h.net.engine.plan_for(N, 48, 48, True)
        chains = [a for op, a in tp.fwd + tp.bwd if op == 'rumpy_res_chain']
        assert (len(chains) == 2 and all(a.nblocks == blocks and a.edge_w for a in chains)) == (no_chain == '0')
        assert h.net.engine.exchange_status() == 0
        res.append((losses, out.clone(), ev.clone(), h.net.flat_p.detach().cpu().clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])


def test_chain_is_deterministic_at_the_headline_shape():
    kw = dict(scale=4, num_blocks=16, res_scale=0.1)
    x, y = O.synthetic_batch(671, 32, lr_hw=48, scale=4)
    h = _handler('edsr', lr=1e-3, **kw)
    h.net.load_state_dict(O.seeded_state_dict(O.build_oracle('edsr', **kw), 826))
    xd, yd = x.cuda(), y.cuda()
    ref = None
    for rep in range(12):
        _, out = h.net.fused_l1_forward_backward(xd, yd)
        torch.cuda.synchronize()
        cur = (out.detach().clone(), h.net.flat_g.detach().clone())
        if ref is None:
            ref = cur
        else:
            assert torch.equal(ref[0], cur[0]) and torch.equal(ref[1].view(torch.int32), cur[1].view(torch.int32)), rep
    h.net.take_early_loss()
    plan = h.net.engine.plan_for(32, 48, 48, True)
    assert [op for op, _ in plan.fwd].count('rumpy_res_chain') == 1 and h.net.engine.exchange_status() == 0


# ---- round 6: a hand-off that times out.  The chain needs all 256 strips co-resident; a foreign kernel that holds CUs for longer than the
# watchdog (0.5 s, chain_common.hpp::CH_TIMEOUT) breaks that; the tests provoke the same time-out through the work buffer's test hook (_stall_strip_zero).  What must happen: the launch gives up within milliseconds of the first time-out, the optimizer launch of the
# step reads the status word on the device and changes NOTHING, the host warns, switches the engine to one launch per block and training goes on.
def _stall_strip_zero(plan, on=True):
    """the test hook of the chain's work buffer (chain_common.hpp::CH_W_STALL): the workgroup of strip 0 publishes its first hand-off 0.7 s late - its neighbour's
    poll runs into the watchdog (CH_TIMEOUT = 0.5 s).  Deterministic; a foreign kernel that takes CUs breaks the co-residency only when the dispatcher places it
    just so (tests/tools/watchdog_probe.py: up to 96 workgroups of 80 KiB LDS disturb nothing, 160 sometimes, 256 keep the whole launch waiting)."""
    plan.chain_work.view(torch.int32)[2] = 1 if on else 0
    torch.cuda.synchronize()


@pytest.mark.parametrize('generic', [False, True])
def test_a_chain_hand_off_that_times_out_skips_the_step_and_falls_back_to_per_block_launches(generic):
    kw = dict(scale=2, num_blocks=4, res_scale=0.1)
    h, _ = _pair('edsr', 513, sched=False, **kw)
    if generic:                        # criterion other than the stock nn.L1Loss: the whole-network autograd node (the path that never read the word before)
        class MyL1(torch.nn.Module):
            def forward(self, a, b):
                return (a - b).abs().mean()
        h.criterion = MyL1()
    x, y = O.synthetic_batch(760, 32, lr_hw=48, scale=2)
    h.run_train(x=x, y=y)
    eng = h.net.engine
    assert eng.use_chain and 'rumpy_res_chain' in [op for op, _ in eng.plan_for(32, 48, 48, True).fwd]
    before = h.net.flat_p.detach().clone()
    m_before = h.optimizer.flat_m.detach().clone()
    _stall_strip_zero(eng.plan_for(32, 48, 48, True))
    with pytest.warns(RuntimeWarning, match='one launch per block'):
        h.run_train(x=x, y=y)
    torch.cuda.synchronize()
    assert torch.equal(before.view(torch.int32), h.net.flat_p.view(torch.int32)), 'the step whose hand-off timed out reached the weights'
    assert torch.equal(m_before.view(torch.int32), h.optimizer.flat_m.view(torch.int32))
    assert not eng.use_chain
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        loss, out = h.run_train(x=x, y=y)
    torch.cuda.synchronize()
    names = [op for op, _ in eng.plan_for(32, 48, 48, True).fwd]
    assert 'rumpy_res_chain' not in names and names.count('rumpy_conv_block') == 4
    assert np.isfinite(float(loss)) and torch.isfinite(out).all() and not torch.equal(before, h.net.flat_p)
    # ... and the weights it arrives at are those of a handler that never ran the chain and took the same two good steps (Adam's step count aside:
    # the skipped step still advanced the host's counter - bias correction of step 3 instead of step 2: compare directions, not bits)
    ref, _ = _pair('edsr', 513, sched=False, **kw)
    if generic:
        ref.criterion = h.criterion
    ref.net._ensure_engine()
    ref.net.engine.use_chain = False
    ref.run_train(x=x, y=y)
    ref.run_train(x=x, y=y)
    torch.cuda.synchronize()
    d, dr = (h.net.flat_p - before).double(), (ref.net.flat_p - before).double()
    assert float((d @ dr) / (d.norm() * dr.norm())) > 0.98


def test_strict_watchdog_raises_with_the_right_switch_named(monkeypatch):
    monkeypatch.setenv('RUMPY_WATCHDOG_STRICT', '1')
    h, _ = _pair('edsr', 514, sched=False, scale=2, num_blocks=3, res_scale=0.1)
    x, y = O.synthetic_batch(761, 32, lr_hw=48, scale=2)
    h.run_train(x=x, y=y)
    before = h.net.flat_p.detach().clone()
    _stall_strip_zero(h.net.engine.plan_for(32, 48, 48, True))
    with pytest.raises(RuntimeError, match='RUMPY_NO_CHAIN=1'):
        h.run_train(x=x, y=y)
    torch.cuda.synchronize()
    assert torch.equal(before.view(torch.int32), h.net.flat_p.view(torch.int32))


def test_an_evaluation_pass_whose_chain_timed_out_is_run_again_with_per_block_launches():
    h = _handler('edsr', eval_mode=True, scale=2, num_blocks=4, res_scale=0.1)
    h.net.load_state_dict(O.seeded_state_dict(O.build_oracle('edsr', scale=2, num_blocks=4, res_scale=0.1), 515))
    x, _ = O.synthetic_batch(762, 32, lr_hw=48, scale=2)
    good, _, _ = h.run_eval(x=x)
    assert 'rumpy_res_chain' in [op for op, _ in h.net.engine.plan_for(32, 48, 48, False, h.net.engine.eval_fmt).fwd]
    _stall_strip_zero(h.net.engine.plan_for(32, 48, 48, False, h.net.engine.eval_fmt))
    with pytest.warns(RuntimeWarning, match='run again'):
        again, _, _ = h.run_eval(x=x)
    assert torch.equal(good, again) and not h.net.engine.use_chain

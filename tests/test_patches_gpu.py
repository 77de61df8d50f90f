"""-m gpu parity tests of the device-side training-patch pipeline: rumpy_patch_gather through
rumpy_amd.sr_tools.device_patches.DevicePatchSource against (a) the golden patches returned by the imported reference
functions (G10) and (b) the CPU oracle, bit for bit."""
import itertools
import os
import random

import numpy as np
import pytest
import torch

from oracle import patch_oracle as PO

pytestmark = pytest.mark.gpu


def _source(lrs, hrs, scale, crop):
    from rumpy_amd.sr_tools.device_patches import DevicePatchSource
    return DevicePatchSource(lrs, hrs, scale, crop, device='cuda:0')


def test_patches_match_the_reference_functions_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g10_patches.npz'))
    crop, scale, seed_img, cases = [int(v) for v in g['meta']]
    sizes = [tuple(int(v) for v in s) for s in g['sizes']]
    lrs, hrs = PO.synthetic_images(seed_img, sizes, scale)
    src = _source(lrs, hrs, scale, crop)
    for case in range(cases):
        random.seed(1000 + case)                       # the generator state the reference functions started from
        lr, hr = src.sample([case % len(sizes)], rng=random)
        assert np.array_equal(lr[0].cpu().numpy(), g['lr_%d' % case]), case
        assert np.array_equal(hr[0].cpu().numpy(), g['hr_%d' % case]), case


def test_batch_matches_oracle_every_flip_combination_and_corner():
    crop, scale = 48, 4
    sizes = [(48, 48), (48, 97), (130, 48), (77, 121), (64, 64)]
    lrs, hrs = PO.synthetic_images(5, sizes, scale)
    src = _source(lrs, hrs, scale, crop)
    indices, params = [], []
    for k, (h, w) in enumerate(sizes):
        for (hf, vf, rt) in itertools.product((False, True), repeat=3):
            ah, aw = (w, h) if rt else (h, w)
            for (y, x) in ((0, 0), (ah - crop, aw - crop), ((ah - crop) // 2, (aw - crop) // 3)):
                indices.append(k)
                params.append((hf, vf, rt, y, x))
    lr, hr = src.gather(indices, params)
    lr, hr = lr.cpu(), hr.cpu()
    for n, (k, (hf, vf, rt, y, x)) in enumerate(zip(indices, params)):
        a_lr = PO.augment(PO.to_tensor_u8(lrs[k]), hf, vf, rt)
        a_hr = PO.augment(PO.to_tensor_u8(hrs[k]), hf, vf, rt)
        assert torch.equal(lr[n], a_lr[:, y:y + crop, x:x + crop]), (n, k, hf, vf, rt, y, x)
        assert torch.equal(hr[n], a_hr[:, y * scale:(y + crop) * scale, x * scale:(x + crop) * scale]), (n, k, hf, vf, rt, y, x)


def test_headline_batch_random_draws_match_oracle_and_feed_the_handler():
    """N = 32 patches of 48x48 (x4) drawn with the reference's generator calls; the oracle consumes the same generator state."""
    crop, scale, N = 48, 4, 32
    sizes = [(96 + 7 * i, 120 - 5 * i) for i in range(6)]
    lrs, hrs = PO.synthetic_images(9, sizes, scale)
    src = _source(lrs, hrs, scale, crop)
    order = [i % len(sizes) for i in range(N)]
    random.seed(4242)
    lr, hr = src.sample(order, rng=random)
    random.seed(4242)
    for n, k in enumerate(order):
        lp, hp, _ = PO.sample_patch(lrs[k], hrs[k], crop, scale, random)
        assert torch.equal(lr[n].cpu(), lp) and torch.equal(hr[n].cpu(), hp), n
    assert lr.shape == (N, 3, crop, crop) and hr.shape == (N, 3, crop * scale, crop * scale) and lr.dtype == torch.float32
    # size-independent properties: every value is k/255 for an integer k in [0, 255]
    hc = hr.cpu()                                                  # (IEEE division on the CPU; torch's GPU div need not be)
    v = (hc * 255.0).round()
    assert float(v.min()) >= 0 and float(v.max()) <= 255 and torch.equal(v / 255.0, hc)
    # the batch is what handler.run_train takes
    from rumpy_amd.shared_framework.models import define_model
    import tempfile
    h = define_model('edsr', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=False, checkpoint_load=False, loss_masking=False,
                     metadata_list=None, scale=4, n_resblocks=2, lr=1e-4)
    loss, out = h.run_train(lr, hr, keep_on_device=True)
    assert np.isfinite(loss) and out.shape == hr.shape


def test_crop_outside_the_image_is_refused():
    lrs, hrs = PO.synthetic_images(1, [(20, 20)], 2)
    src = _source(lrs, hrs, 2, 16)
    with pytest.raises(ValueError):
        src.gather([0], [(False, False, False, 5, 0)])
    src16 = _source(lrs, hrs, 2, 24)
    with pytest.raises(ValueError):
        src16.gather([0], [(False, False, False, 0, 0)])

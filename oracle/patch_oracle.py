"""CPU restatement of the reference's training-patch pipeline.  TEST INFRASTRUCTURE ONLY: imported by tests/
(and tests/golden/make_golden_patches.py); the product path (rumpy_amd/sr_tools/device_patches.py + rumpy_patch_gather) never
imports it.  Parity PINNED: tests/golden/g10_patches.npz holds the outputs of the imported reference functions
(random_flip_rotate, image_patch_selection) for seeded inputs; tests/test_oracle_golden.py checks this file against them bit for bit.

Reference lines (relative to /root/reference):
  * ToTensor in front of every image: rumpy/sr_tools/data_handler.py:472-486 (torchvision.transforms.ToTensor - third party,
    unpinned in requirements.txt, absent from this image; published behaviour for a uint8 HWC ndarray: permute to CHW, convert
    to float32, divide by 255)
  * image_augment_crop: rumpy/sr_tools/data_handler.py:570-610 - augment first, crop second
  * random_flip_rotate: rumpy/image_tools/image_manipulation/image_functions.py:346-362
  * random_patch_selection / image_patch_selection / extract_image_patch: image_functions.py:245-252, 287-329
"""
import numpy as np
import torch


def to_tensor_u8(img_hwc):
    """torchvision ToTensor for a uint8 HWC ndarray."""
    return torch.from_numpy(np.ascontiguousarray(img_hwc.transpose(2, 0, 1))).to(torch.float32).div(255)


def flip_rotate_params(rng, hflip=True, vflip=True, rot=True):
    """image_functions.py:348-350: one rng.random() per ENABLED augmentation (short-circuit `and`), in this order."""
    h = bool(hflip and rng.random() < 0.5)
    v = bool(vflip and rng.random() < 0.5)
    r = bool(rot and rng.random() < 0.5)
    return h, v, r


def augment(img, h, v, r):
    """image_functions.py:352-360 for a CHW tensor."""
    if h:
        img = torch.flip(img, [2])
    if v:
        img = torch.flip(img, [1])
    if r:
        img = torch.transpose(img, 1, 2)
    return img


def random_patch_params(rng, H, W, crop):
    """image_functions.py:287-294: row first, then column, both inclusive randint."""
    y = rng.randint(0, max(0, H - crop))
    x = rng.randint(0, max(0, W - crop))
    return y, x


def sample_patch(lr_u8, hr_u8, crop, scale, rng, hflip=True, vflip=True, rot=True):
    """One image through data_handler.py:570-610 (random_augment on, patch_type 'random', one patch).
    -> (lr patch [C,crop,crop], hr patch [C,crop*scale,crop*scale], (h, v, r, y, x))"""
    lr, hr = to_tensor_u8(lr_u8), to_tensor_u8(hr_u8)
    h, v, r = flip_rotate_params(rng, hflip, vflip, rot)
    lr, hr = augment(lr, h, v, r), augment(hr, h, v, r)
    y, x = random_patch_params(rng, lr.shape[1], lr.shape[2], crop)
    lp = lr[:, y:y + crop, x:x + crop]
    yg, xg = int(y * scale), int(x * scale)
    cs = int(crop * scale)
    hp = hr[:, yg:yg + cs, xg:xg + cs]
    return lp.contiguous(), hp.contiguous(), (h, v, r, y, x)


def synthetic_images(seed, sizes, scale, C=3):
    """uint8 HWC LR / HR image pairs of the given LR (H, W) sizes (HR is NOT a resampled LR: the pipeline never relates them)."""
    gen = np.random.default_rng(seed)
    lrs = [gen.integers(0, 256, (h, w, C), dtype=np.uint8) for (h, w) in sizes]
    hrs = [gen.integers(0, 256, (h * scale, w * scale, C), dtype=np.uint8) for (h, w) in sizes]
    return lrs, hrs

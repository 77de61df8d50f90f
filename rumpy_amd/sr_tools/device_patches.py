"""Device-side training-patch source: the GPU replacement of the per-item work of the reference's
`SuperResImages.__getitem__` for the training configuration of the hot path (random_augment on, patch_type 'random', one patch
per image; rumpy/sr_tools/data_handler.py:570-645).

The uint8 images live in HBM (one flat buffer; DIV2K's 800 training pairs are 10 GB at x4, 288 GB are available), the random
numbers are drawn on the host from the SAME generator calls in the SAME order as the reference
(rumpy/image_tools/image_manipulation/image_functions.py:348-350 one random() per enabled augmentation - hflip, vflip, rot -
then :287-294 randint for the row, randint for the column of the AUGMENTED image), and one kernel launch
(`rumpy_patch_gather`) produces the fp32 NCHW batch `handler.run_train` consumes: crop, flips, transpose and the
torchvision-ToTensor conversion (uint8 -> float / 255) are a single gather.  With a seeded `random` module the batch is
bit-identical to what the reference's DataLoader (num_workers = 0) would deliver for the same image order.

No CPU path: without the HIP library / a GPU construction fails.
"""
import ctypes as C
import random as _random

import numpy as np
import torch

from rumpy_amd import _lib as L
from rumpy_amd.staging import PinnedRing


# numpy mirror of rumpy_patch_item (include/rumpy_amd.h); tests/test_host_cpu.py checks it against the ctypes struct
ITEM_DTYPE = np.dtype([('lr_off', '<i8'), ('hr_off', '<i8'), ('lr_h', '<i4'), ('lr_w', '<i4'), ('hflip', '<i4'), ('vflip', '<i4'),
                       ('rot', '<i4'), ('y', '<i4'), ('x', '<i4'), ('pad_', '<i4')])
ITEM_BYTES = ITEM_DTYPE.itemsize


class DevicePatchSource:
    def __init__(self, lr_images, hr_images, scale, crop, device='cuda:0', hflip=True, vflip=True, rot=True):
        """lr_images / hr_images: lists of uint8 HWC numpy arrays (HR exactly `scale` times the LR size)."""
        if not torch.cuda.is_available():
            raise RuntimeError('rumpy_amd: DevicePatchSource needs an MI355X and the HIP kernel library; there is no CPU path')
        L.lib()
        if len(lr_images) != len(hr_images) or not lr_images:
            raise ValueError('need the same, non-zero number of LR and HR images')
        self.scale, self.crop, self.device = int(scale), int(crop), torch.device(device)
        self.use_hflip, self.use_vflip, self.use_rotation = hflip, vflip, rot
        self.C = int(lr_images[0].shape[2])
        self.meta = []
        chunks, off = [], 0
        for lr, hr in zip(lr_images, hr_images):
            if lr.dtype != np.uint8 or hr.dtype != np.uint8 or lr.ndim != 3 or lr.shape[2] != self.C:
                raise ValueError('images must be uint8 HWC with the same channel count')
            if hr.shape[0] != lr.shape[0] * self.scale or hr.shape[1] != lr.shape[1] * self.scale:
                raise ValueError('HR image must be exactly scale x the LR image')
            lo = off
            chunks.append(np.ascontiguousarray(lr).reshape(-1)); off += lr.size
            ho = off
            chunks.append(np.ascontiguousarray(hr).reshape(-1)); off += hr.size
            self.meta.append((lo, ho, int(lr.shape[0]), int(lr.shape[1])))
        self.images = torch.from_numpy(np.concatenate(chunks)).to(self.device)
        self._meta_np = np.asarray(self.meta, dtype=np.int64)
        self._ring = PinnedRing((256, ITEM_BYTES), torch.uint8)

    def __len__(self):
        return len(self.meta)

    def draw(self, index, rng=_random):
        """The reference's random choices for one image: (hflip, vflip, rot, y, x) - same calls, same order."""
        lo, ho, h, w = self.meta[index]
        hf = bool(self.use_hflip and rng.random() < 0.5)
        vf = bool(self.use_vflip and rng.random() < 0.5)
        rt = bool(self.use_rotation and rng.random() < 0.5)
        ah, aw = (w, h) if rt else (h, w)                      # size of the augmented image
        y = rng.randint(0, max(0, ah - self.crop))
        x = rng.randint(0, max(0, aw - self.crop))
        return hf, vf, rt, y, x

    def gather(self, indices, params):
        """indices: image index per patch; params: (hflip, vflip, rot, y, x) per patch -> (lr [N,C,crop,crop], hr [N,C,s*crop,s*crop])"""
        n = len(indices)
        meta = self._meta_np[np.asarray(indices, dtype=np.int64)]            # [n, 4]: lr_off, hr_off, h, w
        p = np.asarray(params, dtype=np.int64).reshape(n, 5)                # hflip, vflip, rot, y, x
        ah = np.where(p[:, 2] != 0, meta[:, 3], meta[:, 2])
        aw = np.where(p[:, 2] != 0, meta[:, 2], meta[:, 3])
        bad = (p[:, 3] < 0) | (p[:, 4] < 0) | (p[:, 3] + self.crop > ah) | (p[:, 4] + self.crop > aw)
        if bad.any():
            k = int(np.argmax(bad))
            raise ValueError('patch %d: crop %d at (%d, %d) leaves the %dx%d augmented image' % (k, self.crop, p[k, 3], p[k, 4], ah[k], aw[k]))
        # the item table in the C struct's layout (rumpy_patch_item), staged through a small ring of pinned buffers
        slot_i, slot = self._ring.acquire(min_rows=n)
        rec = np.zeros(n, dtype=ITEM_DTYPE)
        rec['lr_off'], rec['hr_off'], rec['lr_h'], rec['lr_w'] = meta[:, 0], meta[:, 1], meta[:, 2], meta[:, 3]
        rec['hflip'], rec['vflip'], rec['rot'], rec['y'], rec['x'] = p[:, 0] != 0, p[:, 1] != 0, p[:, 2] != 0, p[:, 3], p[:, 4]
        host = slot[:n]
        host.numpy().view(ITEM_DTYPE).reshape(n)[:] = rec
        items_dev = host.to(self.device, non_blocking=True)
        self._ring.sent(slot_i)
        hc = self.crop * self.scale
        lr = torch.empty(n, self.C, self.crop, self.crop, dtype=torch.float32, device=self.device)
        hr = torch.empty(n, self.C, hc, hc, dtype=torch.float32, device=self.device)
        a = L.PatchArgs(images=self.images.data_ptr(), items=items_dev.data_ptr(), out_lr=lr.data_ptr(), out_hr=hr.data_ptr(),
                        N=n, C=self.C, crop=self.crop, scale=self.scale)
        L.call('rumpy_patch_gather', a, torch.cuda.current_stream(self.device).cuda_stream)
        lr._items_keepalive = items_dev        # the launch is asynchronous
        return lr, hr

    def sample(self, indices, rng=_random):
        """One training batch for the given image order, random choices drawn like the reference's dataset would."""
        params = [self.draw(i, rng) for i in indices]
        return self.gather(indices, params)

"""Image saving for the evaluation / sample-output paths - mirror of rumpy/sr_tools/visualization.py:31-62 (``safe_image_save``).

Same signature and file results.  A batch that is still on the GPU (what ``run_eval(..., keep_on_device=True)`` /
``SISRInterface.net_run_and_process`` produce) is clipped, scaled, truncated to uint8 and transposed to HWC by one kernel
(``rumpy_to_uint8_hwc``), so a quarter of the bytes cross PCIe; numpy batches take the reference's numpy expression.  The 'ycbcr'
configuration converts on the host with the interface's 'jpg' matrix (image_functions.py:108-121).  Files are written with PIL
(the reference's ``imsave`` = skimage.io, which writes through PIL / imageio with the same uint8 array).
"""
import os

import numpy as np
import torch

from rumpy_amd import _lib as L


def create_dir_if_empty(path):
    if path and not os.path.isdir(path):
        os.makedirs(path, exist_ok=True)


def to_uint8_hwc(images, max_val=1):
    """[N,C,H,W] float (numpy or torch, CPU or GPU) -> numpy uint8 [N,H,W,C] = trunc(clip(im * 255 / max_val, 0, 255)) (visualization.py:53-56)"""
    if isinstance(images, torch.Tensor) and images.is_cuda:
        x = images.detach().float().contiguous()
        n, c, h, w = x.shape
        out = torch.empty(n, h, w, c, dtype=torch.uint8, device=x.device)
        L.check(L.lib().rumpy_to_uint8_hwc(x.data_ptr(), out.data_ptr(), n, c, h, w, float(max_val), torch.cuda.current_stream(x.device).cuda_stream),
                'rumpy_to_uint8_hwc')
        return out.cpu().numpy()
    arr = images.detach().cpu().numpy() if isinstance(images, torch.Tensor) else np.asarray(images)
    return np.clip(arr.transpose(0, 2, 3, 1) * 255 / max_val, 0, 255).astype(np.uint8)


def safe_image_save(images, out_loc, names, config, max_val=1, im_type='jpg'):
    """visualization.py:31-62: save a batch (BxCxHxW, or BxHxWxC numpy) under out_loc/names[i] after conversion to uint8."""
    from PIL import Image
    create_dir_if_empty(out_loc)
    if config == 'ycbcr':
        if im_type != 'jpg':
            raise NotImplementedError("rumpy_amd safe_image_save: only the 'jpg' YCbCr matrix is built (what the SISR interface passes)")
        from rumpy_amd.SISR.models.interface import SISRInterface
        arr = images.detach().cpu().numpy() if isinstance(images, torch.Tensor) else np.asarray(images)
        rgb = SISRInterface.ycbcr_jpg_to_rgb(arr, max_val=max_val)          # image_functions.py:108-121, per image as visualization.py:47-48
        u8 = np.clip(rgb.transpose(0, 2, 3, 1) * 255 / max_val, 0, 255).astype(np.uint8)
    elif isinstance(images, torch.Tensor) or np.asarray(images).shape[1] == 3:
        u8 = to_uint8_hwc(images, max_val)
    else:                                                   # already HWC
        u8 = np.clip(np.asarray(images) * 255 / max_val, 0, 255).astype(np.uint8)
    for index in range(u8.shape[0]):
        output_path = os.path.join(out_loc, names[index])
        create_dir_if_empty(os.path.dirname(output_path))
        im = u8[index]
        Image.fromarray(im[:, :, 0] if im.shape[2] == 1 else im).save(output_path)

"""PSNR as the reference defines "eval PSNR" (rumpy/sr_tools/metrics.py:33-44, 109-121): float32 mse over the Y channel
of the whole batch, max_value = 1, no border shave, 100 when identical."""
import numpy as np


def psnr(img1, img2, max_value=255.0):
    mse = np.mean((np.array(img1, dtype=np.float32) - np.array(img2, dtype=np.float32)) ** 2)
    if mse == 0:
        return 100
    return 20 * np.log10(max_value / (np.sqrt(mse)))


def psnr_from_sse(sse, count, max_value=1.0):
    """Same quantity from a device-side sum of squared errors."""
    if sse == 0:
        return 100
    return 20 * np.log10(max_value / np.sqrt(np.float32(sse) / np.float32(count)))


class Metrics:
    def run_psnr(self, im_a, im_ref, single_values=False, multichannel=False, max_value=1):
        if im_ref is None:
            raise Exception('Need a reference to calculate PSNR.')
        if single_values:
            return [psnr(im_a[i, 0, :, :], im_ref[i, 0, :, :], max_value=max_value) for i in range(im_a.shape[0])]
        if multichannel:
            return psnr(im_a, im_ref, max_value=max_value)
        return psnr(im_a[:, 0, :, :], im_ref[:, 0, :, :], max_value=max_value)

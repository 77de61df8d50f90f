"""PSNR as the reference defines "eval PSNR" (rumpy/sr_tools/metrics.py:33-44, 109-121): float32 mse over the Y channel
of the whole batch, max_value = 1, no border shave, 100 when identical."""
import numpy as np
import torch

from rumpy_amd import _lib as L


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

    def run_ssim(self, im_a, im_ref, single_values=False, multichannel=False, max_value=1):
        """metrics.py:123-149 (skimage structural_similarity with gaussian_weights=True, sigma=1.5, use_sample_covariance=False),
        evaluated on the GPU by rumpy_ssim.  im_a / im_ref: [N,C,H,W] numpy arrays or tensors; same selection of planes and the
        same averaging as the reference (channel 0 only unless multichannel; mean over images unless single_values)."""
        if im_ref is None:
            raise Exception('Need a reference to calculate SSIM.')
        vals = ssim_planes(im_a if multichannel else im_a[:, :1], im_ref if multichannel else im_ref[:, :1], max_value)   # [N, C']
        per_image = vals.mean(axis=1)
        if multichannel:
            return float(per_image.sum() / len(per_image))
        return [float(v) for v in per_image] if single_values else float(per_image.mean())


def ssim_planes(a, b, data_range=1.0, device='cuda:0'):
    """Mean SSIM of every [H,W] plane of two [N,C,H,W] arrays -> numpy [N,C].  No CPU path."""
    if not torch.cuda.is_available():
        raise RuntimeError('rumpy_amd: SSIM runs on the MI355X through the HIP kernel library; there is no CPU path')
    ta = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(device=device, dtype=torch.float32).contiguous()
    tb = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).to(device=device, dtype=torch.float32).contiguous()
    if ta.shape != tb.shape or ta.dim() != 4:
        raise ValueError('expected two [N,C,H,W] arrays of the same shape')
    n, c, h, w = ta.shape
    lib = L.lib()
    partial = torch.empty(max(1, int(lib.rumpy_ssim_partial_floats(n * c, h, w))), dtype=torch.float32, device=ta.device)
    out = torch.empty(n * c, dtype=torch.float32, device=ta.device)
    L.call('rumpy_ssim', L.SsimArgs(a=ta.data_ptr(), b=tb.data_ptr(), partial=partial.data_ptr(), out=out.data_ptr(), P=n * c, H=h, W=w,
                                    data_range=float(data_range)), torch.cuda.current_stream(ta.device).cuda_stream)
    return out.cpu().numpy().reshape(n, c)

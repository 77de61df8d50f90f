"""CPU restatement of the SSIM the reference's evaluation reports.  TEST INFRASTRUCTURE ONLY.

Reference call site: rumpy/sr_tools/metrics.py:123-149 (Metrics.run_ssim) ->
    skimage.metrics.structural_similarity(a, b, data_range=max_value, gaussian_weights=True, use_sample_covariance=False,
                                          sigma=1.5[, multichannel=True])
scikit-image (>= 0.16.2, requirements.txt:13) is a third-party dependency that is ABSENT from this image, so the algorithm is
restated from its published implementation (skimage/metrics/_structural_similarity.py, Wang et al. 2004 settings): truncate 3.5 ->
radius 5, 11-tap window; local statistics through scipy.ndimage.gaussian_filter (mode 'reflect' - the very function skimage
calls, present here); population covariance; K1 = 0.01, K2 = 0.03; crop (win_size - 1) // 2 pixels; mean; multichannel = mean of
the per-channel values.  PARITY STATUS: pinned to scipy's gaussian_filter only - no skimage output and no reference fixture
exists for this metric (the reference has no test or golden value for it): partial.
"""
import numpy as np
from scipy.ndimage import gaussian_filter


def ssim_plane(a, b, data_range=1.0, sigma=1.5, truncate=3.5):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    r = int(truncate * sigma + 0.5)
    win = 2 * r + 1
    f = lambda z: gaussian_filter(z, sigma=sigma, truncate=truncate, mode='reflect')
    ux, uy = f(a), f(b)
    uxx, uyy, uxy = f(a * a), f(b * b), f(a * b)
    vx, vy, vxy = uxx - ux * ux, uyy - uy * uy, uxy - ux * uy      # cov_norm = 1 (use_sample_covariance=False)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return float(S[pad:-pad, pad:-pad].mean())


def run_ssim(im_a, im_ref, single_values=False, multichannel=False, max_value=1):
    """Metrics.run_ssim (metrics.py:123-149) for [N,C,H,W] arrays."""
    if im_ref is None:
        raise Exception('Need a reference to calculate SSIM.')
    if multichannel:
        vals = [np.mean([ssim_plane(im_a[i, c], im_ref[i, c], max_value) for c in range(im_a.shape[1])]) for i in range(im_a.shape[0])]
        return sum(vals) / len(vals)
    per = [ssim_plane(im_a[i, 0], im_ref[i, 0], max_value) for i in range(im_a.shape[0])]
    return per if single_values else float(np.mean(per))

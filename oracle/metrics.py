"""Oracle (test infrastructure): final-report image quality metrics on the host, restating what the
reference obtains from scikit-image 0.18 at dvp_linear_inv_2_stage_ADMM_tensor_online.py:316-321 / :542-547:
`peak_signal_noise_ratio(X, x, data_range=1.)` and `structural_similarity(X, x, data_range=1.)`
(7x7 uniform window, sample covariance, K1=0.01, K2=0.03, border of 3 px cropped) per frame."""
import numpy as np
from scipy.ndimage import uniform_filter


def psnr(ref, img, data_range=1.0):
    ft = np.result_type(ref.dtype, img.dtype, np.float32)
    err = np.mean((ref.astype(ft) - img.astype(ft)) ** 2, dtype=np.float64)
    return float(10 * np.log10((data_range ** 2) / err))


def ssim(ref, img, data_range=1.0, win=7, K1=0.01, K2=0.03):
    a = ref.astype(np.float64)
    b = img.astype(np.float64)
    npx = win ** a.ndim
    cov_norm = npx / (npx - 1)
    ux, uy = uniform_filter(a, size=win), uniform_filter(b, size=win)
    uxx, uyy, uxy = uniform_filter(a * a, size=win), uniform_filter(b * b, size=win), uniform_filter(a * b, size=win)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return float(S[tuple(slice(pad, -pad) for _ in range(a.ndim))].mean())


def psnr_frames(orig, recon):
    return [psnr(orig[:, :, t], recon[:, :, t]) for t in range(orig.shape[2])]


def ssim_frames(orig, recon):
    return [ssim(orig[:, :, t], recon[:, :, t]) for t in range(orig.shape[2])]

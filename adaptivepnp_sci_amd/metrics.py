"""Final-report image quality metrics on the device (csrc/metrics.hip): per-frame PSNR and SSIM of the mosaic cube,
what the reference obtains from scikit-image 0.18 at dvp_linear_inv_2_stage_ADMM_tensor_online.py:316-321 / :542-547
(`peak_signal_noise_ratio(X, x, data_range=1.)`, `structural_similarity(X, x, data_range=1.)`: 7x7 uniform window,
sample covariance, K1 = 0.01, K2 = 0.03, 3-pixel border cropped)."""
import numpy as np
import torch

from . import ops


def frame_metrics(ref_state, img_state, data_range=1.0, win=7):
    """plane-major states [B][4][M][N] of the ground truth and the reconstruction -> (psnr [B], ssim [B]) lists."""
    B, _, M, N = ref_state.shape
    H, W = 2 * M, 2 * N
    if min(H, W) < win:
        raise ValueError('win_size exceeds image extent')
    part = torch.empty(B, ops.frame_metrics_nblocks(M, N, B, win), 2, dtype=torch.float64, device=ref_state.device)
    ops.frame_metrics(ref_state, img_state, part, win, data_range)
    s = ops.to_host(part).sum(1)                       # [B][blocks][2] fp64 partials: the last reduction on the host
    mse = s[:, 0] / (H * W)
    psnr = [float(10 * np.log10((data_range ** 2) / e)) for e in mse]
    ssim = [float(v / ((H - win + 1) * (W - win + 1))) for v in s[:, 1]]
    return psnr, ssim

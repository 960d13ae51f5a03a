"""Oracle (test infrastructure): the two PnP-ADMM solver loops of the reference restated on
PyTorch-CPU, returning every iterate so that the HIP path can be compared step by step.

two_stage_admm  <- dvp_linear_inv_2_stage_ADMM_tensor_online.py:40-324  (twoStageAdmm_denoise_bayer)
one_stage_admm  <- dvp_linear_inv_2_stage_ADMM_tensor_online.py:326-552 (admm_denoise_bayer_demosaic_pre)

one_stage_admm_gray -- PARITY UNPINNED: the reference has no grayscale (non-Bayer) solver; this is
                  the one-stage loop above with the Bayer split and the demosaic removed, restated here only so that the
                  HIP gray mode has a CPU statement of the same arithmetic to be compared with.

Semantics that are easy to lose and are kept on purpose (all verified bit-for-bit against the
imported reference by tools/make_golden.py):
  * at entry x, theta and the start point are ONE tensor (:87-89 / :375-377).  On the CNN branches
    the in-place writes theta[..., c] = out[...] (:206-209, :256-259) therefore also overwrite x
    in the first iteration, so the first dual update is b += theta_unclipped - theta_clipped;
    `torch.clip` (:265) ends the aliasing.  The TV branches rebind theta first and are unaffected.
  * two-stage: p = theta - b/rho, denominator alpha*rho + Phi_sum, b += x - theta, reports theta;
    one-stage: v = theta + b, denominator Phi_sum + gamma, b -= x - theta, reports x.
  * the finetune gate uses the global iteration counter k, not the per-stage one (:200, :247).
"""
import numpy as np
import torch

from . import sci_ops as ops
from .denoisers import ddnet_pass, fastdvdnet_pass, ffdnet_pass
from .malvar import malvar_demosaic_cube
from .tv_chambolle import tv_chambolle_multichannel


def psnr_np(ref, img, data_range=1.0):
    """skimage.metrics.peak_signal_noise_ratio for float inputs (skimage 0.18 simple_metrics.py)."""
    ft = np.result_type(ref.dtype, img.dtype, np.float32)
    err = np.mean((ref.astype(ft) - img.astype(ft)) ** 2, dtype=np.float64)
    return 10 * np.log10((data_range ** 2) / err)


def _as_list(sigma, iter_max):
    if not isinstance(sigma, list):
        sigma = [sigma]
    if not isinstance(iter_max, list):
        iter_max = [iter_max] * len(sigma)
    return sigma, iter_max


def _tv(planes_in):
    M, N, B, _ = planes_in.shape
    v = planes_in.reshape(M, N, B * 4).numpy()
    return torch.from_numpy(tv_chambolle_multichannel(v, 0.1, n_iter_max=5)).reshape(M, N, B, 4)


def two_stage_admm(y_bayer, Phi_bayer, denoiser='tv', iter_max=50, sigma=None, x0_bayer=None,
                   X_orig=None, model_denoise=None, lr=1e-6, inital_iter=1, interval_iter=5,
                   update=False, update_per_iter=1, update_times=-1, finetune_trace=None,
                   denoiser_io=None, close_form_demosaic=False, model_demosaic=None):
    """Returns dict(theta_iterates=[(H,W,B) np], psnr_all, x_bayer, rgb (CNN branches), model)."""
    y_bayer = torch.as_tensor(y_bayer)
    Phi_bayer = torch.as_tensor(Phi_bayer)
    sigma, iter_max = _as_list(sigma, iter_max)
    yall, Phiall, Phi_sum, x0 = ops.setup_planes(y_bayer, Phi_bayer,
                                                 None if x0_bayer is None else torch.as_tensor(x0_bayer))
    H, W, nB = Phi_bayer.shape
    x = x0
    theta = x0                      # alias, see module docstring
    b = torch.zeros_like(x0)
    w = torch.zeros(H, W, 3, nB)
    alpha = 0.01 if denoiser == 'tv' else 1
    rho = 0.55 if denoiser == 'fastdvd_color' else 1
    tau = 100
    if close_form_demosaic:
        # closed-form x_rgb update for k > 0 (reference :112-118): tau = 10, rho = 0.55 for BOTH denoisers
        tau = 10
        rho = 0.55
        R_m, G_m, B_m = ops.cfa_masks((H, W))
        bayer_mask = torch.cat([R_m.unsqueeze(2), G_m.unsqueeze(2), B_m.unsqueeze(2)], dim=2)
        inv_3ch = torch.repeat_interleave((rho * bayer_mask + tau).unsqueeze(3), nB, dim=3)
    k = 0
    n_updates = 0
    iterates, psnr_all = [], []
    rgb_out = None
    for stage, nsig in enumerate(sigma):
        for _ in range(iter_max[stage]):
            ops.project_two_stage(theta, b, Phiall, yall, Phi_sum, rho, alpha, out=x)
            if denoiser == 'tv':
                theta = _tv(x + (1 / rho) * b)
                is_tv = True
            elif denoiser in ('ffdnet_color', 'fastdvd_color'):
                is_tv = False
                if close_form_demosaic and k > 0:
                    # reference :175-182 (FFDNet branch, clipped) / :224-230 (FastDVDnet branch, NOT clipped)
                    x_rgb = (rho * ops.four_to_three_channel(x) + ops.four_to_three_channel(b) + tau * rgb_out + w) / inv_3ch
                    if denoiser == 'ffdnet_color':
                        x_rgb = x_rgb.clip(0, 1)
                elif model_demosaic is None:
                    mosaic = ops.bayer_merge(x + (1 / rho) * b)
                    x_rgb = malvar_demosaic_cube(mosaic)
                else:
                    # deep demosaicking (reference :192-194 / :242-244)
                    x_rgb = ddnet_pass(ops.one_to_three_channel(ops.bayer_merge(x + (1 / rho) * b)), model_demosaic)
                x_rgb_w = x_rgb - (1 / tau) * w
                gate = update and k > inital_iter and k % interval_iter == 0
                if denoiser == 'ffdnet_color':
                    if gate:
                        rgb_out, model_denoise = ffdnet_pass(x_rgb_w, yall, Phiall, nsig, model_denoise, lr,
                                                             True, update_per_iter, trace=finetune_trace)
                    else:
                        rgb_out = ffdnet_pass(x_rgb_w, yall, Phiall, nsig, model_denoise, lr)
                else:
                    if gate and (n_updates < update_times or update_times < 0):
                        rgb_out, model_denoise = fastdvdnet_pass(x_rgb_w, nsig, yall, Phiall, model_denoise, lr,
                                                                 True, update_per_iter, trace=finetune_trace)
                        n_updates += 1
                    else:
                        rgb_out = fastdvdnet_pass(x_rgb_w, nsig, yall, Phiall, model_denoise, lr)
                rgb_out = rgb_out.detach()
                if denoiser_io is not None:
                    denoiser_io.append((x_rgb_w.clone(), rgb_out.clone()))
                # in-place on theta: while theta is x (first iteration) this rewrites x too
                theta[..., 0] = rgb_out[0::2, 0::2, 0, :]
                theta[..., 1] = rgb_out[0::2, 1::2, 1, :]
                theta[..., 2] = rgb_out[1::2, 0::2, 1, :]
                theta[..., 3] = rgb_out[1::2, 1::2, 2, :]
            else:
                raise ValueError('Unsupported denoiser {}!'.format(denoiser))
            theta = torch.clip(theta, 0, 1)
            b = b + (x - theta)
            if not is_tv:
                w = w + (x_rgb - rgb_out)
            it_mosaic = ops.bayer_merge(theta).numpy()
            iterates.append(it_mosaic)
            if X_orig is not None:
                psnr_all.append(psnr_np(X_orig, it_mosaic))
            k += 1
    x_bayer = ops.bayer_merge(theta).numpy()
    return dict(theta_iterates=iterates, psnr_all=psnr_all, x_bayer=x_bayer,
                rgb=None if rgb_out is None else rgb_out.numpy(), model=model_denoise)


def one_stage_admm(y_bayer, Phi_bayer, _lambda=1, gamma=0.01, denoiser='tv', iter_max=50, sigma=None,
                   x0_bayer=None, X_orig=None, model=None, lr=1e-6, inital_iter=1, interval_iter=5,
                   update=False, update_per_iter=1):
    """Returns dict(x_iterates=[(H,W,B) np], psnr_all, x_bayer, rgb, model).  Reports x, not theta
    (reference :509, :540)."""
    y_bayer = torch.as_tensor(y_bayer)
    Phi_bayer = torch.as_tensor(Phi_bayer)
    sigma, iter_max = _as_list(sigma, iter_max)
    yall, Phiall, Phi_sum, x0 = ops.setup_planes(y_bayer, Phi_bayer,
                                                 None if x0_bayer is None else torch.as_tensor(x0_bayer))
    x = x0
    theta = x0
    b = torch.zeros_like(x0)
    k = 0
    iterates, psnr_all = [], []
    rgb_out = None
    for stage, nsig in enumerate(sigma):
        for _ in range(iter_max[stage]):
            ops.project_one_stage(theta, b, Phiall, yall, Phi_sum, _lambda, gamma, out=x)
            if denoiser == 'tv':
                theta = _tv(x - b)
            elif denoiser in ('ffdnet_color', 'fastdvd_color'):
                x_rgb = malvar_demosaic_cube(ops.bayer_merge(x - b))
                if denoiser == 'ffdnet_color':
                    if update and k > inital_iter and k % interval_iter == 0:
                        rgb_out, model = ffdnet_pass(x_rgb, yall, Phiall, nsig, model, lr, True, update_per_iter)
                    else:
                        rgb_out = ffdnet_pass(x_rgb, yall, Phiall, nsig, model, lr, False)
                else:
                    rgb_out = fastdvdnet_pass(x_rgb, nsig, yall, Phiall, model, lr)
                rgb_out = rgb_out.detach()
                theta[..., 0] = rgb_out[0::2, 0::2, 0, :]
                theta[..., 1] = rgb_out[0::2, 1::2, 1, :]
                theta[..., 2] = rgb_out[1::2, 0::2, 1, :]
                theta[..., 3] = rgb_out[1::2, 1::2, 2, :]
            else:
                raise ValueError('Unsupported denoiser {}!'.format(denoiser))
            theta = torch.clip(theta, 0, 1)
            b = b - (x - theta)
            it_mosaic = ops.bayer_merge(x).numpy()
            iterates.append(it_mosaic)
            if X_orig is not None:
                psnr_all.append(psnr_np(X_orig, it_mosaic))
            k += 1
    x_bayer = ops.bayer_merge(x).numpy()
    return dict(x_iterates=iterates, psnr_all=psnr_all, x_bayer=x_bayer,
                rgb=None if rgb_out is None else rgb_out.numpy(), model=model)


def one_stage_admm_gray(y, Phi, _lambda=1, gamma=0.01, denoiser='tv_gray', iter_max=50, sigma=None, x0=None, X_orig=None,
                        model=None, Phi_sum=None):
    """PARITY UNPINNED (no reference function exists): grayscale PnP-ADMM on a (H,W,B) cube -- the reference's one-stage
    loop (dvp...:385-407, :500-509) without the Bayer split / demosaic:
        x = (theta+b) + lambda * At((y - A(theta+b)) / (Phi_sum + gamma));  theta = clip(D(x - b), 0, 1);  b = b - (x - theta)
    D = skimage-0.18 Chambolle TV on the frames (weight 0.1, 5 iterations, multichannel) or FFDNet-gray per frame.
    Returns dict(x_iterates, psnr_all, x)."""
    y = torch.as_tensor(y)
    Phi = torch.as_tensor(Phi)
    sigma, iter_max = _as_list(sigma, iter_max)
    if Phi_sum is None:
        Phi_sum = torch.sum(Phi, dim=2)
    else:
        Phi_sum = torch.as_tensor(Phi_sum).clone()
    Phi_sum[Phi_sum == 0] = 1
    x = theta = ops.transpose_At(y, Phi) if x0 is None else torch.as_tensor(x0).clone()
    b = torch.zeros_like(x)
    iterates, psnr_all = [], []
    for stage, nsig in enumerate(sigma):
        for _ in range(iter_max[stage]):
            yb = ops.forward_A(theta + b, Phi)
            x = theta + b + _lambda * ops.transpose_At((y - yb) / (Phi_sum + gamma), Phi)
            if denoiser == 'tv_gray':
                theta = torch.from_numpy(tv_chambolle_multichannel((x - b).numpy(), 0.1, n_iter_max=5))
            elif denoiser == 'ffdnet_gray':
                v = x - b
                theta = torch.empty_like(v)
                with torch.no_grad():
                    for t in range(v.shape[2]):
                        frame = v[:, :, t][None, None]
                        theta[:, :, t] = model(frame, torch.full((1, 1, 1, 1), nsig).type_as(frame))[0, 0]
            else:
                raise ValueError('Unsupported denoiser {}!'.format(denoiser))
            theta = torch.clip(theta, 0, 1)
            b = b - (x - theta)
            iterates.append(x.numpy().copy())
            if X_orig is not None:
                psnr_all.append(psnr_np(X_orig, iterates[-1]))
    return dict(x_iterates=iterates, psnr_all=psnr_all, x=x.numpy())

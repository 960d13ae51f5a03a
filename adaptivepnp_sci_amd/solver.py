"""Drop-in equivalents of the reference's two PnP-ADMM solvers, driving the HIP kernels.

twoStageAdmm_denoise_bayer       <- dvp_linear_inv_2_stage_ADMM_tensor_online.py:40-324
admm_denoise_bayer_demosaic_pre  <- dvp_linear_inv_2_stage_ADMM_tensor_online.py:326-552

Same positional/keyword arguments, same return tuples, same log text.  Everything between the
input conversion and the final read-back stays on the GPU in the plane-major layout
(include/scipnp.h): per iteration the host only enqueues kernels on the current HIP stream -- no
device->host copy for TV (the reference goes through NumPy every iteration, :153-160) and none
for the per-iteration PSNR (:274-279): squared-error partials are reduced on device and read
back once after the last iteration, when the log lines are emitted.

Documented deviations from the reference:
  * `demosaic_method` other than 'malvar2004' raises ValueError (the reference silently feeds
    zeros to the denoiser, :187-191);
  * `logf=None` is accepted (no-op writer); arrays may be NumPy or CUDA tensors;
  * log lines are printed after the loop instead of during it (identical text).
"""
import math

import numpy as np
import torch

from . import _lib, ops
from .metrics import psnr_frames, ssim_frames
from .nets import FFDNetEngine

F32 = torch.float32


# Test/diagnostic hook: callable(k, mosaic (H,W,B) CUDA tensor) invoked after every iteration with the
# iterate the reference reports (theta for the two-stage solver, x for the one-stage one).  Costs one
# extra layout kernel per iteration when set; None in production.
ITERATE_HOOK = None


class _NullLog:
    def write(self, s):
        pass


def _dev(a, device):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    return a.to(device=device, dtype=F32).contiguous()


def _as_lists(sigma, iter_max):
    if not isinstance(sigma, list):
        sigma = [sigma]
    if not isinstance(iter_max, list):
        iter_max = [iter_max] * len(sigma)
    return sigma, iter_max


class _Problem:
    """Device-resident plane-major problem state shared by both solvers (reference :48-95 / :335-381)."""

    def __init__(self, y_bayer, Phi_bayer, x0_bayer, X_orig):
        _lib.load()
        _lib.require_gpu()
        self.device = torch.device('cuda', torch.cuda.current_device())
        Phi = _dev(Phi_bayer, self.device)
        y = _dev(y_bayer, self.device)
        if Phi.dim() != 3 or y.shape != Phi.shape[:2] or Phi.shape[0] % 2 or Phi.shape[1] % 2:
            raise ValueError(f'expected y (H,W) and Phi (H,W,B) with even H,W; got {tuple(y.shape)} {tuple(Phi.shape)}')
        self.H, self.W, self.B = Phi.shape
        self.M, self.N = self.H // 2, self.W // 2
        self.Phi = ops.mosaic_to_state(Phi)
        self.y = ops.y_to_meas(y)
        self.Phisum, x0 = ops.pm_setup(self.Phi, self.y, want_x0=x0_bayer is None)
        if x0_bayer is not None:
            x0 = ops.mosaic_to_state(_dev(x0_bayer, self.device))
        self.theta = x0                               # start point; x and theta are one tensor in the reference
        self.x = torch.empty_like(x0)
        self.b = torch.zeros_like(x0)
        self.theta_raw = torch.empty_like(x0)
        self.orig = None
        self.orig_np = None
        if X_orig is not None:
            self.orig_np = X_orig if isinstance(X_orig, np.ndarray) else X_orig.detach().cpu().numpy()
            self.orig = ops.mosaic_to_state(_dev(X_orig, self.device))
        self.sse_rows = []

    def new_sse(self, nblocks):
        t = torch.empty(nblocks, dtype=torch.float64, device=self.device)
        self.sse_rows.append(t)
        return t

    def psnr_all(self):
        """One read-back for all iterations: PSNR_k = 10 log10(1 / mean sq err) (skimage formula)."""
        if not self.sse_rows:
            return []
        sse = torch.stack([r.sum() for r in self.sse_rows]).cpu().numpy()
        n = float(self.H) * self.W * self.B
        return [float(10 * np.log10(1.0 / (s / n))) for s in sse]


def _log_lines(denoiser, schedule, psnr_all, noise_estimate, logf, have_orig, two_stage):
    """Reference log text (dvp...:282-309 / :513-535), emitted after the loop."""
    name = denoiser.upper()
    k = 0
    for nsig, iters in schedule:
        for _ in range(iters):
            if have_orig and (k + 1) % 2 == 0:
                if not noise_estimate and nsig is not None:
                    if nsig < 1:
                        line = '  ADMM-{0} iteration {1: 3d}, sigma {2: 3g}/255, PSNR {3:2.2f} dB.'.format(
                            name, k + 1, nsig * 255, psnr_all[k])
                        print(line)
                        logf.write(line + ' \n')
                    else:
                        line = '  ADMM-{0} iteration {1: 3d}, sigma {2: 3g}, PSNR {3:2.2f} dB.'.format(
                            name, k + 1, nsig, psnr_all[k])
                        print(line)
                        logf.write(line + '\n')
                else:
                    line = '  ADMM-{0} iteration {1: 3d}, PSNR {2:2.2f} dB.'.format(name, k + 1, psnr_all[k])
                    print(line)
                    logf.write(line + '\n')
            k += 1
            if two_stage and (not have_orig) and ((k + 1) % 2 == 0):
                logf.write('  ADMM-{0} iteration {1: 3d}, sigma {2: 3g}/255 \n'.format(name, k + 1, nsig * 255))


def _final_report(P, mosaic_np):
    if P.orig_np is None:
        return [], []
    return psnr_frames(P.orig_np, mosaic_np), ssim_frames(P.orig_np, mosaic_np)


def _check_demosaic(demosaic_method, model_demosaic=None):
    if model_demosaic is not None:
        raise NotImplementedError('deep demosaicking (DDnet) is a "next" row of the scope table; pass model_demosaic=None')
    if demosaic_method != 'malvar2004':
        raise ValueError("demosaic_method must be 'malvar2004' (the reference's other branches are dead code)")


def twoStageAdmm_denoise_bayer(y_bayer, Phi_bayer, _lambda=1, gamma=0.01,
                               denoiser='tv', iter_max=50, noise_estimate=True, sigma=None,
                               x0_bayer=None,
                               X_orig=None, model_denoise=None, model_demosaic=None, show_iqa=True,
                               demosaic_method='malvar2004', lr_=0.000001,
                               inital_iter=1, interval_iter=5, logf=None, useGPU=True, update_=False,
                               update_per_iter=1, close_form_demosaic=False,
                               large=False, update_times=-1, args=None):
    if denoiser not in ('tv', 'ffdnet_color', 'fastdvd_color'):
        raise ValueError('Unsupported denoiser {}!'.format(denoiser))
    if close_form_demosaic:
        raise NotImplementedError('close_form_demosaic is a "next" row of the scope table')
    logf = logf or _NullLog()
    sigma, iter_max = _as_lists(sigma, iter_max)
    P = _Problem(y_bayer, Phi_bayer, x0_bayer, X_orig)
    iqa = bool(show_iqa and X_orig is not None)
    alpha = 0.01 if denoiser == 'tv' else 1
    rou = 0.55 if denoiser == 'fastdvd_color' else 1
    tau = 100
    inv_rho, inv_tau = 1 / rou, 1 / tau
    B, M, N, H, W = P.B, P.M, P.N, P.H, P.W
    total_iters = sum(iter_max)
    out_rgb = None
    if denoiser == 'tv':
        plan = ops.TvPlan(M, N, 4 * B, 5, P.device)
    else:
        _check_demosaic(demosaic_method, model_demosaic)
        w = torch.zeros(B, 3, H, W, dtype=F32, device=P.device)
        x_rgb = torch.empty_like(w)
        out_rgb = torch.empty_like(w)
        if denoiser == 'ffdnet_color':
            eng = FFDNetEngine(model_denoise, B, M, N, P.device)
        else:
            from .fastdvd import FastDVDEngine
            eng = FastDVDEngine(model_denoise, B, H, W, P.device)
            rgb_w = torch.empty_like(w)
    k = 0
    update_i = 0
    for idx, nsig in enumerate(sigma):
        for it in range(iter_max[idx]):
            ops.pm_project(P.theta, P.b, P.Phi, P.y, P.Phisum, 0, inv_rho, alpha * rou, out=P.x)
            last = (k == total_iters - 1)
            if denoiser == 'tv':
                ops.tv_chambolle(P.x.view(4 * B, M, N), P.b.view(4 * B, M, N), inv_rho,
                                 P.theta_raw.view(4 * B, M, N), plan, 0.1)
                nb = ops.sse_nblocks(P.x.numel()) if iqa else 0
                ops.pm_dual_update(P.theta_raw, P.x, P.theta, P.b, +1.0, P.orig if iqa else None,
                                   P.new_sse(nb) if iqa else None, which=0)
            else:
                gate = bool(update_ and k > inital_iter and k % interval_iter == 0)
                if denoiser == 'ffdnet_color':
                    ops.pm_pre_denoise(P.x, P.b, w, x_rgb, None, eng.in_c8, inv_rho, inv_tau, nsig)
                    if gate:
                        from .finetune import ffdnet_online_finetune
                        ffdnet_online_finetune(model_denoise, None, P.y, P.Phi, nsig, lr_, update_per_iter,
                                               engine=eng, logf=logf)
                    eng.forward()
                    src_rgb, src_c8 = None, eng.out_c8
                else:
                    ops.pm_pre_denoise(P.x, P.b, w, x_rgb, rgb_w, None, inv_rho, inv_tau, nsig)
                    if gate and (update_i < update_times or update_times < 0):
                        from .finetune import fastdvdnet_online_finetune
                        fastdvdnet_online_finetune(model_denoise, rgb_w, P.y, P.Phi, nsig, lr_, update_per_iter,
                                                   engine=eng, logf=logf)
                        update_i += 1
                    src_rgb, src_c8 = eng.forward(rgb_w, nsig), None
                ops.pm_post_denoise(src_rgb, src_c8, out_rgb if (last and src_c8 is not None) else None,
                                    P.x, x_rgb, P.theta, P.b, w, k == 0, P.orig if iqa else None,
                                    P.new_sse(ops.post_nblocks(M, N, B)) if iqa else None)
                if last and src_rgb is not None:
                    out_rgb = src_rgb
            if ITERATE_HOOK is not None:
                ITERATE_HOOK(k, ops.state_to_mosaic(P.theta))
            k += 1
    psnr_all = P.psnr_all()
    _log_lines(denoiser, list(zip(sigma, iter_max)), psnr_all, noise_estimate, logf, iqa, True)
    x_bayer_np = ops.state_to_mosaic(P.theta).cpu().numpy()
    psnr_, ssim_ = _final_report(P, x_bayer_np)
    if denoiser == 'tv':
        return x_bayer_np, psnr_, ssim_, psnr_all
    return ops.rgb_to_cube(out_rgb).cpu().numpy(), x_bayer_np, psnr_, ssim_, psnr_all, model_denoise, model_demosaic


def admm_denoise_bayer_demosaic_pre(y_bayer, Phi_bayer, _lambda=1, gamma=0.01,
                                    denoiser='tv', iter_max=50, noise_estimate=True, sigma=None,
                                    x0_bayer=None,
                                    X_orig=None, model=None, show_iqa=True, demosaic_method='malvar2004',
                                    lr_=0.000001,
                                    inital_iter=1, interval_iter=5, logf=None, useGPU=True, device=0,
                                    update_=False, update_per_iter=1):
    if denoiser not in ('tv', 'ffdnet_color', 'fastdvd_color'):
        raise ValueError('Unsupported denoiser {}!'.format(denoiser))
    logf = logf or _NullLog()
    sigma, iter_max = _as_lists(sigma, iter_max)
    P = _Problem(y_bayer, Phi_bayer, x0_bayer, X_orig)
    iqa = bool(show_iqa and X_orig is not None)
    B, M, N, H, W = P.B, P.M, P.N, P.H, P.W
    total_iters = sum(iter_max)
    out_rgb = None
    if denoiser == 'tv':
        plan = ops.TvPlan(M, N, 4 * B, 5, P.device)
    else:
        _check_demosaic(demosaic_method)
        x_rgb = torch.empty(B, 3, H, W, dtype=F32, device=P.device)
        out_rgb = torch.empty_like(x_rgb)
        if denoiser == 'ffdnet_color':
            eng = FFDNetEngine(model, B, M, N, P.device)
        else:
            from .fastdvd import FastDVDEngine
            eng = FastDVDEngine(model, B, H, W, P.device)
    k = 0
    for idx, nsig in enumerate(sigma):
        for it in range(iter_max[idx]):
            ops.pm_project(P.theta, P.b, P.Phi, P.y, P.Phisum, 1, _lambda, gamma, out=P.x)
            last = (k == total_iters - 1)
            if denoiser == 'tv':
                ops.tv_chambolle(P.x.view(4 * B, M, N), P.b.view(4 * B, M, N), -1.0,
                                 P.theta_raw.view(4 * B, M, N), plan, 0.1)
                nb = ops.sse_nblocks(P.x.numel()) if iqa else 0
                ops.pm_dual_update(P.theta_raw, P.x, P.theta, P.b, -1.0, P.orig if iqa else None,
                                   P.new_sse(nb) if iqa else None, which=1)
            else:
                # CNN branches of the one-stage solver (:439-496): demosaic(x - b), no w, b -= x - theta;
                # at k = 0 x and theta are one tensor, so x becomes the raw denoiser output (reported, :509)
                gate = bool(update_ and k > inital_iter and k % interval_iter == 0)
                neg_b = P.b.neg()
                if denoiser == 'ffdnet_color':
                    ops.pm_pre_denoise(P.x, neg_b, None, x_rgb, None, eng.in_c8, 1.0, 0.0, nsig)
                    if gate:
                        from .finetune import ffdnet_online_finetune
                        ffdnet_online_finetune(model, None, P.y, P.Phi, nsig, lr_, update_per_iter, engine=eng,
                                               logf=logf)
                    eng.forward()
                    src_rgb, src_c8 = None, eng.out_c8
                else:
                    ops.pm_pre_denoise(P.x, neg_b, None, x_rgb, None, None, 1.0, 0.0, nsig)
                    src_rgb, src_c8 = eng.forward(x_rgb, nsig), None
                # post kernel computes b_tmp = (-b) + (x_eff - theta) = -(b - (x_eff - theta)) -> negate back
                ops.pm_post_denoise(src_rgb, src_c8, out_rgb if (last and src_c8 is not None) else None,
                                    P.x, None, P.theta, neg_b, None, k == 0, None, None)
                P.b = neg_b.neg()
                if last and src_rgb is not None:
                    out_rgb = src_rgb
                if iqa:
                    ops.sse_partials(P.orig, P.x, P.new_sse(ops.sse_nblocks(P.x.numel())))
            if ITERATE_HOOK is not None:
                ITERATE_HOOK(k, ops.state_to_mosaic(P.x))
            k += 1
    psnr_all = P.psnr_all()
    _log_lines(denoiser, list(zip(sigma, iter_max)), psnr_all, noise_estimate, logf, iqa, False)
    x_bayer_np = ops.state_to_mosaic(P.x).cpu().numpy()
    psnr_, ssim_ = _final_report(P, x_bayer_np)
    if denoiser == 'tv':
        return x_bayer_np, psnr_, ssim_, psnr_all
    return ops.rgb_to_cube(out_rgb).cpu().numpy(), x_bayer_np, psnr_, ssim_, psnr_all, model


def admm_denoise(y, Phi, Phi_sum=None, denoiser='tv', **kw):
    """PnP-SCI-style alias named by the task brief: (y, Phi, Phi_sum, denoiser, ...) -> two-stage ADMM.
    Phi_sum is recomputed on device exactly as the reference does (:72-75); the argument is accepted
    for signature compatibility and checked for shape only."""
    if Phi_sum is not None and tuple(np.shape(Phi_sum)) != tuple(np.shape(y)):
        raise ValueError('Phi_sum must have the shape of y')
    return twoStageAdmm_denoise_bayer(y, Phi, denoiser=denoiser, **kw)


def gap_denoise(y, Phi, Phi_sum=None, denoiser='tv', **kw):
    """Alias for the one-stage ("GAP form") solver, see `admm_denoise`."""
    if Phi_sum is not None and tuple(np.shape(Phi_sum)) != tuple(np.shape(y)):
        raise ValueError('Phi_sum must have the shape of y')
    return admm_denoise_bayer_demosaic_pre(y, Phi, denoiser=denoiser, **kw)

"""Drop-in equivalents of the reference's two PnP-ADMM solvers, driving the HIP kernels.

twoStageAdmm_denoise_bayer       <- dvp_linear_inv_2_stage_ADMM_tensor_online.py:40-324
admm_denoise_bayer_demosaic_pre  <- dvp_linear_inv_2_stage_ADMM_tensor_online.py:326-552

Same positional/keyword arguments, same return tuples, same log text.  Everything between the
input conversion and the final read-back stays on the GPU in the plane-major layout
(include/scipnp.h): per iteration the host only enqueues kernels on the current HIP stream -- no
device->host copy for TV (the reference goes through NumPy every iteration, :153-160) and none
for the per-iteration PSNR (:274-279): squared-error partials are reduced on device; `psnr_all` is read
back once after the last iteration, the logged iterations' values stream out asynchronously.

`AdmmRun` is the stepper both entry points (and bench.py) drive: one `step()` = one ADMM iteration.

Documented deviations from the reference:
  * `demosaic_method` other than 'malvar2004' raises ValueError (the reference silently feeds
    zeros to the denoiser, :187-191);
  * `logf=None` is accepted (no-op writer); arrays may be NumPy or CUDA tensors.
Log lines are written while the iterations run, like the reference's (:282-309): the PSNR of a logged iteration is
reduced on the device, copied to page-locked host memory without blocking the stream, and its line is emitted as
soon as that copy has landed (in iteration order; nothing waits for it inside the loop).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, config, ops
from .metrics import frame_metrics
from .nets import FFDNetEngine

F32 = torch.float32

# Test/diagnostic hook: callable(k, mosaic (H,W,B) CUDA tensor) invoked after every iteration with the
# iterate the reference reports (theta for the two-stage solver, x for the one-stage one).  Costs one
# extra layout kernel per iteration when set; None in production.
ITERATE_HOOK = None

DENOISERS = ('tv', 'ffdnet_color', 'fastdvd_color')


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


_OVERFLOW_MSG = ('denoiser activations left fp16 range in the split-fp16 convolution path; rerun with '
                 'SCIPNP_CONV_PRECISION=f32 (inputs are expected in [0,1] units like the reference)')


class AdmmRun:
    """Device-resident state of one reconstruction and its per-iteration kernel sequence.

    two_stage=True  : p = theta - b/rho, denominator alpha*rho + Phi_sum, b += x - theta, w dual on the
                      RGB cube, reports theta                                   (reference :121-305)
    two_stage=False : v = theta + b, denominator Phi_sum + gamma, b -= x - theta, reports x (:385-536)
    """

    def __init__(self, *args, config=None, **kw):
        """arguments of _construct below, plus config= (adaptivepnp_sci_amd.config.Config, default: the configuration current at
        construction): the kernel forms of THIS solve -- precision, fp32 form, F(4x4) on / off, weight-gradient form, side
        streams, ADMM-TV paths -- kept for its whole life: construction, every step(), split() run under it whatever the
        environment or the caller's configuration says later.  conv_precision= is the short form of config.replace(precision=...)."""
        from . import config as _config
        base = config if config is not None else _config.current()
        prec = kw.get('conv_precision', args[19] if len(args) > 19 else None)
        self.config = base if prec is None else base.replace(precision=prec)
        self._pinned = _config.FIELDS if config is not None else (('precision',) if prec is not None else ())
        with _config.solve_scope(self.config, self._pinned):
            self._construct(*args, **kw)

    def step(self, *args, **kw):
        """one ADMM iteration (see _step) under this solve's configuration"""
        with config.solve_scope(self.config, self._pinned):
            return self._step(*args, **kw)

    def _construct(self, y_bayer, Phi_bayer, denoiser, two_stage, x0_bayer=None, X_orig=None, model=None,
                   show_iqa=True, _lambda=1, gamma=0.01, lr_=1e-6, inital_iter=1, interval_iter=5, update_=False,
                   update_per_iter=1, update_times=-1, logf=None, close_form_demosaic=False, model_demosaic=None,
                   conv_precision=None, Phi_sum=None, units=None):
        """units=U (round 4): a UNIT BATCH -- y_bayer, Phi_bayer (and x0_bayer, X_orig when given) are sequences of U
        problems of ONE shape that share the denoiser weights; they are stepped by ONE launch sequence (the reference loops
        its measurements one after the other, two_stage_ADMM_Online_FFD_Warm.py:241-275).  State layout [B][U][4][M][N]
        (frame f = t*U + u): the projection sees 4 M N U pixels, every other kernel B*U frames / 4*B*U planes; each unit's
        numbers are bit-identical to its own single-unit run.  Every denoiser and demosaic: FastDVDnet's and DDnet's temporal
        windows stay inside a unit (round 5: scipnp_fastdvd_pack_triplets_units, DDnetEngine(units=), the loop being replaced is
        two_stage_ADMM_Online_FastDVD_Warm.py:276-310);
        a finetune event needs per-unit weights: split() the batch before the gate fires.  result_mosaic(), psnr_all() and
        final_report() then return one entry per unit."""
        if str(denoiser).lower() not in DENOISERS:
            raise ValueError('Unsupported denoiser {}!'.format(denoiser))
        denoiser = denoiser.lower()                # the reference compares denoiser.lower() (:146, :164, :214)
        _lib.load()
        _lib.require_gpu()
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.denoiser, self.two_stage, self.model = denoiser, two_stage, model
        self.logf = logf or _NullLog()
        self.U = U = 1 if units is None else int(units)
        if units is not None:
            if U < 1 or len(y_bayer) != U or len(Phi_bayer) != U or any(v is not None and len(v) != U for v in (x0_bayer, X_orig)):
                raise ValueError(f'units={units}: y_bayer, Phi_bayer (x0_bayer, X_orig) must be sequences of {units} problems')
            if Phi_sum is not None:
                raise ValueError('unit batches: no Phi_sum (the admm_denoise / gap_denoise aliases solve one problem)')
            ys, Phis = list(y_bayer), list(Phi_bayer)
            x0s = None if x0_bayer is None else list(x0_bayer)
            origs = None if X_orig is None else list(X_orig)
        else:
            ys, Phis = [y_bayer], [Phi_bayer]
            x0s = None if x0_bayer is None else [x0_bayer]
            origs = None if X_orig is None else [X_orig]
        Phis = [_dev(p, self.device) for p in Phis]
        ys = [_dev(v, self.device) for v in ys]
        Phi, y = Phis[0], ys[0]
        if Phi.dim() != 3 or y.shape != Phi.shape[:2] or Phi.shape[0] % 2 or Phi.shape[1] % 2:
            raise ValueError(f'expected y (H,W) and Phi (H,W,B) with even H,W; got {tuple(y.shape)} {tuple(Phi.shape)}')
        if any(p.shape != Phi.shape for p in Phis) or any(v.shape != y.shape for v in ys):
            raise ValueError('unit batches: every unit must have the shape of the first one')
        self.H, self.W, self.B = Phi.shape
        self.M, self.N = self.H // 2, self.W // 2
        B, M, N, H, W = self.B, self.M, self.N, self.H, self.W
        self.BU = BU = B * U

        def batch_state(cubes):        # U cubes (H,W,B) -> [B][U][4][M][N] as a (B*U, 4, M, N) tensor (one cube: [B][4][M][N])
            st = [ops.mosaic_to_state(_dev(c, self.device)) for c in cubes]
            return st[0] if U == 1 else torch.stack(st, dim=1).reshape(BU, 4, M, N)

        # ---- setup (reference :59-95 / :347-381)
        self.Phi = batch_state(Phis)
        self.y = ops.y_to_meas(y) if U == 1 else torch.stack([ops.y_to_meas(v) for v in ys]).reshape(U * 4, M, N)
        if U == 1:
            self.Phisum, x0 = ops.pm_setup(self.Phi, self.y, want_x0=x0_bayer is None)
        else:
            self.Phisum, x0 = ops.pm_setup_units(self.Phi, self.y, U, want_x0=x0_bayer is None)
        if Phi_sum is not None:
            # admm_denoise / gap_denoise(y, Phi, Phi_sum, ...): the caller's normaliser (H,W) is USED as given, after the
            # reference's zeros -> 1 (:74-75 / :361-362); pm_setup's own sum of Phi over the frames is dropped
            ps = np.array(Phi_sum.detach().cpu().numpy() if torch.is_tensor(Phi_sum) else Phi_sum, dtype=np.float32)
            if ps.shape != tuple(y.shape):
                raise ValueError(f'Phi_sum must have the shape of y {tuple(y.shape)}, got {ps.shape}')
            ps[ps == 0] = 1
            self.Phisum = ops.y_to_meas(_dev(ps, self.device))
        if x0_bayer is not None:
            x0 = batch_state(x0s)
        self.theta = x0                  # x and theta are ONE tensor in the reference until the first clip
        self.x = torch.empty_like(x0)
        self.b = torch.zeros_like(x0)
        self.orig = self.orig_np = None
        if X_orig is not None:
            if U == 1:
                self.orig_np = X_orig if isinstance(X_orig, np.ndarray) else X_orig.detach().cpu().numpy()
            self.orig = batch_state(origs)
        self.iqa = bool(show_iqa and X_orig is not None)
        if U > 1 and denoiser == 'tv' and self.iqa and (4 * M * N) % 2048:
            raise ValueError('unit batches with per-iteration PSNR (X_orig + show_iqa) on the TV path need H*W to be a multiple of '
                             '2048: the squared-error partials are cut at unit boundaries')
        if U > 1 and denoiser == 'tv' and self.iqa:
            # the fused dual-update + projection launch writes one partial per workgroup: its workgroups must tile the units
            nblk = _lib.load().scipnp_pm_dual_project_blocks(M, N, B, U, ops.sse_nblocks(x0.numel()))
            if nblk % U:
                raise ValueError(f'unit batches with per-iteration PSNR on the TV path: the {nblk} workgroups of the fused launch '
                                 f'do not tile {U} units of {H}x{W}x{B}; drop X_orig / show_iqa or solve the units one by one')
        self.sse_rows = []
        # ---- constants (reference :101-110; one-stage uses _lambda/gamma directly)
        if two_stage:
            self.alpha = 0.01 if denoiser == 'tv' else 1
            self.rou = 0.55 if denoiser == 'fastdvd_color' else 1
            self.tau = 100
            if close_form_demosaic:          # reference :112-114: tau = 10, rho = 0.55 for both CNN denoisers
                self.tau = 10
                self.rou = 0.55
        else:
            self._lambda, self.gamma = _lambda, gamma
        self.lr_, self.inital_iter, self.interval_iter = lr_, inital_iter, interval_iter
        self.update_, self.update_per_iter, self.update_times = update_, update_per_iter, update_times
        self.close_form = bool(close_form_demosaic and two_stage and denoiser != 'tv')
        self.update_i = 0
        self.k = 0
        self.out_rgb = None
        self.profile_events = None       # bench.py: list receiving (start,end) events around the body convs
        self.phi_events = None           # bench.py: list receiving (start,end) events around the projection launch
        self.noise_source = None         # finetune.NoisePrefetch set by _run_schedule (FastDVDnet finetune noise)
        self._sse_fixed = None
        # range-guard word of THIS solve's split-fp16 launches (include/scipnp.h, scipnp_bind_overflow_word): bound around
        # every step, read once at the end -- a neighbouring solve on another host thread / stream has its own
        self.ovf_word = None
        self.conv_precision, self.model_demosaic = conv_precision, model_demosaic
        self._init_workspaces()

    def _init_workspaces(self):
        """prior workspaces for the state as it stands (TV plan + the one-call argument block, or the RGB buffers and the
        network engines); also run by split() for each unit of a batch"""
        denoiser, two_stage, model, conv_precision, model_demosaic = (self.denoiser, self.two_stage, self.model, self.conv_precision,
                                                                       self.model_demosaic)
        B, M, N, H, W, U, x0 = self.BU, self.M, self.N, self.H, self.W, self.U, self.theta      # (B: frames of the whole batch)
        if denoiser == 'tv':
            self.plan = ops.TvPlan(M, N, 4 * B, 5, self.device)
            self.theta_raw = torch.empty_like(x0)
            # the whole iteration is one C call (scipnp_admm_tv_iterate): ten launches of 5-25 us are host-bound when
            # issued one ctypes call at a time
            c0, c1 = (self.rou, self.alpha) if two_stage else (self._lambda, self.gamma)
            self._tv_args = _lib.AdmmTvArgs(M, N, self.B, int(two_stage), self.theta.data_ptr(), self.b.data_ptr(), self.x.data_ptr(),
                                            self.theta_raw.data_ptr(), self.Phi.data_ptr(), self.y.data_ptr(),
                                            self.Phisum.data_ptr(), float(c0), float(c1), 0.1, 5, self.plan.ptr,
                                            self.plan.nbytes, 0 if self.orig is None else self.orig.data_ptr(), 0)
            self._tv_args.units = U
            # two launches per iteration: the dual update of an iteration rides in the launch that projects the next one
            # (scipnp_admm_tv_args.defer_state); theta, b and the last squared-error row are brought up to date by flush(),
            # which every reader of the state calls.  SCIPNP_TV_DEFER=0: three launches, nothing pending between steps.
            self._tv_defer = C.c_int(0)
            self._tv_prev_row = None
            self._row_kinds = []             # unit batches: 'flat' / 'fused' per squared-error row (see _unit_rows)
            if config.current().tv_defer:
                self._tv_args.defer_state = C.pointer(self._tv_defer)
        else:
            self.x_rgb = torch.empty(B, 3, H, W, dtype=F32, device=self.device)
            self.w = torch.zeros_like(self.x_rgb) if two_stage else None
            self.out_store = torch.empty_like(self.x_rgb)
            # two-stage + Malvar: only the mosaic x + b/rho travels from the pre- to the post-denoiser kernel (4 E bytes each way instead of
            # x_rgb's 12 E; ops.pm_pre_denoise / pm_post_denoise mosaic=).  x_rgb itself stays for the closed-form and DDnet branches.
            self.mosaic = torch.empty_like(self.x) if two_stage else None
            # the range-guard word exists BEFORE the engines do: their constructors pack the split-fp16 weights, and a weight
            # outside the representable range (|w| >= 31.9, e.g. after a BatchNorm fold) must raise THIS solve's word
            from .nets import default_precision
            if (conv_precision or default_precision()) == 'f16x3':
                self.ovf_word = torch.zeros(1, dtype=torch.int32, device=self.device)
            with ops.overflow_scope(self.ovf_word):
                if denoiser == 'ffdnet_color':
                    self.eng = FFDNetEngine(model, B, M, N, self.device, precision=conv_precision)
                    if (getattr(self, 'update_', False) and getattr(self, 'U', 1) == 1 and getattr(self, 'update_per_iter', 0) > 0
                            and config.current().resident_trainer):
                        # the online finetune's trainer (device master weights, activation stash, workspaces) is built with the
                        # engine, like every other buffer of the solve; an event reuses it (finetune._FFDNetTrainer.reuse)
                        from .finetune import _FFDNetTrainer
                        self.eng._ft_trainer = _FFDNetTrainer(model, self.eng)
                else:
                    from .fastdvd import FastDVDEngine
                    self.eng = FastDVDEngine(model, B, H, W, self.device, precision=conv_precision, units=U)
                    self.rgb_w = torch.empty_like(self.x_rgb)
                self.dd = None
                if model_demosaic is not None:   # deep demosaicking instead of Malvar (reference :192-194 / :242-244)
                    if not two_stage:
                        raise ValueError('model_demosaic is an argument of the two-stage solver only (as in the reference)')
                    from .ddnet import DDnetEngine
                    self.dd = DDnetEngine(model_demosaic, B, H, W, self.device, precision=conv_precision, units=U)
                    self.dd_planes = torch.empty_like(x0)
                    self.dd_mosaic = torch.empty(B, H, W, dtype=F32, device=self.device)

    # ------------------------------------------------------------------ one ADMM iteration
    def _step(self, nsig, last=False):
        B, M, N = self.BU, self.M, self.N          # (B: frames of the whole unit batch)
        k = self.k
        if self.denoiser == 'tv' and self.phi_events is None:
            part = self._new_sse(ops.sse_nblocks(self.x.numel())) if self.iqa else None
            self._tv_args.sse_part = 0 if part is None else part.data_ptr()
            self._tv_args.sse_part_prev = 0 if self._tv_prev_row is None else self._tv_prev_row.data_ptr()
            self._tv_prev_row = part
            pending = self._tv_defer.value
            _lib.check(_lib.load().scipnp_admm_tv_iterate(C.byref(self._tv_args), None,
                                                          _lib.stream_ptr()),
                       'scipnp_admm_tv_iterate')
            if part is not None and self.U > 1:          # which launch wrote which row (their partial layouts differ, _unit_rows)
                if pending and len(self._row_kinds) >= 1:
                    self._row_kinds[-1] = 'fused'         # the previous iteration's row: by this call's fused launch
                self._row_kinds.append(None if self._tv_defer.value else self._tv_row_kind())
            if ITERATE_HOOK is not None:
                self.flush()
                ITERATE_HOOK(k, self._reported_mosaic())
            self.k += 1
            return
        self.flush()
        if self.phi_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.two_stage:
            inv_rho = 1 / self.rou
            ops.pm_project(self.theta, self.b, self.Phi, self.y, self.Phisum, 0, inv_rho, self.alpha * self.rou,
                           out=self.x, units=self.U)
            coef, sign, which = inv_rho, +1.0, 0
        else:
            ops.pm_project(self.theta, self.b, self.Phi, self.y, self.Phisum, 1, self._lambda, self.gamma, out=self.x,
                           units=self.U)
            coef, sign, which = -1.0, -1.0, 1
        if self.phi_events is not None:
            ev[1].record()
            self.phi_events.append(ev)
        if self.denoiser == 'tv':
            ops.tv_chambolle(self.x.view(4 * B, M, N), self.b.view(4 * B, M, N), coef,
                             self.theta_raw.view(4 * B, M, N), self.plan, 0.1)
            part = self._new_sse(ops.sse_nblocks(self.x.numel())) if self.iqa else None
            ops.pm_dual_update(self.theta_raw, self.x, self.theta, self.b, sign, self.orig if self.iqa else None,
                               part, which=which)
        else:
            with ops.overflow_scope(self.ovf_word):
                self._cnn_step(nsig, k, last)
        if ITERATE_HOOK is not None:
            ITERATE_HOOK(k, self._reported_mosaic())
        self.k += 1

    def _cnn_step(self, nsig, k, last):
        B, M, N = self.BU, self.M, self.N
        gate = bool(self.update_ and k > self.inital_iter and k % self.interval_iter == 0)
        if gate and self.U > 1:
            raise _lib.ScipnpError(f'iteration {k} fires the online finetune: the units of a batch share ONE set of weights; '
                                   'split() the batch into per-unit runs (each with its own model) before this iteration')
        if self.two_stage:
            b_in, inv_rho, inv_tau, w = self.b, 1 / self.rou, 1 / self.tau, self.w
        else:
            # one-stage CNN branches (:439-496): demosaic(x - b), no w dual, b -= x - theta.  Run the
            # kernels on -b:  x + 1*(-b) = x - b exactly, and (-b) + (x - theta) = -(b - (x - theta)).
            b_in, inv_rho, inv_tau, w = ops.negate(self.b), 1.0, 0.0, None
        closed = self.close_form and k > 0      # closed-form RGB update (reference :175-182 / :224-230), Malvar at k = 0
        pending = None
        via_mosaic = False                      # this iteration's pre kernel stored the mosaic instead of x_rgb
        if self.denoiser == 'ffdnet_color':
            split = self.eng.precision == 'f16x3'
            # the finetune's weight-gradient kernel (fp32 MFMA) needs the fp32 c8 input as well
            c8 = self.eng.in_c8 if (gate or not split) else None
            c8s = self.eng.in_c8s if split else None
            if closed:
                ops.pm_pre_closed_form(self.x, self.b, w, self.out_store, self.x_rgb, None, c8, self.rou, self.tau, True,
                                       nsig, net_in_c8s=c8s)
            elif self.dd is not None:
                self._deep_demosaic(b_in, inv_rho)
                ops.pm_pre_rgb(w, self.x_rgb, None, c8, inv_tau, nsig, net_in_c8s=c8s)
            elif self.two_stage:
                ops.pm_pre_denoise(self.x, b_in, w, None, None, c8, inv_rho, inv_tau, nsig, net_in_c8s=c8s, mosaic=self.mosaic)
                via_mosaic = True
            else:
                ops.pm_pre_denoise(self.x, b_in, w, self.x_rgb, None, c8, inv_rho, inv_tau, nsig, net_in_c8s=c8s)
            if gate:
                from .finetune import ffdnet_online_finetune
                pending = ffdnet_online_finetune(self.model, self.eng, self.y, self.Phi, nsig, self.lr_, self.update_per_iter,
                                                 logf=self.logf, defer_write_back=True)
            self.eng.forward(events=self.profile_events)
            src_rgb, src_c8 = None, self.eng.out_c8
        else:
            net_in = self.rgb_w if self.two_stage else self.x_rgb
            if closed:
                ops.pm_pre_closed_form(self.x, self.b, w, self.eng.out, self.x_rgb, self.rgb_w, None, self.rou, self.tau,
                                       False, nsig)
            elif self.dd is not None:
                self._deep_demosaic(b_in, inv_rho)
                ops.pm_pre_rgb(w, self.x_rgb, self.rgb_w, None, inv_tau, nsig)
            elif self.two_stage:
                ops.pm_pre_denoise(self.x, b_in, w, None, self.rgb_w, None, inv_rho, inv_tau, nsig, mosaic=self.mosaic)
                via_mosaic = True
            else:
                ops.pm_pre_denoise(self.x, b_in, w, self.x_rgb, None, None, inv_rho, inv_tau, nsig)
            if gate and self.two_stage and (self.update_i < self.update_times or self.update_times < 0):
                from .finetune import fastdvdnet_online_finetune
                fastdvdnet_online_finetune(self.model, self.eng, net_in, self.y, self.Phi, nsig, self.lr_,
                                           self.update_per_iter, logf=self.logf,
                                           noise=None if self.noise_source is None else self.noise_source.get())
                self.update_i += 1
            src_rgb, src_c8 = self.eng.forward(net_in, nsig), None
        iqa_here = self.iqa and self.two_stage
        part = self._new_sse(ops.post_nblocks(M, N, B)) if iqa_here else None
        ops.pm_post_denoise(src_rgb, src_c8, self.out_store if ((last or self.close_form) and src_c8 is not None) else None,
                            self.x, self.x_rgb if (self.two_stage and not via_mosaic) else None, self.theta, b_in, w, k == 0,
                            self.orig if iqa_here else None, part, mosaic=self.mosaic if via_mosaic else None)
        if not self.two_stage:
            ops.negate(b_in, out=self.b)
            if self.iqa:
                ops.sse_partials(self.orig, self.x, self._new_sse(ops.sse_nblocks(self.x.numel())))
        if last:
            self.out_rgb = self.out_store if src_c8 is not None else src_rgb
        if pending is not None and hasattr(pending, 'finish_write_back'):
            pending.finish_write_back()         # the event's losses are printed and the module receives the updated weights once
            if not config.current().resident_trainer and getattr(self.eng, '_ft_trainer', None) is pending:
                self.eng._ft_trainer = None         # (opt-out: the trainer's ~GBs live for the event only; the engine keeps its own packs)
                                                # everything of this iteration is enqueued: both travelled beside the evaluation pass

    def _deep_demosaic(self, b_in, inv_rho):
        """x_rgb = DDnet(mosaic of x + b/rho)  (reference :168-171, :192-194)"""
        ops.pm_ddnet_inputs(self.x, b_in, inv_rho, self.dd_planes, self.dd_mosaic)
        self.dd.forward(self.dd_planes, self.dd_mosaic, self.x_rgb)

    def check_overflow(self):
        """raise if a split-fp16 launch of THIS solve wrote a value outside fp16's range since its word was last cleared
        (the solver entry points call this at the end of their schedule; bench.py after its timed steps)"""
        if self.ovf_word is not None and ops.split_overflow(word=self.ovf_word):
            raise _lib.ScipnpError(_OVERFLOW_MSG)

    def flush(self):
        """bring theta, b and the newest squared-error row up to date (the ADMM-TV path may leave the dual update of its last
        step pending, see __init__); a no-op otherwise"""
        if self.denoiser == 'tv' and self._tv_defer.value:
            _lib.check(_lib.load().scipnp_admm_tv_flush(C.byref(self._tv_args), None, _lib.stream_ptr()), 'scipnp_admm_tv_flush')
            if self._row_kinds and self._row_kinds[-1] is None:
                self._row_kinds[-1] = 'flat'              # (the flush's stand-alone dual update wrote the pending row)

    @property
    def log_lag(self):
        """1 while the squared error of step k only exists after step k + 1 (or flush()) has been enqueued"""
        return 1 if (self.denoiser == 'tv' and bool(self._tv_args.defer_state) and self.phi_events is None) else 0

    # ------------------------------------------------------------------ reporting
    def _new_sse(self, nblocks):
        if self._sse_fixed is not None:           # hipGraph replay: one fixed buffer, rows are collected on the device
            return self._sse_fixed
        t = torch.empty(nblocks, dtype=torch.float64, device=self.device)
        self.sse_rows.append(t)
        return t

    _SPLIT_KEEP = ('device', 'denoiser', 'two_stage', 'logf', 'H', 'W', 'B', 'M', 'N', 'iqa', 'alpha', 'rou', 'tau', '_lambda', 'gamma',
                   'lr_', 'inital_iter', 'interval_iter', 'update_', 'update_per_iter', 'update_times', 'close_form',
                   'noise_source', '_sse_fixed', 'conv_precision', 'model_demosaic', 'config', '_pinned')

    def prepare_split(self, models=None):
        """Build the U single-unit runs split() hands out -- their state buffers, RGB buffers, TV plans / network engines
        (packed from models[u], unit u's own copy of the denoiser; None: the batch's shared model) -- WITHOUT touching the
        batch: allocation and weight packing happen here, split() then only copies the state across.  Optional: split()
        prepares on its own when this was not called."""
        if self.U == 1:
            return
        with config.solve_scope(self.config, self._pinned):
            self._prepare_split(models)

    def _prepare_split(self, models):
        U, B, M, N = self.U, self.B, self.M, self.N
        self._split_runs = []
        for u in range(U):
            r = AdmmRun.__new__(AdmmRun)
            for name in self._SPLIT_KEEP:
                if name in self.__dict__:
                    setattr(r, name, self.__dict__[name])
            r.U, r.BU = 1, B
            r.model = self.model if models is None else models[u]
            r.Phi, r.orig = self._unit_state(self.Phi, u), (None if self.orig is None else self._unit_state(self.orig, u))
            r.y, r.Phisum = self.y[4 * u:4 * u + 4].contiguous(), self.Phisum[4 * u:4 * u + 4].contiguous()
            r.theta, r.x, r.b = (torch.empty(B, 4, M, N, dtype=F32, device=self.device) for _ in range(3))
            r.orig_np = None
            r.sse_rows, r.out_rgb, r.ovf_word, r.update_i, r.k = [], None, None, 0, 0
            r.profile_events = r.phi_events = None    # (bench.py's event lists stay with the batch)
            r._init_workspaces()
            self._split_runs.append(r)

    def split(self, models=None):
        """The units of a batch as U independent single-unit runs that continue exactly where the batch stands (iteration count,
        state, duals, RGB buffers, per-iteration squared-error history): for the part of a schedule that needs per-unit
        weights -- from the first online-finetune event on, models[u] (its own copy of the denoiser) is unit u's model.
        The batch must not be stepped afterwards."""
        if self.U == 1:
            return [self]
        self.flush()
        if getattr(self, '_split_runs', None) is None:
            self.prepare_split(models)
        U, B, H, W = self.U, self.B, self.H, self.W
        rows = self._unit_rows(self.sse_rows) if self.sse_rows else [[] for _ in range(U)]
        runs, self._split_runs = self._split_runs, None
        for u, r in enumerate(runs):
            r.k, r.update_i = self.k, self.update_i
            for name in ('theta', 'x', 'b'):
                getattr(r, name).copy_(getattr(self, name).view(B, U, 4, self.M, self.N)[:, u])
            r.sse_rows = list(rows[u])
            if self.denoiser != 'tv':
                for name in ('x_rgb', 'w', 'out_store'):            # (mosaic: written and read inside one step)
                    src = getattr(self, name)
                    if src is not None:
                        getattr(r, name).copy_(src.view(B, U, 3, H, W)[:, u])
                if self.denoiser == 'fastdvd_color':        # (the closed-form update reads the engine's last denoised frames)
                    r.eng.out.copy_(self.eng.out.view(B, U, 3, H, W)[:, u])
        return runs

    def _unit_state(self, t, u):
        """unit u of a batched state tensor [B][U][4][M][N] as its own contiguous [B][4][M][N]"""
        return t if self.U == 1 else t.view(self.B, self.U, 4, self.M, self.N)[:, u].contiguous()

    def _reported_mosaic(self):
        st = self.theta if self.two_stage else self.x
        if self.U == 1:
            return ops.state_to_mosaic(st)
        return [ops.state_to_mosaic(self._unit_state(st, u)) for u in range(self.U)]

    def _unit_rows(self, rows):
        """squared-error partial rows of a unit batch -> per unit, in the order of the unit's own single-unit run: the
        post-denoise kernel writes its partials frame-major (frame f = t*U + u), the flat dual-update / fused-projection kernels
        pixel-major over [B][U][4 M N] resp. [U][4 M N] with unit-aligned blocks (checked in __init__)"""
        U, B = self.U, self.B
        out = [[] for _ in range(U)]
        for i, r in enumerate(rows):
            n = r.numel()
            if self.denoiser == 'tv' and self._row_kinds[i] == 'plane':     # whole-plane kernel: one partial per plane, then zeros
                v = r[:4 * B * U].view(B, U, 4).permute(1, 0, 2).reshape(U, -1)
            elif self.denoiser != 'tv' or self._row_kinds[i] == 'flat':
                v = r.view(B, U, n // (B * U)).permute(1, 0, 2).reshape(U, -1)
            else:                                         # fused launch: one partial per workgroup over the pixel axis, then zeros
                g = self._tv_fused_blocks()
                v = r[:g].view(U, g // U)
            for u in range(U):
                out[u].append(v[u].contiguous())
        return out

    def _tv_row_kind(self):
        """which kernel writes the squared-error row of a NON-deferred ADMM-TV call on this state (csrc/iterate.hip): the
        whole-plane kernel ('plane': one partial per plane) or the stand-alone dual update ('flat')"""
        return 'plane' if _lib.load().scipnp_admm_tv_plane_path(C.byref(self._tv_args)) else 'flat'

    def _tv_fused_blocks(self):
        """workgroups (= real squared-error partials) of the fused dual-update + projection launch on this state"""
        return _lib.load().scipnp_pm_dual_project_blocks(self.M, self.N, self.B, self.U, ops.sse_nblocks(self.x.numel()))

    def psnr_all(self):
        """One read-back for all iterations: PSNR_k = 10 log10(1 / mean sq err) (skimage formula).  Unit batch: one list per unit."""
        if not self.sse_rows:
            return [] if self.U == 1 else [[] for _ in range(self.U)]
        self.flush()
        if self.U > 1:
            n = float(self.H) * self.W * self.B
            res = []
            for rows in self._unit_rows(self.sse_rows):
                n0 = rows[0].numel()
                if all(r.numel() == n0 for r in rows):
                    sse = ops.sum_rows_f64(torch.stack(rows)).cpu().numpy()
                else:                                     # (fused-launch rows and stand-alone dual-update rows differ in length)
                    sse = torch.cat([ops.sum_rows_f64(r) for r in rows]).cpu().numpy()
                res.append([float(10 * np.log10(1.0 / (v / n))) for v in sse])
            return res
        n0 = self.sse_rows[0].numel()
        if all(r.numel() == n0 for r in self.sse_rows):            # one launch for the whole table
            sse = ops.sum_rows_f64(torch.stack(self.sse_rows)).cpu().numpy()
        else:
            sse = torch.cat([ops.sum_rows_f64(r) for r in self.sse_rows]).cpu().numpy()
        n = float(self.H) * self.W * self.B
        return [float(10 * np.log10(1.0 / (s / n))) for s in sse]

    def result_mosaic(self):
        """(H,W,B) CUDA tensor of the reported iterate (theta two-stage, x one-stage; reference :312-315 / :538-541); unit batch:
        the list of the units' mosaics."""
        self.flush()
        return self._reported_mosaic()

    def final_report(self, mosaic_np=None):
        """per-frame PSNR / SSIM of the final reconstruction (reference :316-321 / :542-547), computed on the device"""
        if self.orig is None:
            return [], []
        self.flush()
        st = self.theta if self.two_stage else self.x
        if self.U > 1:
            return [frame_metrics(self._unit_state(self.orig, u), self._unit_state(st, u)) for u in range(self.U)]
        return frame_metrics(self.orig, st)


GRAY_DENOISERS = ('tv_gray', 'ffdnet_gray')


class GrayAdmmRun:
    """Grayscale (non-Bayer) PnP-ADMM, SURVEY 8(f) rank 4 -- PARITY UNPINNED: the reference has no grayscale solver.  This is
    its one-stage loop (dvp...:385-407, :500-509) with the Bayer split and the demosaic removed, i.e. the ADMM of the
    PnP-SCI family the reference derives from:
        v = theta + b;  x = v + lambda * Phi ((y - sum_t v Phi) / (Phi_sum + gamma));  theta = clip(D(x - b), 0, 1);
        b = b - (x - theta);   reports x
    with D = Chambolle TV on the full frames (weight 0.1, 5 iterations, 'tv_gray') or the model zoo's FFDNet-gray per frame
    ('ffdnet_gray', model_zoo/ffdnet_gray.pth, pinned against the reference network class in tests/golden/ffdnet_gray_*).
    The projection / dual update / PSNR partials are the plane-major kernels (per pixel: any consistent layout).  With
    FFDNet-gray the state lives pixel-unshuffled [B][4][M][N] -- exactly FFDNet's own 2x2 unshuffle -- so the network input
    is the state plus the sigma map; with TV the state is the frames themselves, [B][H][W]."""

    two_stage = False
    U = 1                  # (no unit batches in the grayscale mode)
    update_ = False
    update_i = 0
    update_times = -1
    inital_iter = 1
    interval_iter = 5
    noise_source = None
    phi_events = None
    ovf_word = None

    def __init__(self, y, Phi, denoiser, x0=None, X_orig=None, model=None, show_iqa=True, _lambda=1, gamma=0.01,
                 Phi_sum=None, conv_precision=None):
        denoiser = str(denoiser).lower()
        if denoiser == 'ffdnet':
            denoiser = 'ffdnet_gray'
        if denoiser not in GRAY_DENOISERS:
            raise ValueError('Unsupported denoiser {}!'.format(denoiser))
        _lib.load()
        _lib.require_gpu()
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.denoiser, self.model = denoiser, model
        Phi_d, y_d = _dev(Phi, self.device), _dev(y, self.device)
        if Phi_d.dim() != 3 or y_d.shape != Phi_d.shape[:2] or Phi_d.shape[0] % 2 or Phi_d.shape[1] % 2:
            raise ValueError(f'expected y (H,W) and Phi (H,W,B) with even H,W; got {tuple(y_d.shape)} {tuple(Phi_d.shape)}')
        self.H, self.W, self.B = Phi_d.shape
        self.M, self.N = self.H // 2, self.W // 2
        B, M, N, H, W = self.B, self.M, self.N, self.H, self.W
        self.unshuffled = denoiser == 'ffdnet_gray'
        self.Phi = self._state(Phi_d)
        self.y = ops.y_to_meas(y_d) if self.unshuffled else y_d.reshape(4, M, N)
        self.Phisum, x0_s = ops.pm_setup(self.Phi, self.y, want_x0=x0 is None)
        if Phi_sum is not None:
            ps = np.array(Phi_sum.detach().cpu().numpy() if torch.is_tensor(Phi_sum) else Phi_sum, dtype=np.float32)
            if ps.shape != (H, W):
                raise ValueError(f'Phi_sum must have the shape of y {(H, W)}, got {ps.shape}')
            ps[ps == 0] = 1
            ps = _dev(ps, self.device)
            self.Phisum = ops.y_to_meas(ps) if self.unshuffled else ps.reshape(4, M, N)
        if x0 is not None:
            x0_s = self._state(_dev(x0, self.device))
        self.theta = x0_s
        self.x = torch.empty_like(x0_s)
        self.b = torch.zeros_like(x0_s)
        self.theta_raw = torch.empty_like(x0_s)
        self.orig = None
        if X_orig is not None:
            self.orig = self._state(_dev(X_orig, self.device))
        self.iqa = bool(show_iqa and X_orig is not None)
        self._lambda, self.gamma = _lambda, gamma
        self.sse_rows = []
        self.k = 0
        if self.unshuffled:
            if model is None:
                raise ValueError("denoiser 'ffdnet_gray' needs model= (an FFDNet(in_nc=1, out_nc=1, nc=64, nb=15) with the "
                                 'ffdnet_gray weights loaded)')
            from .nets import default_precision
            if (conv_precision or default_precision()) == 'f16x3':      # before the engine packs its split-fp16 weights
                self.ovf_word = torch.zeros(1, dtype=torch.int32, device=self.device)
            with ops.overflow_scope(self.ovf_word):
                self.eng = FFDNetEngine(model, B, M, N, self.device, precision=conv_precision)
            if self.eng.in_ch != 5:
                raise ValueError('ffdnet_gray needs the grayscale network (5 -> nc -> 4 channels)')
        else:
            if (H * W) % 16:
                raise ValueError('tv_gray needs H*W to be a multiple of 16')
            self.plan = ops.TvPlan(H, W, B, 5, self.device)

    def _state(self, cube):
        """(H,W,B) device cube -> the run's state layout, as a [B][4][M][N] view for the per-pixel kernels"""
        if self.unshuffled:
            return ops.mosaic_to_state(cube)
        return ops.cube_to_frames(cube).view(self.B, 4, self.M, self.N)

    def _cube(self, state):
        return ops.state_to_mosaic(state) if self.unshuffled else ops.frames_to_cube(state.view(self.B, self.H, self.W))

    def step(self, nsig, last=False):
        B, M, N, H, W = self.B, self.M, self.N, self.H, self.W
        ops.pm_project(self.theta, self.b, self.Phi, self.y, self.Phisum, 1, self._lambda, self.gamma, out=self.x)
        if self.unshuffled:
            with ops.overflow_scope(self.ovf_word):
                ops.gray_net_input(self.x, self.b, nsig, self.eng.in_c8)
                if self.eng.in_c8s is not None:
                    ops.c8_to_c8s(self.eng.in_c8, out=self.eng.in_c8s)
                ops.gray_net_output(self.eng.forward(), self.theta_raw)
        else:
            ops.tv_chambolle(self.x.view(B, H, W), self.b.view(B, H, W), -1.0, self.theta_raw.view(B, H, W), self.plan, 0.1)
        part = None
        if self.iqa:
            part = torch.empty(ops.sse_nblocks(self.x.numel()), dtype=torch.float64, device=self.device)
            self.sse_rows.append(part)
        ops.pm_dual_update(self.theta_raw, self.x, self.theta, self.b, -1.0, self.orig if self.iqa else None, part, which=1)
        if ITERATE_HOOK is not None:
            ITERATE_HOOK(self.k, self._cube(self.x))
        self.k += 1

    psnr_all = AdmmRun.psnr_all
    check_overflow = AdmmRun.check_overflow

    def flush(self):                                   # (nothing is ever left pending on this path)
        pass

    def result_cube(self):
        return self._cube(self.x)

    def final_report(self):
        if self.orig is None:
            return [], []
        if self.unshuffled:
            return frame_metrics(self.orig, self.x)
        return frame_metrics(ops.mosaic_to_state(self._cube(self.orig)), ops.mosaic_to_state(self._cube(self.x)))


class PartLanes:
    """Steps the single-unit runs of a split() batch on `lanes` host threads, each with its own HIP stream (round 5).  After the
    first online-finetune event every unit has its own weights, so its launches can no longer ride in the batch's: one after
    the other they leave the chip partly idle (a 256x256x16 tile is 768 workgroups of the F(4x4) kernel for 512 slots: the
    second generation is half empty) and every event stalls the host twice (the loss read-back, the weights' write-back into
    the module).  Units are independent, so lane j steps parts[j::lanes] behind the caller's stream, and the caller's stream
    continues behind all lanes; each part keeps its own configuration (AdmmRun(config=)), a network pass inside a lane runs on
    one stream (the lanes are the concurrency).  Results are those of stepping the parts in order (the `loss:` lines of
    different units interleave)."""

    def __init__(self, parts, lanes=4):
        import concurrent.futures as cf
        self.parts = list(parts)
        self.n = max(1, min(int(lanes), len(self.parts)))
        dev = self.parts[0].device if self.parts else None
        self.streams = [torch.cuda.Stream(dev) for _ in range(self.n)] if self.n > 1 else []
        self.pool = cf.ThreadPoolExecutor(self.n) if self.n > 1 else None

    def _lane(self, j, sigma):
        dev = self.parts[0].device
        torch.cuda.set_device(dev)
        with torch.cuda.stream(self.streams[j]), config.use(streams=1):
            for p in self.parts[j::self.n]:
                p.step(sigma)

    def step(self, sigma):
        if self.n <= 1:
            for p in self.parts:
                p.step(sigma)
            return
        cur = torch.cuda.current_stream(self.parts[0].device)
        for st in self.streams:
            st.wait_stream(cur)
        try:
            for f in [self.pool.submit(self._lane, j, sigma) for j in range(self.n)]:
                f.result()
        finally:                               # also after a failed step: the caller's stream never runs ahead of queued lane work
            for st in self.streams:
                cur.wait_stream(st)

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()
            self.pool = None


def _count_finetune_events(update_, two_stage, denoiser, total, inital_iter, interval_iter, update_times, k0=0, done=0):
    """number of FastDVDnet finetune events a schedule of `total` iterations will fire (the gate of _cnn_step, evaluated
    ahead of time)"""
    if not (update_ and two_stage and denoiser == 'fastdvd_color'):
        return 0
    n = 0
    for k in range(k0, k0 + total):
        if k > inital_iter and k % interval_iter == 0 and (done + n < update_times or update_times < 0):
            n += 1
    return n


def _run_tv_graphed(run, total):
    """ADMM-TV iterations are ten small launches (projection, 6 for Chambolle, dual update, PSNR row) of 5-25 us each:
    launch-bound on the host for quarter-resolution planes up to ~128 x 128.  Every pointer and scalar of an iteration
    is fixed, so iteration 0 runs eagerly (lazy one-time setup), iteration 1 is captured into a hipGraph and replayed
    for the rest; the per-iteration squared-error partials go to a fixed buffer and are appended to a device table by
    an index_copy_ driven by a device-side counter inside the graph.  Opt-in (SCIPNP_HIPGRAPH=1): measured on MI355X at
    256 x 256 x 8, 50 iterations, the whole call takes 5.0 ms with capture + instantiate + 49 replays against 4.1 ms for
    50 eager iterations -- one solve is too short to amortise the capture; it pays for schedules of several hundred
    iterations."""
    run.flush()                              # a pending deferred dual update lands before deferral is switched off
    run._tv_args.defer_state = None          # (a captured step must leave its squared-error row complete)
    run.step(0)
    n = total - 1
    table = kdev = None
    if run.iqa:
        nb = ops.sse_nblocks(run.x.numel())
        table = torch.zeros(n, nb, dtype=torch.float64, device=run.device)
        kdev = torch.zeros(1, dtype=torch.int64, device=run.device)
        run._sse_fixed = torch.empty(nb, dtype=torch.float64, device=run.device)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    k0 = run.k
    try:
        with torch.cuda.graph(g):
            run.step(0)
            if table is not None:
                table.index_copy_(0, kdev, run._sse_fixed.unsqueeze(0))
                kdev.add_(1)
    finally:
        run._sse_fixed = None
        run.k = k0                                 # the capture recorded an iteration, it did not run one
    for _ in range(n):
        g.replay()
        run.k += 1
    if table is not None:
        run.sse_rows.extend(table[i] for i in range(n))


class _LogStream:
    """The reference's per-iteration log text (dvp...:282-309 / :513-535), written while the loop runs.  A logged
    iteration's squared error is summed on the device and copied to page-locked memory on the solver's stream; `poll`
    emits, in iteration order, every line whose copy has landed -- the loop never waits for one."""

    def __init__(self, run, denoiser, noise_estimate, logf, two_stage):
        self.run, self.name, self.noise_estimate, self.logf = run, denoiser.upper(), noise_estimate, logf
        self.two_stage = two_stage
        self.have_orig = run.iqa
        self.no_orig = run.orig is None
        self.pending = []
        self.n = float(run.H) * run.W * run.B
        self._pool, self._free = torch.empty(0, dtype=torch.float64), 0

    def after_step(self, k, nsig):
        """k = index of the iteration that has just been enqueued (0-based)"""
        if self.have_orig and (k + 1) % 2 == 0:
            if self._free >= self._pool.numel():                       # page-locked slots, 64 at a time
                self._pool, self._free = torch.empty(64, dtype=torch.float64).pin_memory(), 0
            host = self._pool[self._free:self._free + 1]
            self._free += 1
            host.copy_(ops.sum_rows_f64(self.run.sse_rows[k]), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.pending.append((k, nsig, host, ev))
        self.poll()
        if self.two_stage and self.no_orig and ((k + 2) % 2 == 0):     # only when X_orig is None (reference :307-309)
            self.logf.write('  ADMM-{0} iteration {1: 3d}, sigma {2: 3g}/255 \n'.format(self.name, k + 2, nsig * 255))

    def poll(self, block=False):
        while self.pending and (block or self.pending[0][3].query()):
            k, nsig, host, ev = self.pending.pop(0)
            ev.synchronize()
            psnr = float(10 * np.log10(1.0 / (float(host[0]) / self.n)))
            self._emit(k, nsig, psnr)

    def _emit(self, k, nsig, psnr):
        line, tail = iteration_log_line(self.name, k, nsig, psnr, self.noise_estimate)
        print(line)
        self.logf.write(line + tail)


def iteration_log_line(name, k, nsig, psnr, noise_estimate):
    """(line, line ending) of the reference's per-iteration PSNR report for 0-based iteration k (dvp...:282-304 / :513-535: three
    formats, the sigma < 1 one with a blank before the newline)"""
    if not noise_estimate and nsig is not None:
        if nsig < 1:
            return '  ADMM-{0} iteration {1: 3d}, sigma {2: 3g}/255, PSNR {3:2.2f} dB.'.format(name, k + 1, nsig * 255, psnr), ' \n'
        return '  ADMM-{0} iteration {1: 3d}, sigma {2: 3g}, PSNR {3:2.2f} dB.'.format(name, k + 1, nsig, psnr), '\n'
    return '  ADMM-{0} iteration {1: 3d}, PSNR {2:2.2f} dB.'.format(name, k + 1, psnr), '\n'


def _run_schedule(run, sigma, iter_max, log=None):
    """the iterations of one solver call; `log`: _LogStream (lines stream out during the loop) or None"""
    total = sum(iter_max)
    if (run.denoiser == 'tv' and total >= 4 and ITERATE_HOOK is None and run.phi_events is None
            and (run.config.hipgraph if 'hipgraph' in getattr(run, '_pinned', ()) else config.current().hipgraph)):
        _run_tv_graphed(run, total)
        if log is not None:                              # graph replay: the values exist only after the replays
            k = 0
            for nsig, iters in zip(sigma, iter_max):
                for _ in range(iters):
                    log.after_step(k, nsig)
                    k += 1
            log.poll(block=True)
        return
    split = getattr(run, 'eng', None) is not None and run.eng.precision == 'f16x3'
    own_noise = None
    if run.noise_source is None:
        n_events = _count_finetune_events(run.update_, run.two_stage, run.denoiser, total, run.inital_iter, run.interval_iter,
                                          run.update_times, run.k, run.update_i)
        if n_events:
            # drawn ahead on a worker thread (the 65 ms NumPy draw overlaps the iterations before the gate); created only
            # now that the run exists, closed below whatever happens (unconsumed draws are given back to the global RNG)
            from .finetune import NoisePrefetch
            own_noise = run.noise_source = NoisePrefetch((run.B, 3, run.H, run.W), n_events)
    try:
        held = None                              # (k, nsig) of a step whose squared-error row exists one step later
        for idx, nsig in enumerate(sigma):
            for _ in range(iter_max[idx]):
                k = run.k
                run.step(nsig, last=(run.k == total - 1))
                if log is not None:
                    if held is not None:
                        log.after_step(*held)
                        held = None
                    if getattr(run, 'log_lag', 0):
                        held = (k, nsig)
                    else:
                        log.after_step(k, nsig)
        if log is not None:
            if held is not None:
                run.flush()
                log.after_step(*held)
            log.poll(block=True)
    finally:
        if own_noise is not None:
            own_noise.close()
            run.noise_source = None
    if split:
        run.check_overflow()


def _check_demosaic(denoiser, demosaic_method, model_demosaic=None):
    if denoiser == 'tv':
        return
    if model_demosaic is not None:
        return                                  # deep demosaicking: demosaic_method is not consulted (reference :185/:192)
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
    """dvp_linear_inv_2_stage_ADMM_tensor_online.py:40-324, same arguments and return tuple"""
    return _two_stage(y_bayer, Phi_bayer, denoiser, iter_max, noise_estimate, sigma, x0_bayer, X_orig, model_denoise,
                      model_demosaic, show_iqa, demosaic_method, lr_, inital_iter, interval_iter, logf, update_,
                      update_per_iter, close_form_demosaic, update_times, None)


def _two_stage(y_bayer, Phi_bayer, denoiser, iter_max, noise_estimate, sigma, x0_bayer, X_orig, model_denoise,
               model_demosaic, show_iqa, demosaic_method, lr_, inital_iter, interval_iter, logf, update_, update_per_iter,
               close_form_demosaic, update_times, _Phi_sum):
    if str(denoiser).lower() not in DENOISERS:
        raise ValueError('Unsupported denoiser {}!'.format(denoiser))
    denoiser = denoiser.lower()
    _check_demosaic(denoiser, demosaic_method, model_demosaic)
    logf = logf or _NullLog()
    sigma, iter_max = _as_lists(sigma, iter_max)
    run = AdmmRun(y_bayer, Phi_bayer, denoiser, True, x0_bayer, X_orig, model_denoise, show_iqa, lr_=lr_,
                  inital_iter=inital_iter, interval_iter=interval_iter, update_=update_,
                  update_per_iter=update_per_iter, update_times=update_times, logf=logf,
                  close_form_demosaic=close_form_demosaic, model_demosaic=model_demosaic, Phi_sum=_Phi_sum)
    _run_schedule(run, sigma, iter_max, _LogStream(run, denoiser, noise_estimate, logf, True))
    psnr_all = run.psnr_all()
    x_bayer_np = ops.to_host(run.result_mosaic())
    psnr_, ssim_ = run.final_report(x_bayer_np)
    if denoiser == 'tv':
        return x_bayer_np, psnr_, ssim_, psnr_all
    return ops.to_host(ops.rgb_to_cube(run.out_rgb)), x_bayer_np, psnr_, ssim_, psnr_all, model_denoise, model_demosaic


def admm_denoise_bayer_demosaic_pre(y_bayer, Phi_bayer, _lambda=1, gamma=0.01,
                                    denoiser='tv', iter_max=50, noise_estimate=True, sigma=None,
                                    x0_bayer=None,
                                    X_orig=None, model=None, show_iqa=True, demosaic_method='malvar2004',
                                    lr_=0.000001,
                                    inital_iter=1, interval_iter=5, logf=None, useGPU=True, device=0,
                                    update_=False, update_per_iter=1):
    """dvp_linear_inv_2_stage_ADMM_tensor_online.py:326-552, same arguments and return tuple"""
    return _one_stage(y_bayer, Phi_bayer, _lambda, gamma, denoiser, iter_max, noise_estimate, sigma, x0_bayer, X_orig, model,
                      show_iqa, demosaic_method, lr_, inital_iter, interval_iter, logf, update_, update_per_iter, None)


def _one_stage(y_bayer, Phi_bayer, _lambda, gamma, denoiser, iter_max, noise_estimate, sigma, x0_bayer, X_orig, model,
               show_iqa, demosaic_method, lr_, inital_iter, interval_iter, logf, update_, update_per_iter, _Phi_sum):
    if str(denoiser).lower() not in DENOISERS:
        raise ValueError('Unsupported denoiser {}!'.format(denoiser))
    denoiser = denoiser.lower()
    _check_demosaic(denoiser, demosaic_method)
    logf = logf or _NullLog()
    sigma, iter_max = _as_lists(sigma, iter_max)
    run = AdmmRun(y_bayer, Phi_bayer, denoiser, False, x0_bayer, X_orig, model, show_iqa, _lambda=_lambda, gamma=gamma,
                  lr_=lr_, inital_iter=inital_iter, interval_iter=interval_iter, update_=update_,
                  update_per_iter=update_per_iter, logf=logf, Phi_sum=_Phi_sum)
    _run_schedule(run, sigma, iter_max, _LogStream(run, denoiser, noise_estimate, logf, False))
    psnr_all = run.psnr_all()
    x_bayer_np = ops.to_host(run.result_mosaic())
    psnr_, ssim_ = run.final_report(x_bayer_np)
    if denoiser == 'tv':
        return x_bayer_np, psnr_, ssim_, psnr_all
    return ops.to_host(ops.rgb_to_cube(run.out_rgb)), x_bayer_np, psnr_, ssim_, psnr_all, model


def admm_denoise_gray(y, Phi, Phi_sum=None, _lambda=1, gamma=0.01, denoiser='tv_gray', iter_max=50, noise_estimate=True,
                      sigma=None, x0=None, X_orig=None, model=None, show_iqa=True, logf=None):
    """Grayscale (non-Bayer) PnP-ADMM on a (H,W,B) cube, see `GrayAdmmRun` (parity unpinned: the reference has no such
    solver).  denoiser: 'tv_gray' or 'ffdnet_gray' (alias 'ffdnet'; model = FFDNet(in_nc=1, out_nc=1,
    nc=64, nb=15) with model_zoo/ffdnet_gray.pth).  Arguments and log text follow `admm_denoise_bayer_demosaic_pre`;
    returns (x (H,W,B), psnr per frame, ssim per frame, psnr_all)."""
    logf = logf or _NullLog()
    sigma, iter_max = _as_lists(sigma, iter_max)
    run = GrayAdmmRun(y, Phi, denoiser, x0, X_orig, model, show_iqa, _lambda, gamma, Phi_sum)
    _run_schedule(run, sigma, iter_max, _LogStream(run, run.denoiser, noise_estimate, logf, False))
    psnr_all = run.psnr_all()
    psnr_, ssim_ = run.final_report()
    return ops.to_host(run.result_cube()), psnr_, ssim_, psnr_all


def _is_gray(denoiser):
    return str(denoiser).lower() in GRAY_DENOISERS + ('ffdnet',)


def _bind(fn, y, Phi, denoiser, kw):
    """the public entry point's own defaults for everything the alias caller left out"""
    import inspect
    ba = inspect.signature(fn).bind(y, Phi, denoiser=denoiser, **kw)
    ba.apply_defaults()
    return ba.arguments


def admm_denoise(y, Phi, Phi_sum=None, denoiser='tv', **kw):
    """PnP-SCI-style alias named by the task brief: (y, Phi, Phi_sum, denoiser, ...) -> two-stage ADMM; keyword arguments
    as `twoStageAdmm_denoise_bayer`.  Phi_sum (H,W), if given, IS the normaliser of the Euclidean projection (after the
    reference's zeros -> 1, :74-75): pass the sum of Phi over the frames to reproduce `twoStageAdmm_denoise_bayer` bit for
    bit, or e.g. the sum of Phi**2 for non-binary masks.  None: computed on the device as the reference does (:72-75).
    denoiser 'tv_gray' / 'ffdnet_gray': the grayscale (non-Bayer) mode, `admm_denoise_gray` (its keyword arguments)."""
    if _is_gray(denoiser):
        return admm_denoise_gray(y, Phi, Phi_sum, denoiser=denoiser, **kw)
    a = _bind(twoStageAdmm_denoise_bayer, y, Phi, denoiser, kw)
    return _two_stage(a['y_bayer'], a['Phi_bayer'], a['denoiser'], a['iter_max'], a['noise_estimate'], a['sigma'],
                      a['x0_bayer'], a['X_orig'], a['model_denoise'], a['model_demosaic'], a['show_iqa'],
                      a['demosaic_method'], a['lr_'], a['inital_iter'], a['interval_iter'], a['logf'], a['update_'],
                      a['update_per_iter'], a['close_form_demosaic'], a['update_times'], Phi_sum)


def gap_denoise(y, Phi, Phi_sum=None, denoiser='tv', **kw):
    """Alias for the one-stage ("GAP form") solver `admm_denoise_bayer_demosaic_pre`; Phi_sum as in `admm_denoise`
    (reference :359-362); the gray denoisers go to `admm_denoise_gray`."""
    if _is_gray(denoiser):
        return admm_denoise_gray(y, Phi, Phi_sum, denoiser=denoiser, **kw)
    a = _bind(admm_denoise_bayer_demosaic_pre, y, Phi, denoiser, kw)
    return _one_stage(a['y_bayer'], a['Phi_bayer'], a['_lambda'], a['gamma'], a['denoiser'], a['iter_max'],
                      a['noise_estimate'], a['sigma'], a['x0_bayer'], a['X_orig'], a['model'], a['show_iqa'],
                      a['demosaic_method'], a['lr_'], a['inital_iter'], a['interval_iter'], a['logf'], a['update_'],
                      a['update_per_iter'], Phi_sum)

"""Thin Python wrappers over the C ABI (include/scipnp.h).  PyTorch-ROCm is used only as the
device allocator and stream provider: every function takes CUDA(ROCm) float32 tensors, checks
them, and passes raw device pointers plus the current HIP stream to libscipnp.so.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib

F32 = torch.float32


def _stream():
    return _lib.stream_ptr()


def _p(t, name='tensor', dtype=F32):
    if t is None:
        return C.c_void_p(0)
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.ScipnpError(f'{name} must be a CUDA/ROCm tensor (the hot path has no CPU fallback)')
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}')
    return C.c_void_p(t.data_ptr())


def _call(name, *args):
    lib = _lib.load()
    _lib.require_gpu()
    _lib.check(getattr(lib, name)(*args), name)


# Measurement hook (bench.py's per-layer-class tables): while a list is installed, every convolution wrapper below
# brackets its launch with a HIP event pair on the current stream and appends
# (kernel family, n, Cin, Cout, h, w, flags, start, end).  None in production: no events, no overhead.
LAUNCH_LOG = None


def _timed_call(family, shape, name, *args):
    log = LAUNCH_LOG
    if log is None:
        return _call(name, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _call(name, *args)
    e1.record()
    log.append((family,) + tuple(shape) + (e0, e1))


# ------------------------------------------------------------------ reference-layout operators
def A_(x, Phi):
    """y = sum_t x*Phi on planes (M,N,B,4) -> (M,N,4); reference utilspy.py:28-33."""
    M, N, B, _ = Phi.shape
    y = torch.empty(M, N, 4, device=Phi.device, dtype=F32)
    _call('scipnp_A', _p(x, 'x'), _p(Phi, 'Phi'), _p(y), M, N, B, _stream())
    return y


def At_(y, Phi):
    """x = y*Phi broadcast over frames; reference utilspy.py:35-44."""
    M, N, B, _ = Phi.shape
    x = torch.empty_like(Phi)
    _call('scipnp_At', _p(y, 'y'), _p(Phi, 'Phi'), _p(x), M, N, B, _stream())
    return x


def phisum(Phi):
    M, N, B, _ = Phi.shape
    out = torch.empty(M, N, 4, device=Phi.device, dtype=F32)
    _call('scipnp_phisum', _p(Phi, 'Phi'), _p(out), M, N, B, _stream())
    return out


def bayer_split(mosaic):
    H, W, B = mosaic.shape
    out = torch.empty(H // 2, W // 2, B, 4, device=mosaic.device, dtype=F32)
    _call('scipnp_bayer_split', _p(mosaic, 'mosaic'), _p(out), H // 2, W // 2, B, _stream())
    return out


def bayer_merge(planes):
    M, N, B, _ = planes.shape
    out = torch.empty(2 * M, 2 * N, B, device=planes.device, dtype=F32)
    _call('scipnp_bayer_merge', _p(planes, 'planes'), _p(out), M, N, B, _stream())
    return out


def proj_twostage(theta, b, Phi, y, Phisum, inv_rho, alpha_rho, out=None):
    M, N, B, _ = Phi.shape
    out = torch.empty_like(theta) if out is None else out
    _call('scipnp_proj_twostage', _p(theta, 'theta'), _p(b, 'b'), _p(Phi, 'Phi'), _p(y, 'y'), _p(Phisum, 'Phisum'),
          _p(out, 'out'), M, N, B, float(np.float32(inv_rho)), float(np.float32(alpha_rho)), _stream())
    return out


def proj_onestage(theta, b, Phi, y, Phisum, lam, gamma, out=None):
    M, N, B, _ = Phi.shape
    out = torch.empty_like(theta) if out is None else out
    _call('scipnp_proj_onestage', _p(theta, 'theta'), _p(b, 'b'), _p(Phi, 'Phi'), _p(y, 'y'), _p(Phisum, 'Phisum'),
          _p(out, 'out'), M, N, B, float(np.float32(lam)), float(np.float32(gamma)), _stream())
    return out


# ------------------------------------------------------------------ layout conversions
def mosaic_to_state(mosaic):
    H, W, B = mosaic.shape
    out = torch.empty(B, 4, H // 2, W // 2, device=mosaic.device, dtype=F32)
    _call('scipnp_mosaic_to_state', _p(mosaic, 'mosaic'), _p(out), H // 2, W // 2, B, _stream())
    return out


def state_to_mosaic(state):
    B, _, M, N = state.shape
    out = torch.empty(2 * M, 2 * N, B, device=state.device, dtype=F32)
    _call('scipnp_state_to_mosaic', _p(state, 'state'), _p(out), M, N, B, _stream())
    return out


def y_to_meas(y):
    H, W = y.shape
    out = torch.empty(4, H // 2, W // 2, device=y.device, dtype=F32)
    _call('scipnp_y_to_meas', _p(y, 'y'), _p(out), H // 2, W // 2, _stream())
    return out


def rgb_to_cube(rgb):
    B, _, H, W = rgb.shape
    out = torch.empty(H, W, 3, B, device=rgb.device, dtype=F32)
    _call('scipnp_rgb_to_cube', _p(rgb, 'rgb'), _p(out), H, W, B, _stream())
    return out


def cube_to_rgb(cube):
    H, W, _, B = cube.shape
    out = torch.empty(B, 3, H, W, device=cube.device, dtype=F32)
    _call('scipnp_cube_to_rgb', _p(cube, 'cube'), _p(out), H, W, B, _stream())
    return out


# ------------------------------------------------------------------ plane-major engine
def pm_setup(Phi, y, want_x0=True):
    B, _, M, N = Phi.shape
    Phisum = torch.empty(4, M, N, device=Phi.device, dtype=F32)
    x0 = torch.empty_like(Phi) if want_x0 else None
    _call('scipnp_pm_setup', _p(Phi, 'Phi'), _p(y, 'y'), _p(Phisum), _p(x0), M, N, B, _stream())
    return Phisum, x0


def pm_project(theta, b, Phi, y, Phisum, mode, c0, c1, out, units=1):
    """units > 1: the unit-batched layout (include/scipnp.h, "Unit batches"): Phi [B*U][4][M][N] with frame f = t*U + u,
    y / Phisum [U*4][M][N]"""
    BU, _, M, N = Phi.shape
    if units == 1:
        _call('scipnp_pm_project', _p(theta, 'theta'), _p(b, 'b'), _p(Phi, 'Phi'), _p(y, 'y'), _p(Phisum, 'Phisum'),
              _p(out, 'out'), M, N, BU, int(mode), float(np.float32(c0)), float(np.float32(c1)), _stream())
    else:
        _call('scipnp_pm_project_units', _p(theta, 'theta'), _p(b, 'b'), _p(Phi, 'Phi'), _p(y, 'y'), _p(Phisum, 'Phisum'),
              _p(out, 'out'), M, N, BU // units, int(units), int(mode), float(np.float32(c0)), float(np.float32(c1)), _stream())
    return out


def pm_setup_units(Phi, y, units, want_x0=True):
    """pm_setup on the unit-batched layout: Phi [B*U][4][M][N] (frame f = t*U + u), y [U*4][M][N] -> Phisum [U*4][M][N], x0"""
    BU, _, M, N = Phi.shape
    Phisum = torch.empty(units * 4, M, N, device=Phi.device, dtype=F32)
    x0 = torch.empty_like(Phi) if want_x0 else None
    _call('scipnp_pm_setup_units', _p(Phi, 'Phi'), _p(y, 'y'), _p(Phisum), _p(x0), M, N, BU // units, int(units), _stream())
    return Phisum, x0


class TvPlan:
    """Workspace for the Chambolle TV prior on C channels of M x N (allocated once per solve)."""

    def __init__(self, M, N, C_, n_iter_max, device):
        self.M, self.N, self.C, self.n_iter_max = M, N, C_, n_iter_max
        nbytes = _lib.load().scipnp_tv_workspace_bytes(M, N, C_, n_iter_max)
        self.nbytes = nbytes
        # (no kernel needs an initial value any more -- the banded kernels keep per-band partial sums, no counters or atomics, and
        # write every word they later read -- but a zeroed workspace keeps debug reads of unused slots deterministic)
        self.ws = torch.zeros(nbytes + 256, dtype=torch.uint8, device=device)
        off = (-self.ws.data_ptr()) % 256
        self.ptr = self.ws.data_ptr() + off
        self.stop_iter = torch.empty(C_, dtype=torch.int32, device=device)


def tv_chambolle(x, b, coef, theta, plan, weight=0.1, eps=2e-4, kernel=0):
    """theta = TV(x + coef*b) channel by channel; x, b, theta: (C, M, N) views of plane-major state.
    kernel: 0 = the library's choice, 1 = tiled (a launch per iteration), 2 = whole plane (one launch, planes <= 128x128),
    3 = banded (one launch of many workgroups per plane + the stop-test launch; <= 256 columns, <= 5 iterations), 4 = banded
    candidate form (one launch, nothing recomputed) + a selection launch (n_iter >= 2)."""
    _call('scipnp_tv_chambolle_ex', _p(x, 'x'), _p(b, 'b'), float(np.float32(coef)), _p(theta, 'theta'),
          plan.M, plan.N, plan.C, float(np.float32(weight)), float(np.float32(eps)), plan.n_iter_max,
          C.c_void_p(plan.ptr), plan.nbytes, _p(plan.stop_iter, 'stop_iter', torch.int32), int(kernel), _stream())
    return theta


def pm_dual_update(theta_raw, x, theta, b, sign, orig=None, sse_part=None, which=0):
    B, _, M, N = x.shape
    nb = C.c_int(0)
    _call('scipnp_pm_dual_update', _p(theta_raw, 'theta_raw'), _p(x, 'x'), _p(theta, 'theta'), _p(b, 'b'),
          _p(orig, 'orig'), _p(sse_part, 'sse_part', torch.float64), int(which), float(sign), M, N, B,
          C.byref(nb), _stream())
    return nb.value


def pm_pre_denoise(x, b, w, x_rgb, rgb_w, net_in_c8, inv_rho, inv_tau, sigma, net_in_c8s=None, mosaic=None):
    """mosaic (B,4,M,N) given: x_rgb is not stored (pass None) -- the kernel keeps the mosaic x + b/rho it demosaicked instead, for
    pm_post_denoise(..., mosaic=) to rebuild x_rgb from (a third of the bytes; bit-identical w update)"""
    B, _, M, N = x.shape
    if mosaic is not None:
        if x_rgb is not None:
            raise ValueError('pm_pre_denoise: give x_rgb or mosaic, not both')
        _call('scipnp_pm_pre_denoise_mosaic', _p(x, 'x'), _p(b, 'b'), _p(w, 'w'), _p(mosaic, 'mosaic'), _p(rgb_w, 'rgb_w'),
              _p(net_in_c8, 'net_in_c8'), _p(net_in_c8s, 'net_in_c8s', torch.float16), M, N, B,
              float(np.float32(inv_rho)), float(np.float32(inv_tau)), float(np.float32(sigma)), _stream())
        return
    _call('scipnp_pm_pre_denoise_ex', _p(x, 'x'), _p(b, 'b'), _p(w, 'w'), _p(x_rgb, 'x_rgb'), _p(rgb_w, 'rgb_w'),
          _p(net_in_c8, 'net_in_c8'), _p(net_in_c8s, 'net_in_c8s', torch.float16), M, N, B,
          float(np.float32(inv_rho)), float(np.float32(inv_tau)), float(np.float32(sigma)), _stream())


def pm_pre_closed_form(x, b, w, out_prev, x_rgb, rgb_w, net_in_c8, rho, tau, clip, sigma, net_in_c8s=None):
    B, _, M, N = x.shape
    _call('scipnp_pm_pre_closed_form', _p(x, 'x'), _p(b, 'b'), _p(w, 'w'), _p(out_prev, 'out_prev'), _p(x_rgb, 'x_rgb'),
          _p(rgb_w, 'rgb_w'), _p(net_in_c8, 'net_in_c8'), _p(net_in_c8s, 'net_in_c8s', torch.float16), M, N, B,
          float(np.float32(rho)), float(np.float32(tau)), float(np.float32(1 / tau)), int(bool(clip)),
          float(np.float32(sigma)), _stream())


def pm_pre_rgb(w, x_rgb, rgb_w, net_in_c8, inv_tau, sigma, net_in_c8s=None):
    """x_rgb (B,3,H,W) given (deep demosaicking): rgb_w = x_rgb - inv_tau*w and/or the FFDNet input layouts."""
    B, _, H, W = x_rgb.shape
    _call('scipnp_pm_pre_rgb', _p(w, 'w'), _p(x_rgb, 'x_rgb'), _p(rgb_w, 'rgb_w'), _p(net_in_c8, 'net_in_c8'),
          _p(net_in_c8s, 'net_in_c8s', torch.float16), H // 2, W // 2, B, float(np.float32(inv_tau)),
          float(np.float32(sigma)), _stream())


def pm_post_denoise(out_rgb, out_c8, out_rgb_store, x, x_rgb, theta, b, w, first_iter_alias, orig=None,
                    sse_part=None, mosaic=None):
    B, _, M, N = x.shape
    nb = C.c_int(0)
    if mosaic is not None:                   # x_rgb rebuilt from the mosaic pm_pre_denoise(..., mosaic=) stored
        if x_rgb is not None:
            raise ValueError('pm_post_denoise: give x_rgb or mosaic, not both')
        _call('scipnp_pm_post_denoise_mosaic', _p(out_rgb, 'out_rgb'), _p(out_c8, 'out_c8'), _p(out_rgb_store, 'out_rgb_store'),
              _p(x, 'x'), _p(mosaic, 'mosaic'), _p(theta, 'theta'), _p(b, 'b'), _p(w, 'w'), _p(orig, 'orig'),
              _p(sse_part, 'sse_part', torch.float64), int(bool(first_iter_alias)), M, N, B, C.byref(nb), _stream())
        if sse_part is not None and nb.value != sse_part.numel():
            raise _lib.ScipnpError(f'SSE partial count mismatch: kernel wrote {nb.value}, buffer has {sse_part.numel()}')
        return nb.value
    _call('scipnp_pm_post_denoise', _p(out_rgb, 'out_rgb'), _p(out_c8, 'out_c8'), _p(out_rgb_store, 'out_rgb_store'),
          _p(x, 'x'), _p(x_rgb, 'x_rgb'), _p(theta, 'theta'), _p(b, 'b'), _p(w, 'w'), _p(orig, 'orig'),
          _p(sse_part, 'sse_part', torch.float64), int(bool(first_iter_alias)), M, N, B, C.byref(nb), _stream())
    if sse_part is not None and nb.value != sse_part.numel():
        raise _lib.ScipnpError(f'SSE partial count mismatch: kernel wrote {nb.value}, buffer has {sse_part.numel()}')
    return nb.value


def frame_metrics_nblocks(M, N, B, win=7):
    nb = C.c_int(0)
    lib = _lib.load()
    _lib.check(lib.scipnp_frame_metrics(C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), M, N, B, win, 1.0, C.byref(nb),
                                        C.c_void_p(0)), 'scipnp_frame_metrics')
    return nb.value


def frame_metrics(ref_state, img_state, part, win=7, data_range=1.0):
    B, _, M, N = ref_state.shape
    nb = C.c_int(0)
    _call('scipnp_frame_metrics', _p(ref_state, 'ref_state'), _p(img_state, 'img_state'), _p(part, 'part', torch.float64),
          M, N, B, win, float(data_range), C.byref(nb), _stream())
    if nb.value * B * 2 != part.numel():
        raise _lib.ScipnpError('frame-metrics partial buffer has the wrong size')


def sse_nblocks(n):
    nb = C.c_int(0)
    lib = _lib.load()
    _lib.check(lib.scipnp_sse_partials(C.c_void_p(1), C.c_void_p(1), n, C.c_void_p(0), C.byref(nb), C.c_void_p(0)),
               'scipnp_sse_partials')
    return nb.value


def sse_partials(a, b, part):
    nb = C.c_int(0)
    _call('scipnp_sse_partials', _p(a, 'a'), _p(b, 'b'), a.numel(), _p(part, 'part', torch.float64), C.byref(nb),
          _stream())
    return part


def post_nblocks(M, N, B):
    """number of SSE partials scipnp_pm_post_denoise writes (its launch grid; mirrored from demosaic.hip)"""
    threads = 256 if N >= 256 else (128 if N >= 128 else 64)
    return ((N + threads - 1) // threads) * M * B


def sse(a, b):
    """sum((a-b)^2) with float32 differences and float64 accumulation (skimage PSNR's MSE numerator)."""
    n = a.numel()
    part = torch.empty(sse_nblocks(n), dtype=torch.float64, device=a.device)
    nb = C.c_int(0)
    _call('scipnp_sse_partials', _p(a, 'a'), _p(b, 'b'), n, _p(part, 'part', torch.float64), C.byref(nb), _stream())
    return float(part.sum().item())


# ------------------------------------------------------------------ convolutions
def pack_conv3x3(weight, bias=None, bn_scale=None, bn_shift=None, Cin=None, Cout=None, device=None):
    """OIHW float32 weights (CPU or GPU tensor) -> packed device buffer for scipnp_conv3x3_c8."""
    w = weight.detach().to('cpu', F32).contiguous()
    co, ci = w.shape[0], w.shape[1]
    Cin = Cin or (ci + 7) // 8 * 8
    Cout = Cout or (co + 7) // 8 * 8
    lib = _lib.load()
    n = lib.scipnp_conv3x3_packed_floats(Cin, Cout)
    packed = torch.empty(n, dtype=F32)

    def hp(t):
        if t is None:
            return C.c_void_p(0), None
        t = t.detach().to('cpu', F32).contiguous()
        return C.c_void_p(t.data_ptr()), t

    bp, _b = hp(bias)
    sp, _s = hp(bn_scale)
    hp_, _h = hp(bn_shift)
    _lib.check(lib.scipnp_pack_conv3x3_weights(C.c_void_p(w.data_ptr()), bp, sp, hp_, ci, co, Cin, Cout,
                                               C.c_void_p(packed.data_ptr())), 'scipnp_pack_conv3x3_weights')
    return packed.to(device) if device is not None else packed


def conv3x3_c8(x, packed, Cout, relu=False, residual=None, out=None, head=False, stride2=False, shuffle=False):
    """3x3 conv on c8 activations [n][Cin/8][h][w][8]; Cout = channels the conv produces.  With `shuffle`
    the output is the PixelShuffle(2)-ed tensor [n][Cout/32][2h][2w][8]; with `stride2` it is half size."""
    n, cg, h, w, _ = x.shape
    if out is None:
        if shuffle:
            out = torch.empty(n, Cout // 32, 2 * h, 2 * w, 8, device=x.device, dtype=F32)
        elif stride2:
            out = torch.empty(n, Cout // 8, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 8, device=x.device, dtype=F32)
        else:
            out = torch.empty(n, Cout // 8, h, w, 8, device=x.device, dtype=F32)
    flags = ((1 if relu else 0) | (2 if residual is not None else 0) | (4 if stride2 else 0) | (8 if shuffle else 0) |
             (0x100 if head else 0))
    _timed_call('conv3x3_c8_kernel', (n, cg * 8, Cout, h, w, flags), 'scipnp_conv3x3_c8', _p(x, 'x'), _p(packed, 'packed'),
                _p(out, 'out'), _p(residual, 'residual'), n, cg * 8, Cout, h, w, flags, _stream())
    return out


def pack_conv3x3_wino(packed_f32, Cin, Cout, out=None):
    """fp32 direct packing (device buffer of pack_conv3x3 / pack_conv3x3_device) -> Winograd F(2x2,3x3) packing for
    conv3x3_c8w (U = G g G^T in double, on the device)."""
    if out is None:
        out = torch.empty(_lib.load().scipnp_conv3x3_wino_packed_floats(Cin, Cout), dtype=F32, device=packed_f32.device)
    _call('scipnp_pack_conv3x3_wino', _p(packed_f32, 'packed_f32'), _p(out, 'packed_wino'), Cin, Cout, _stream())
    return out


def conv3x3_c8w(x, packed_wino, Cout, relu=False, residual=None, mask_src=None, out=None, head=False, rows16=False,
                shuffle=False):
    """stride-1 3x3 conv on c8 activations in fp32 Winograd F(2x2,3x3) arithmetic (csrc/conv_wino.hip).
    rows16: 16-row workgroups of 8 waves instead of the default 8-row workgroups of 4 waves (same results).
    shuffle: PixelShuffle(2) folded into the store, out [n][Cout/32][2h][2w][8] (residual, if any, in that layout)."""
    n, cg, h, w, _ = x.shape
    if out is None:
        out = (torch.empty(n, Cout // 32, 2 * h, 2 * w, 8, device=x.device, dtype=F32) if shuffle else
               torch.empty(n, Cout // 8, h, w, 8, device=x.device, dtype=F32))
    flags = ((1 if relu else 0) | (2 if residual is not None else 0) | (16 if mask_src is not None else 0) |
             (0x100 if head else 0) | (0x200 if rows16 else 0) | (8 if shuffle else 0))
    if isinstance(packed_wino, WinoPacked):
        if packed_wino.f4 is not None and not rows16 and (packed_wino.w is None or wino_f4_enabled()):
            return conv3x3_c8w4(x, packed_wino.f4, Cout, relu=relu, residual=residual, mask_src=mask_src, out=out, head=head,
                                shuffle=shuffle)
        if packed_wino.w is None:
            raise _lib.ScipnpError('this layer holds only its F(4x4,3x3) packing (latched at construction): rows16 / the F(2x2) '
                                   'kernel cannot run it')
        packed_wino = packed_wino.w
    _timed_call('conv3x3_c8w_kernel', (n, cg * 8, Cout, h, w, flags), 'scipnp_conv3x3_c8w', _p(x, 'x'),
                _p(packed_wino, 'packed_wino'), _p(out, 'out'), _p(residual, 'residual'), _p(mask_src, 'mask_src'), n, cg * 8,
                Cout, h, w, flags, _stream())
    return out


def pack_conv3x3_wino4(packed_f32, Cin, Cout, out=None):
    """fp32 direct packing (device buffer) -> Winograd F(4x4,3x3) packing for conv3x3_c8w4 (U = G g G^T in double, on the device)"""
    if out is None:
        out = torch.empty(_lib.load().scipnp_conv3x3_wino4_packed_floats(Cin, Cout), dtype=F32, device=packed_f32.device)
    _call('scipnp_pack_conv3x3_wino4', _p(packed_f32, 'packed_f32'), _p(out, 'packed_wino4'), Cin, Cout, _stream())
    return out


def pack_conv3x3_device_multi(jobs, scales=None):
    """device-side packs of many layers in ONE launch: jobs = [(w, bias or None, packed, Cin_real, Cout_real, Cin, Cout,
    transpose_flip)], scales = per-job folded BatchNorm scale tensors (or None)"""
    n = len(jobs)
    if n == 0:
        return
    P, I = C.c_void_p * n, C.c_int * n
    ptr = lambda t: None if t is None else t.data_ptr()
    _call('scipnp_pack_conv3x3_device_multi', n, P(*[ptr(j[0]) for j in jobs]), P(*[ptr(j[1]) for j in jobs]),
          (P(*[ptr(t) for t in scales]) if scales is not None else None), P(*[ptr(j[2]) for j in jobs]),
          I(*[j[3] for j in jobs]), I(*[j[4] for j in jobs]), I(*[j[5] for j in jobs]), I(*[j[6] for j in jobs]),
          I(*[int(j[7]) for j in jobs]), _stream())


def pack_conv3x3_wino4_multi(jobs):
    """F(4x4,3x3) packs of many layers in ONE launch: jobs = [(packed_f32, packed_wino4, Cin, Cout)]"""
    n = len(jobs)
    if n == 0:
        return
    P, I = C.c_void_p * n, C.c_int * n
    _call('scipnp_pack_conv3x3_wino4_multi', n, P(*[j[0].data_ptr() for j in jobs]), P(*[j[1].data_ptr() for j in jobs]),
          I(*[j[2] for j in jobs]), I(*[j[3] for j in jobs]), _stream())


def conv3x3_c8w4(x, packed_wino4, Cout, relu=False, residual=None, mask_src=None, out=None, head=False, shuffle=False):
    """stride-1 3x3 conv on c8 activations in fp32 Winograd F(4x4,3x3) arithmetic (csrc/conv_wino4.hip): 2.25 multiply-adds per
    output on the matrix cores instead of 4 (conv3x3_c8w) or 9 (conv3x3_c8).
    shuffle: PixelShuffle(2) folded into the store, out [n][Cout/32][2h][2w][8] (residual, if any, in that layout)."""
    n, cg, h, w, _ = x.shape
    if out is None:
        out = (torch.empty(n, Cout // 32, 2 * h, 2 * w, 8, device=x.device, dtype=F32) if shuffle else
               torch.empty(n, Cout // 8, h, w, 8, device=x.device, dtype=F32))
    flags = ((1 if relu else 0) | (2 if residual is not None else 0) | (16 if mask_src is not None else 0) |
             (0x100 if head else 0) | (8 if shuffle else 0))
    _timed_call('conv3x3_c8w4_kernel', (n, cg * 8, Cout, h, w, flags), 'scipnp_conv3x3_c8w4', _p(x, 'x'),
                _p(packed_wino4, 'packed_wino4'), _p(out, 'out'), _p(residual, 'residual'), _p(mask_src, 'mask_src'), n, cg * 8,
                Cout, h, w, flags, _stream())
    return out


class WinoPacked:
    """Winograd-domain weights of one layer for conv3x3_c8w: `w` = the scipnp_pack_conv3x3_wino packing (every shape and
    epilogue), `f4` = the F(4x4,3x3) packing of scipnp_conv3x3_c8w4 for the layer shapes that kernel is used for, else None.
    (The persistent F(2x2) kernel of round 3, csrc/conv_winop.hip, lives in libscipnp_diag.so: measured, not adopted.)"""
    __slots__ = ('w', 'f4', 'cin', 'cout')

    def __init__(self, w, cin, cout, f4=None):
        self.w, self.f4, self.cin, self.cout = w, f4, cin, cout

    def data_ptr(self):                                   # (C-entry callers pass the classic packing)
        if self.w is None:
            raise _lib.ScipnpError('this layer holds only its F(4x4,3x3) packing (latched at construction): no F(2x2) buffer')
        return self.w.data_ptr()


def wino_f4_enabled():
    """SCIPNP_WINO_F4=0 keeps every fp32 Winograd layer on the F(2x2,3x3) kernel; default: layers with at least 16 input and 32
    output channels run as F(4x4,3x3) (csrc/conv_wino4.hip)."""
    from . import config
    return config.current().wino_f4


def wino_f4_shape(Cin, Cout):
    """layer shapes the F(4x4,3x3) kernel is used for: at least 24 output channels.  Its workgroup computes 32 output channels,
    so a narrower layer pays for the padding: 96 -> 16 (FFDNet's tail) 92 against 76 us, 24 -> 8 322 against 313 us on F(2x2),
    but 96 -> 24 (DDnet, 24 evaluations of 512 x 512) 1049 against 1374 us, 24 -> 24 386 against 516 us; the input width does
    not matter at these sizes (8 -> 96: 743 against 946 us; until round 5 the rule also asked for 16 input channels).
    profiles/r05zj_narrow_layers.txt"""
    return Cout >= 24


def pack_conv3x3_wino_both(packed_f32, Cin, Cout):
    """the Winograd packings of a layer from its fp32 direct packing (device buffers)"""
    w = pack_conv3x3_wino(packed_f32, Cin, Cout)
    f4 = pack_conv3x3_wino4(packed_f32, Cin, Cout) if (wino_f4_enabled() and wino_f4_shape(Cin, Cout)) else None
    return WinoPacked(w, Cin, Cout, f4)


def pack_conv3x3_split(weight, bias=None, Cin=None, Cout=None, device=None, bn_scale=None, bn_shift=None):
    """OIHW fp32 weights (+ optional folded eval-mode BatchNorm) -> packed split-fp16 buffer (uint8 tensor)."""
    w = weight.detach().to('cpu', F32).contiguous()
    co, ci = w.shape[0], w.shape[1]
    Cin = Cin or (ci + 7) // 8 * 8
    Cout = Cout or (co + 7) // 8 * 8
    lib = _lib.load()
    packed = torch.empty(lib.scipnp_conv3x3_split_packed_bytes(Cin, Cout), dtype=torch.uint8)
    keep = [None if t is None else t.detach().to('cpu', F32).contiguous() for t in (bias, bn_scale, bn_shift)]
    ptr = [C.c_void_p(0 if t is None else t.data_ptr()) for t in keep]
    _lib.check(lib.scipnp_pack_conv3x3_split_bn(C.c_void_p(w.data_ptr()), ptr[0], ptr[1], ptr[2], ci, co, Cin, Cout,
                                                C.c_void_p(packed.data_ptr())), 'scipnp_pack_conv3x3_split_bn')
    return packed.to(device) if device is not None else packed


def conv3x3_c8s(x, packed, Cout, relu=False, out=None, f32_out=False, head=False, stride2=False, shuffle=False,
                mask=None, residual=None, variant=0):
    """split-fp16 conv: x c8s [n][Cin/8][2][h][w][8] float16 -> c8s, or fp32 c8 if f32_out / shuffle
    (shuffle: PixelShuffle(2)-ed fp32 c8 [n][Cout/32][2h][2w][8]).  mask: c8s tensor of the output's shape, output
    zeroed where it is not positive (backward-data convolution of the finetune)."""
    n, cg, _two, h, w, _ = x.shape
    ho, wo = ((h - 1) // 2 + 1, (w - 1) // 2 + 1) if stride2 else (h, w)
    shuffle_c8s = bool(shuffle) and ((out is not None and out.dtype == torch.float16) or shuffle == 'c8s')
    if out is None:
        if shuffle_c8s:
            out = torch.empty(n, Cout // 32, 2, 2 * h, 2 * w, 8, device=x.device, dtype=torch.float16)
        elif shuffle:
            out = torch.empty(n, Cout // 32, 2 * h, 2 * w, 8, device=x.device, dtype=F32)
        elif f32_out:
            out = torch.empty(n, Cout // 8, ho, wo, 8, device=x.device, dtype=F32)
        else:
            out = torch.empty(n, Cout // 8, 2, ho, wo, 8, device=x.device, dtype=torch.float16)
    fp32 = f32_out or (shuffle and not shuffle_c8s)
    flags = ((1 if relu else 0) | (4 if stride2 else 0) | (8 if shuffle else 0) | (32 if f32_out else 0) |
             (64 if shuffle_c8s else 0) |
             (0x100 if head else 0) | (16 if mask is not None else 0) | (2 if residual is not None else 0) | variant)
    _timed_call('conv3x3_c8s_kernel', (n, cg * 8, Cout, h, w, flags), 'scipnp_conv3x3_c8s_ex', _p(x, 'x', torch.float16),
                _p(packed, 'packed', torch.uint8), _p(out, 'out', F32 if fp32 else torch.float16),
                _p(residual, 'residual', torch.float16), _p(mask, 'mask', torch.float16), n, cg * 8, Cout, h, w, flags,
                _stream())
    return out


def pack_conv3x3_split_device(w, bias, packed, Cin, Cout, transpose=False, scale=None):
    """device fp32 OIHW weights (+bias, + per-output-channel BN scale) -> `packed` (uint8 device buffer of
    scipnp_conv3x3_split_packed_bytes); transpose=True packs the backward-data convolution (buffer sized for (Cout, Cin))."""
    co, ci = w.shape[0], w.shape[1]
    _call('scipnp_pack_conv3x3_split_device_scaled', _p(w, 'w'), _p(bias, 'bias'), _p(scale, 'scale'),
          _p(packed, 'packed', torch.uint8), ci, co, Cin, Cout, int(bool(transpose)), _stream())
    return packed


def to_host(t):
    """device tensor -> NumPy array through a page-locked buffer of PyTorch's caching host allocator (the array keeps the
    buffer alive; it returns to the cache when the array is dropped).  A pageable `.cpu()` of the 25 MiB RGB cube + 8 MiB
    mosaic of a 512 x 512 x 8 reconstruction takes 3.8 ms on the MI355X boxes, this 0.6 ms (tools/probes/d2h_probe.py)."""
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return h.numpy()


def host_flat(tensors):
    """host tensors -> one flat float32 NumPy array.  NumPy's single-threaded copy on purpose: a CPU-side torch.cat /
    copy_ of a tensor above ATen's grain size wakes the whole intra-op thread pool (one thread per visible core), whose
    idle spinning can exhaust a container's CPU quota and get the launching thread throttled for tens of milliseconds
    (measured on the MI355X boxes: 256 visible cores, a 16-CPU cgroup quota, 50-90 ms stalls after every finetune event)"""
    return np.concatenate([np.asarray(t.detach().numpy(), dtype=np.float32).reshape(-1) for t in tensors]) if tensors \
        else np.zeros(0, np.float32)


def device_params(tensors, device):
    """float32 contiguous device copies of a list of parameter tensors: tensors already on `device` are used as they
    are, host tensors travel in ONE flat upload (per-tensor pageable copies cost ~0.3 ms each)."""
    ts = [t.detach() for t in tensors]
    host = [i for i, t in enumerate(ts) if not t.is_cuda]
    out = [None if not t.is_cuda else t.to(device, F32).contiguous() for t in ts]
    if host:
        flat = torch.from_numpy(host_flat([ts[i] for i in host])).to(device)
        off = 0
        for i in host:
            out[i] = flat[off:off + ts[i].numel()].view(ts[i].shape)
            off += ts[i].numel()
    return out


def pack_conv3x3_device(w, bias, packed, Cin, Cout, transpose=False, scale=None):
    """device fp32 OIHW weights (+bias, + per-output-channel scale) -> `packed` (float32 device buffer of
    scipnp_conv3x3_packed_floats) for conv3x3_c8."""
    co, ci = w.shape[0], w.shape[1]
    _call('scipnp_pack_conv3x3_device_scaled', _p(w, 'w'), _p(bias, 'bias'), _p(scale, 'scale'), _p(packed, 'packed'), ci, co,
          Cin, Cout, int(bool(transpose)), _stream())
    return packed


def bn_fold(gamma, beta, mean, var, eps, scale, shift):
    _call('scipnp_bn_fold', _p(gamma, 'gamma'), _p(beta, 'beta'), _p(mean, 'mean'), _p(var, 'var'), float(eps),
          _p(scale, 'scale'), _p(shift, 'shift'), gamma.numel(), _stream())


def packed_buffer(Cin, Cout, device, split):
    lib = _lib.load()
    if split:
        return torch.empty(lib.scipnp_conv3x3_split_packed_bytes(Cin, Cout), dtype=torch.uint8, device=device)
    return torch.empty(lib.scipnp_conv3x3_packed_floats(Cin, Cout), dtype=F32, device=device)


def c8s_to_c8(x, out=None, scale=1.0):
    n, cg, _two, h, w, _ = x.shape
    if out is None:
        out = torch.empty(n, cg, h, w, 8, device=x.device, dtype=F32)
    _call('scipnp_c8s_to_c8', _p(x, 'x', torch.float16), _p(out, 'out'), float(scale), n, cg * 8, h, w, _stream())
    return out


def c8_scale_to_c8s(x, out=None, scale=1.0):
    n, cg, h, w, _ = x.shape
    if out is None:
        out = torch.empty(n, cg, 2, h, w, 8, device=x.device, dtype=torch.float16)
    _call('scipnp_c8_scale_to_c8s', _p(x, 'x'), _p(out, 'out', torch.float16), float(scale), n, cg * 8, h, w, _stream())
    return out


def c8_add_to_c8s(x, residual_c8s, out=None):
    """c8s(x_fp32_c8 + residual_c8s)"""
    n, cg, h, w, _ = x.shape
    if out is None:
        out = torch.empty(n, cg, 2, h, w, 8, device=x.device, dtype=torch.float16)
    _call('scipnp_c8_add_to_c8s', _p(x, 'x'), _p(residual_c8s, 'residual', torch.float16), _p(out, 'out', torch.float16),
          n, cg * 8, h, w, _stream())
    return out


def c8_to_c8s(x, out=None):
    n, cg, h, w, _ = x.shape
    if out is None:
        out = torch.empty(n, cg, 2, h, w, 8, device=x.device, dtype=torch.float16)
    _call('scipnp_c8_to_c8s', _p(x, 'x'), _p(out, 'out', torch.float16), n, cg * 8, h, w, _stream())
    return out


def ffdnet_pack_input(x, sigma, out_c8=None):
    """planar (n,C,H,W) frames -> FFDNet network input in fp32 c8: replicate pad to even size, pixel-unshuffle, sigma map
    channel (models/network_ffdnet.py:54-64) -- csrc/plugin.hip."""
    n, c, H, W = x.shape
    h, w, cg = (H + 1) // 2, (W + 1) // 2, (4 * c + 1 + 7) // 8
    if out_c8 is None:
        out_c8 = torch.empty(n, cg, h, w, 8, device=x.device, dtype=F32)
    _call('scipnp_ffdnet_pack_input', _p(x, 'x'), float(sigma), _p(out_c8, 'out_c8'), n, c, H, W, _stream())
    return out_c8


def ffdnet_unpack_output(out_c8, C_, H, W, out=None):
    """network output c8 [n][ceil(4C/8)][h][w][8] -> pixel-shuffled, cropped planar (n,C,H,W) (network_ffdnet.py:66-69)"""
    n = out_c8.shape[0]
    if out is None:
        out = torch.empty(n, C_, H, W, device=out_c8.device, dtype=F32)
    _call('scipnp_ffdnet_unpack_output', _p(out_c8, 'out_c8'), _p(out, 'out'), n, C_, H, W, _stream())
    return out


def cube_sum3(cube):
    """(H,W,3,B) -> (H,W,B), sum over the colour axis"""
    H, W, _, B = cube.shape
    out = torch.empty(H, W, B, device=cube.device, dtype=F32)
    _call('scipnp_cube_sum3', _p(cube, 'cube'), _p(out, 'out'), H, W, B, _stream())
    return out


def gray_net_input(x, b, sigma, in_c8):
    B, _, M, N = x.shape
    _call('scipnp_gray_net_input', _p(x, 'x'), _p(b, 'b'), float(np.float32(sigma)), _p(in_c8, 'in_c8'), M, N, B, _stream())
    return in_c8


def gray_net_output(out_c8, theta_raw):
    B, _, M, N = theta_raw.shape
    _call('scipnp_gray_net_output', _p(out_c8, 'out_c8'), _p(theta_raw, 'theta_raw'), M, N, B, _stream())
    return theta_raw


def cube_to_frames(cube):
    """(H,W,B) -> [B][H][W]"""
    H, W, B = cube.shape
    out = torch.empty(B, H, W, device=cube.device, dtype=F32)
    _call('scipnp_cube_to_frames', _p(cube, 'cube'), _p(out, 'frames'), H, W, B, _stream())
    return out


def frames_to_cube(frames):
    """[B][H][W] -> (H,W,B)"""
    B, H, W = frames.shape
    out = torch.empty(H, W, B, device=frames.device, dtype=F32)
    _call('scipnp_frames_to_cube', _p(frames, 'frames'), _p(out, 'cube'), H, W, B, _stream())
    return out


def negate(x, out=None):
    if out is None:
        out = torch.empty_like(x)
    _call('scipnp_negate', _p(x, 'x'), _p(out, 'out'), x.numel(), _stream())
    return out


def fastdvd_noisy_input(v, noise_f64):
    """v + float32(float64(v) + noise): the FastDVDnet finetune's network input (reference test_fastdvdnet.py:359)"""
    out = torch.empty_like(v)
    _call('scipnp_fastdvd_noisy_input', _p(v, 'v'), _p(noise_f64, 'noise', torch.float64), _p(out, 'out'), v.numel(), _stream())
    return out


def sum_rows_f64(part, out=None):
    """row sums of a contiguous fp64 table [rows][n] (or one row [n]) on the device, fixed summation order"""
    t = part.reshape(1, -1) if part.dim() == 1 else part
    if out is None:
        out = torch.empty(t.shape[0], dtype=torch.float64, device=part.device)
    _call('scipnp_sum_rows_f64', _p(t, 'part', torch.float64), _p(out, 'out', torch.float64), t.shape[0], t.shape[1], _stream())
    return out


_SIDE_STREAMS = {}                  # (device, caller's stream) -> its side streams
_SIDE_LOCK = threading.Lock()


def side_stream_count():
    """SCIPNP_STREAMS (default 2): HIP streams a batched network pass is spread over, 1 = everything on the caller's stream"""
    import os
    from . import config
    return config.current().streams


def _side_pool(n):
    dev = torch.cuda.current_device()
    cur = torch.cuda.current_stream(dev)
    key = (dev, cur.cuda_stream)
    with _SIDE_LOCK:                     # harness.py drives several scenes from threads, each on its own stream
        pool = _SIDE_STREAMS.get(key, [])
        if len(pool) < n:
            pool = _SIDE_STREAMS[key] = pool + [torch.cuda.Stream(dev) for _ in range(n - len(pool))]
    return cur, pool


def on_side_streams(n_items, fn):
    """fn(slice) for contiguous chunks of range(n_items), one chunk per side stream; the side streams start behind the
    caller's stream and the caller's stream continues behind all of them.  Independent items (frames of a batched
    network pass) only: grids that do not fill the last generation of workgroups overlap with the other streams' launches."""
    n = min(side_stream_count(), n_items)
    if n <= 1:
        fn(slice(0, n_items))
        return
    cur, pool = _side_pool(n)
    bounds = [round(i * n_items / n) for i in range(n + 1)]
    used = []
    try:
        for st, lo, hi in zip(pool, bounds[:-1], bounds[1:]):
            if hi > lo:
                st.wait_stream(cur)
                used.append(st)
                with torch.cuda.stream(st):
                    fn(slice(lo, hi))
    finally:                             # also after a failed launch: the caller's stream never runs ahead of queued side work
        for st in used:
            cur.wait_stream(st)


def split_overflow(reset=True, word=None):
    """True if the split-fp16 kernels raised the range-guard word since it was last cleared (synchronises).  word: an int32
    device tensor of one element (a solve's own word, see overflow_scope); None = the word the calling thread has bound, or
    the library's process-wide one."""
    flag = C.c_int(0)
    _call('scipnp_read_overflow_word', _p(word, 'word', torch.int32), int(bool(reset)), C.byref(flag), _stream())
    return bool(flag.value)


_OVF_BOUND = threading.local()


class overflow_scope:
    """with overflow_scope(word): every split-fp16 launch of the calling thread raises `word` (an int32 device tensor of one
    element owned by the solve / engine) instead of the process-wide word -- overlapping solves on different host threads
    never see each other's report (include/scipnp.h, scipnp_bind_overflow_word).  Scopes nest; word=None is a no-op."""

    def __init__(self, word):
        self.word = word

    def __enter__(self):
        if self.word is not None:
            self.prev = getattr(_OVF_BOUND, 'word', None)
            _OVF_BOUND.word = self.word
            _call('scipnp_bind_overflow_word', _p(self.word, 'word', torch.int32))
        return self

    def __exit__(self, *exc):
        if self.word is not None:
            _OVF_BOUND.word = self.prev
            _call('scipnp_bind_overflow_word', _p(self.prev, 'word', torch.int32))
        return False


def c8s_to_float(x):
    """c8s (hi, lo') -> fp32 c8 values hi + lo'/2048 (test helper, torch arithmetic)"""
    return x[:, :, 0].float() + x[:, :, 1].float() / 2048.0


def fastdvd_pack_triplets(frames, sigma, out=None, units=1):
    """frames [B*units][3][H][W] (frame t of unit u at t * units + u) -> the circular 3-frame windows of every frame, c8"""
    BU, _, H, W = frames.shape
    if out is None:
        out = torch.empty(BU, 2, H, W, 8, device=frames.device, dtype=F32)
    _call('scipnp_fastdvd_pack_triplets_units', _p(frames, 'frames'), _p(out, 'out'), BU // units, units, H, W,
          float(np.float32(sigma)), _stream())
    return out


def fastdvd_pack_triplets_c8s(frames, sigma, out=None, units=1):
    BU, _, H, W = frames.shape
    if out is None:
        out = torch.empty(BU, 2, 2, H, W, 8, device=frames.device, dtype=torch.float16)
    _call('scipnp_fastdvd_pack_triplets_c8s_units', _p(frames, 'frames'), _p(out, 'out', torch.float16), BU // units, units, H, W,
          float(np.float32(sigma)), _stream())
    return out


def fastdvd_finish(center, x_c8, out=None):
    B, _, H, W = center.shape
    if out is None:
        out = torch.empty_like(center)
    _call('scipnp_fastdvd_finish', _p(center, 'center'), _p(x_c8, 'x_c8'), _p(out, 'out'), B, H, W, _stream())
    return out


# ------------------------------------------------------------------ DDnet glue (csrc/ddnet.hip)
def pm_ddnet_inputs(x, b, coef, planes, mosaic):
    B, _, M, N = x.shape
    _call('scipnp_pm_ddnet_inputs', _p(x, 'x'), _p(b, 'b'), float(np.float32(coef)), _p(planes, 'planes'),
          _p(mosaic, 'mosaic'), M, N, B, _stream())


def ddnet_gather(src, idx, scale, out, C_, h, w):
    """out: fp32 c8 or float16 c8s tensor [E][G][...]; idx int32 [E][3]; scale float32 [E][3][C] or None."""
    E = idx.shape[0]
    split = out.dtype == torch.float16
    _call('scipnp_ddnet_gather', _p(src, 'src'), _p(idx, 'idx', torch.int32), _p(scale, 'scale'),
          _p(None if split else out, 'out'), _p(out if split else None, 'out', torch.float16), E, C_, h, w, _stream())
    return out


def ddnet_finish(src, idx, scale, x_c8, out, C_, Cout, h, w):
    E = x_c8.shape[0]
    _call('scipnp_ddnet_finish', _p(src, 'src'), _p(idx, 'idx', torch.int32), _p(scale, 'scale'), _p(x_c8, 'x_c8'),
          _p(out, 'out'), E, C_, Cout, h, w, _stream())
    return out


def bilinear_up2_c8(planar, out):
    """planar [E][4][h][w] -> out c8 [E][1][2h][2w][8] (float32) or c8s [E][1][2][2h][2w][8] (float16)."""
    E, _, h, w = planar.shape
    split = out.dtype == torch.float16
    _call('scipnp_bilinear_up2_c8', _p(planar, 'planar'), _p(None if split else out, 'out'),
          _p(out if split else None, 'out', torch.float16), E, h, w, _stream())
    return out


def ddnet_mix(branches, gates, out):
    B, _, H, W = out.shape
    _call('scipnp_ddnet_mix', _p(branches, 'branches'), _p(gates, 'gates'), _p(out, 'out'), B, H, W, _stream())
    return out


def to_c8(x):
    """NCHW torch tensor -> c8 layout [n][C/8][h][w][8] (zero-padded channels); test/helper path."""
    n, c, h, w = x.shape
    cp = (c + 7) // 8 * 8
    if cp != c:
        x = torch.cat([x, torch.zeros(n, cp - c, h, w, device=x.device, dtype=x.dtype)], 1)
    return x.reshape(n, cp // 8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous()


def from_c8(x, c=None):
    n, cg, h, w, _ = x.shape
    y = x.permute(0, 1, 4, 2, 3).reshape(n, cg * 8, h, w)
    return y[:, :c].contiguous() if c is not None else y.contiguous()

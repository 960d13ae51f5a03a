"""Denoiser plug-ins with the reference's call signatures, running on the HIP kernels.

ffdnet_rgb_denoise_full_tensor      <- packages/ffdnet/test_ffdnet_ipol.py:240-359
fastdvdnet_denoiser_full_tensor_v2  <- packages/fastdvdnet/test_fastdvdnet.py:325-500
test_ddnet                          <- packages/DDnet/DDnet_test.py:218-321 (deep demosaicking)

Both take / return the reference's (H, W, 3, B) colour cube as a CUDA(ROCm) tensor.  The solver
itself does not go through these wrappers (it keeps everything plane-major and fuses the layout
changes into its pre/post kernels); they exist so that code written against the reference's
plug-in API keeps working.
"""
import torch

from . import ops
from .nets import FFDNetEngine

_ENGINES = {}


def _engine_for(model, B, M, N, device):
    key = (id(model), B, M, N, str(device))
    eng = _ENGINES.get(key)
    if eng is None:
        if len(_ENGINES) > 8:
            _ENGINES.clear()
        eng = _ENGINES[key] = FFDNetEngine(model, B, M, N, device)
    else:
        eng.refresh(model)
    return eng


def _net_input(eng, x, sigma):
    """planar (n,C,H,W) frames -> the engine's own input buffers (c8 fp32, and c8s for the split-fp16 engine): replicate
    pad, pixel-unshuffle and sigma map in one HIP kernel (scipnp_ffdnet_pack_input; reference network_ffdnet.py:54-64)"""
    ops.ffdnet_pack_input(x, sigma, out_c8=eng.in_c8)
    if eng.in_c8s is not None:
        ops.c8_to_c8s(eng.in_c8, out=eng.in_c8s)


def ffdnet_forward_nchw(model, x, sigma):
    """FFDNet forward on n frames: x (n,3,H,W) (colour network) or (n,1,H,W) (ffdnet_gray) CUDA float32, same shape out.
    HIP kernels only: input assembly, the convolution stack, pixel-shuffle + crop (scipnp_ffdnet_unpack_output)."""
    x = x.float().contiguous()
    n, c, H, W = x.shape
    eng = _engine_for(model, n, (H + 1) // 2, (W + 1) // 2, x.device)
    _net_input(eng, x, sigma)
    return ops.ffdnet_unpack_output(eng.forward(), c, H, W)


def ffdnet_rgb_denoise_full_tensor(x, yall, Phiall, sigma, model, useGPU=True, lr_=0.000001, updata_=False,
                                   update_per_iter=4, device=0):
    """x (H,W,3,B) CUDA tensor -> denoised (H,W,3,B); with `updata_` first runs the online
    measurement-loss finetune and returns (out, model) like the reference."""
    rgb = ops.cube_to_rgb(x.float().contiguous())
    if updata_:
        from .finetune import ffdnet_online_finetune
        B, _, H, W = rgb.shape
        eng = _engine_for(model, B, (H + 1) // 2, (W + 1) // 2, x.device)
        _net_input(eng, rgb, sigma)
        ffdnet_online_finetune(model, eng, yall.permute(2, 0, 1).contiguous(), Phiall.permute(2, 3, 0, 1).contiguous(),
                               sigma, lr_, update_per_iter)
        return ops.rgb_to_cube(ops.ffdnet_unpack_output(eng.forward(), 3, H, W)), model
    return ops.rgb_to_cube(ffdnet_forward_nchw(model, rgb, sigma))


def fastdvdnet_denoiser_full_tensor_v2(vnoisy, sigma, y_bayer=None, Phi=None, model=None, useGPU=True, lr_=0.000001,
                                       updata_=False, update_per_iter=1, gray=False, update_times=-1):
    """vnoisy (H,W,3,B) CUDA tensor -> denoised (H,W,3,B); with `updata_` first runs the online finetune on the
    measurement loss (y_bayer / Phi as the reference's Bayer planes (M,N,4) / (M,N,B,4)) and returns (out, model)."""
    from .fastdvd import FastDVDEngine
    if gray:
        raise NotImplementedError('grayscale FastDVDnet is outside the Bayer hot path')
    H, W, _, B = vnoisy.shape
    frames = ops.cube_to_rgb(vnoisy.float().contiguous())
    eng = FastDVDEngine(model, B, H, W, vnoisy.device)
    if updata_:
        from .finetune import fastdvdnet_online_finetune
        y_pm = y_bayer.permute(2, 0, 1).contiguous()
        Phi_pm = Phi.permute(2, 3, 0, 1).contiguous()
        fastdvdnet_online_finetune(model, eng, frames, y_pm, Phi_pm, sigma, lr_, update_per_iter)
        return ops.rgb_to_cube(eng.forward(frames, sigma)), model
    return ops.rgb_to_cube(eng.forward(frames, sigma))


def test_ddnet(vnoisy, yall=None, Phiall=None, model=None, useGPU=True, args=None, gray=False):
    """Deep demosaicking plug-in: vnoisy (H,W,3,B) CUDA tensor holding the mosaic at its CFA sites (`oneCh2ThreeCh`)
    -> demosaicked (H,W,3,B).  The network only sees the channel sum (network_demosaicking.py:425-429), i.e. the
    mosaic.  `args` (an object with dm_update, dm_lr, dm_update_per_iter; the solver never passes one): with dm_update the
    demosaicker is first finetuned online on MSE(input, CFA samples of its output) -- DDnet_test.py:248-296, a new Adam per
    step, `model` updated in place (ddnet_train.ddnet_online_finetune) -- and (out, model) is returned like the reference."""
    from .ddnet import DDnetEngine
    if gray:
        raise NotImplementedError('grayscale DDnet is outside the Bayer hot path')
    H, W, _, B = vnoisy.shape
    mosaic3 = ops.cube_sum3(vnoisy.float().contiguous())                   # (H,W,B): one non-zero term per pixel
    planes = ops.mosaic_to_state(mosaic3)
    mosaic = torch.empty(B, H, W, dtype=torch.float32, device=vnoisy.device)
    ops.pm_ddnet_inputs(planes, None, 0.0, torch.empty_like(planes), mosaic)
    eng = DDnetEngine(model, B, H, W, vnoisy.device)
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=vnoisy.device)
    if args is not None and getattr(args, 'dm_update', False):
        from .ddnet_train import ddnet_online_finetune
        ddnet_online_finetune(model, eng, planes, mosaic, args.dm_lr, args.dm_update_per_iter)
        return ops.rgb_to_cube(eng.forward(planes, mosaic, out)), model
    return ops.rgb_to_cube(eng.forward(planes, mosaic, out))


test_ddnet.__test__ = False        # not a pytest test (the name is the reference's)

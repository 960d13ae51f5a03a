"""Oracle (test infrastructure): the denoiser plug-in wrappers the solver calls once per ADMM
iteration, with their online measurement-loss finetune branches, on PyTorch-CPU.

ffdnet_pass      <- packages/ffdnet/test_ffdnet_ipol.py:240-359  (ffdnet_rgb_denoise_full_tensor)
fastdvdnet_pass  <- packages/fastdvdnet/test_fastdvdnet.py:325-500 (fastdvdnet_denoiser_full_tensor_v2)
                    + packages/fastdvdnet/fastdvdnet.py:82-146     (fastdvdnet_seqdenoise)
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import sci_ops as ops

NUM_IN_FR_EXT = 5  # temporal window (reference test_fastdvdnet.py module constant)


def _ffdnet_frames(x, sigma, model):
    """Frame-by-frame FFDNet forward (reference :340-354 / :256-264): x (H,W,3,B) -> same."""
    out = torch.zeros(x.shape)
    for t in range(x.shape[3]):
        frame = x[:, :, :, t].permute(2, 0, 1).float().unsqueeze(0)
        sig = torch.full((1, 1, 1, 1), sigma).type_as(frame)
        out[:, :, :, t] = model(frame, sig).permute(2, 3, 1, 0).squeeze(3)
    return out


def ffdnet_pass(x, yall, Phiall, sigma, model, lr=1e-6, update=False, update_per_iter=4, trace=None):
    """Returns denoised (H,W,3,B); with `update` also mutates `model` by `update_per_iter` Adam
    steps on  MSE( sum_t Phi * bayer_sample(model(x)) , y )  (reference :248-334) and returns
    (out, model).  The forwards are NOT under no_grad in the reference (:340-354); values are the same."""
    if update:
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        mse = nn.MSELoss()
        for _ in range(update_per_iter):
            den = _ffdnet_frames(x, sigma, model)
            planes = ops.rgb_to_bayer_planes(den)
            loss = mse(torch.sum(planes * Phiall, dim=2), yall)
            opt.zero_grad()
            loss.backward()
            opt.step()
            if trace is not None:
                trace.append(float(loss))
        model.eval()
        out = _ffdnet_frames(x, sigma, model)
        planes = ops.rgb_to_bayer_planes(out)
        final = mse(torch.sum(planes * Phiall, dim=2), yall)
        if trace is not None:
            trace.append(float(final))
        return out, model
    return _ffdnet_frames(x, sigma, model)


def fastdvdnet_seq(seq, noise_std, model, windsize=NUM_IN_FR_EXT):
    """Sliding 5-frame window with CIRCULAR temporal indexing, reflect-pad H,W to multiples of 4
    (reference fastdvdnet.py:106-146).  seq: (N,C,H,W)."""
    N, C, H, W = seq.shape
    hw = (windsize - 1) // 2
    out = torch.empty((N, C, H, W))
    noise_map = noise_std.expand((1, 1, H, W))
    wpad, hpad = (-W) % 4, (-H) % 4
    for n in range(N):
        idx = (torch.arange(n, n + windsize) - hw) % N
        win = seq[idx].reshape((1, -1, H, W))
        win = F.pad(win, (0, wpad, 0, hpad), mode='reflect')
        nm = F.pad(noise_map, (0, wpad, 0, hpad), mode='reflect')
        den = model(win, nm)
        out[n] = den[:, :, :H, :W]
    return out


def _rgb_cube_to_mosaic(rgb):
    """gen_bayer_img(.,1): sum_c RGB*CFA-mask -> (H,W,B)  (reference packages/fastdvdnet/utils.py:69-78)."""
    R, G, B = ops.cfa_masks((rgb.shape[0], rgb.shape[1]))
    mask = torch.cat([R.unsqueeze(2), G.unsqueeze(2), B.unsqueeze(2)], dim=2)
    mask = torch.repeat_interleave(mask.unsqueeze(3), rgb.shape[3], dim=3)
    return torch.sum(rgb * mask, dim=2)


# Test-infrastructure hook (tools/make_golden.py): when a dict is installed here, the first finetune step of
# fastdvdnet_pass also evaluates a float64 copy of the model on the very same inputs and leaves its gradients
# {parameter name: float64 tensor} in the dict -- the yardstick for "how well is this gradient determined in fp32"
# (the reference's own fp32 .grad deviates from it by 0.4 - 1.5e-4 on most layers: ReLU masks flip where an activation
# is within round-off of zero).  None (the default): nothing extra is computed.
GRAD64_SINK = None


def _fastdvd_grad_f64(model, v_plus, noise_map, Phi_mosaic, y_mosaic):
    import copy
    m64 = copy.deepcopy(model).double()
    m64.train()
    for m in m64.module.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
    for p in m64.parameters():
        p.grad = None
    N, C, H, W = v_plus.shape
    vp, nm = v_plus.double(), noise_map.double()
    den = torch.empty((N, C, H, W), dtype=torch.float64)
    for n in range(N):
        idx = (torch.arange(n, n + NUM_IN_FR_EXT) - 2) % N
        den[n] = m64(vp[idx].reshape((1, -1, H, W)), nm)
    den = den.permute(2, 3, 1, 0)
    loss = nn.MSELoss()(torch.sum(_rgb_cube_to_mosaic(den) * Phi_mosaic.double(), dim=2), y_mosaic.double())
    loss.backward()
    return {k: p.grad.detach().clone() for k, p in m64.named_parameters() if p.grad is not None}


def fastdvdnet_pass(vnoisy, sigma, y_planes=None, Phi_planes=None, model=None, lr=1e-6, update=False,
                    update_per_iter=1, trace=None, noise=None):
    """vnoisy (H,W,3,B).  `model` must expose `.module` when `update` (the reference dereferences the
    DataParallel wrapper, test_fastdvdnet.py:377).  Finetune input quirk kept: the helper
    add_gaussian_noise_meas_cuda already returns input+noise, so the net sees 2*v + N(0,(5/255)^2)
    with the noise drawn from the GLOBAL NumPy RNG in float64 (utils/utils_image.py:183-192,
    test_fastdvdnet.py:359).  `noise`, if given, replaces the NumPy draw (same shape as (B,3,H,W))."""
    noisestd = torch.FloatTensor([sigma])
    if not update:
        model.eval()
        with torch.no_grad():
            out = fastdvdnet_seq(vnoisy.permute(3, 2, 0, 1), noisestd, model)
        return out.permute(2, 3, 1, 0)

    steps = [update_per_iter] if isinstance(update_per_iter, int) else list(update_per_iter)
    lrs = [lr] if isinstance(update_per_iter, int) else list(lr)
    mse = nn.MSELoss()
    v = vnoisy.permute(3, 2, 0, 1)
    v_np = v.detach().cpu().numpy()
    if noise is None:
        noise = np.random.normal(0, 5 / 255, v_np.shape)
    v_plus = v + torch.from_numpy(v_np + noise).float()
    Phi_mosaic = ops.bayer_merge(Phi_planes)
    y_mosaic = ops.bayer_merge(y_planes)
    model.train()
    for m in model.module.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
    N, C, H, W = v.shape
    noise_map = noisestd.expand((1, 1, H, W))
    if isinstance(GRAD64_SINK, dict) and not GRAD64_SINK:
        GRAD64_SINK.update(_fastdvd_grad_f64(model, v_plus, noise_map, Phi_mosaic, y_mosaic))
    for n_steps, lr_i in zip(steps, lrs):
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=lr_i)
        for _ in range(n_steps):
            den = torch.empty((N, C, H, W))
            for n in range(N):
                idx = (torch.arange(n, n + NUM_IN_FR_EXT) - 2) % N
                den[n] = model(v_plus[idx].reshape((1, -1, H, W)), noise_map)
            den = den.permute(2, 3, 1, 0)
            loss = mse(torch.sum(_rgb_cube_to_mosaic(den) * Phi_mosaic, dim=2), y_mosaic)
            opt.zero_grad()
            loss.backward()
            opt.step()
            if trace is not None:
                trace.append(float(loss))
    with torch.no_grad():
        out = fastdvdnet_seq(v.detach(), noisestd, model)
    out = out.permute(2, 3, 1, 0)
    if trace is not None:
        planes = ops.bayer_split(_rgb_cube_to_mosaic(out))
        trace.append(float(mse(torch.sum(planes * Phi_planes, dim=2), y_planes)))
    return out, model


def _ddnet_seq(seq, model):
    """packages/DDnet/DDnet_test.py:166-206 (ddnet_seqdenoise): sliding 5-frame window with circular temporal indexing,
    reflect-pad to multiples of 4, one network call per frame"""
    N, C, H, W = seq.shape
    out = torch.empty((N, C, H, W))
    wpad, hpad = (-W) % 4, (-H) % 4
    for n in range(N):
        idx = (torch.arange(n, n + NUM_IN_FR_EXT) - 2) % N
        win = F.pad(seq[idx].reshape((1, -1, H, W)), (0, wpad, 0, hpad), mode='reflect')
        out[n] = model(win)[:, :, :H, :W]
    return out


def bayer_sites(rgb_video):
    """packages/DDnet/DDnet_test.py:208-216 (gen_bayer_img): (H,W,3,F) -> the CFA samples of every frame at their sites,
    zeros elsewhere, as (F,3,H,W)"""
    bayer = torch.zeros_like(rgb_video)
    bayer[0::2, 0::2, 0, :] = rgb_video[0::2, 0::2, 0, :]
    bayer[0::2, 1::2, 1, :] = rgb_video[0::2, 1::2, 1, :]
    bayer[1::2, 0::2, 1, :] = rgb_video[1::2, 0::2, 1, :]
    bayer[1::2, 1::2, 2, :] = rgb_video[1::2, 1::2, 2, :]
    return bayer.permute(3, 2, 0, 1)


def ddnet_pass(x_bayer_3ch, model, dm_update=False, dm_lr=1e-6, dm_update_per_iter=1, trace=None):
    """Deep demosaicking of a (H,W,3,B) CFA-site cube (`oneCh2ThreeCh` of the mosaic): sliding 5-frame window with
    circular temporal indexing, reflect-pad to multiples of 4.  reference packages/DDnet/DDnet_test.py:166-216
    (ddnet_seqdenoise), :218-321 (test_ddnet).

    dm_update (the `args.dm_update` branch, :248-296; the solver never passes `args`, a caller of the plug-in may):
    `dm_update_per_iter` steps of { all frames through the network (train mode: no BatchNorm or dropout in DDnet, so the
    same function), loss = MSE(input cube, CFA samples of the output), a NEW Adam(model.parameters(), lr=dm_lr) every step,
    backward, step }, then the pass itself without gradients.  Returns (out, model) then, like the reference.
    trace: list receiving the loss of every step."""
    seq = x_bayer_3ch.permute(3, 2, 0, 1)
    if dm_update:
        mse = torch.nn.MSELoss()
        model.train()
        for _ in range(dm_update_per_iter):
            outv = _ddnet_seq(seq, model).permute(2, 3, 1, 0)
            total_loss = mse(seq, bayer_sites(outv))
            optimizer = torch.optim.Adam(model.parameters(), lr=dm_lr)
            optimizer.zero_grad()
            total_loss.backward()
            optimizer.step()
            if trace is not None:
                trace.append(float(total_loss))
        with torch.no_grad():
            out = _ddnet_seq(seq, model)
        return out.permute(2, 3, 1, 0), model
    model.eval()
    with torch.no_grad():
        out = _ddnet_seq(seq, model)
    return out.permute(2, 3, 1, 0)

"""Scene harness: the reference's three driver scripts as a library + CLI.

  run_tv_warm_start   <- ADMM_TV_Warm_Start_save.py:36-178          (ADMM-TV, 40 iterations, saves the warm start)
  run_two_stage       <- two_stage_ADMM_Online_FFD_Warm.py:62-330   (FFDNet)  and
                         two_stage_ADMM_Online_FastDVD_Warm.py:60-345 (FastDVDnet): per-measurement loop, /255 scaling,
                         warm start from the TV result, per-scene sigma / iteration schedules, model carry-over policy
                         (`reuse_model`), log text, result .mat with the reference's variable names.

On-disk formats
  scene   : MATLAB v7.3 (HDF5) file with `meas_bayer`, `mask_bayer`, `orig_bayer`, `orig` -- h5py returns MATLAB arrays
            with reversed axes, hence the reference's transpose((2,1,0)) (:189-195), reproduced here.  h5py is an optional
            dependency (absent from the build image): without it v7.3 files raise a clear error; MATLAB <= v7.2 files
            (scipy.io, axes already in MATLAB order) and .npz files with the same variable names are always readable.
  warm    : scipy.io .mat with `v_Admm_tv_denoise (H, W, nmask*nmea)`, `psnr_Admm_tv_denoise`, `ssim_Admm_tv_denoise`
            (ADMM_TV_Warm_Start_save.py:174-178).
  results : scipy.io .mat with `v_twoStageAdmm_{ffd,fastdvd}_gray_bayer`, `psnr_*_gray`, `ssim_*_gray`, `psnr_all_iter`,
            `orig_real`, `meas_bayer` (two_stage_...FFD_Warm.py:320-330).

Measurements of one scene are independent problems unless the finetuned denoiser is carried over
(`update and reuse_model`, :270-271); `run_two_stage(..., shard=True)` distributes them over the ranks of the default
process group (one RCCL gather at the end, adaptivepnp_sci_amd/shard.py) and refuses the carry-over combination.
"""
import argparse
import copy
import os
import time
from statistics import mean

import numpy as np
import torch

from . import _lib, shard as _shard
from ._lib import usable_cpus      # noqa: F401  (re-exported: drivers cap torch.set_num_threads() with it)
from .solver import admm_denoise_bayer_demosaic_pre, twoStageAdmm_denoise_bayer

MAXB = 255.

SCENES = ('Beauty_bayer', 'Bosphorus_bayer', 'Jockey_bayer', 'Runner_bayer', 'ShakeNDry_bayer', 'Traffic_bayer')


def _s(sig255, iters, lr, upi, interval, update_times=-1, deep=None):
    return dict(sigma=[v / 255 for v in sig255], iter_max=list(iters), lr=lr, update_per_iter=upi,
                interval_iter=interval, update_times=update_times, deep=deep or {})


def _d(sig255=None, iters=None, interval=None):
    o = {}
    if sig255 is not None:
        o['sigma'] = [v / 255 for v in sig255]
    if iters is not None:
        o['iter_max'] = list(iters)
    if interval is not None:
        o['interval_iter'] = interval
    return o


# per-scene schedules of the reference drivers (values only; `deep` = overrides when deep demosaicking is on)
SCHEDULES = {
    # two_stage_ADMM_Online_FFD_Warm.py:62-151
    'ffdnet_color': {
        'Beauty_bayer': _s([25, 12, 6], [15, 6, 4], 2e-6, 2, 15, deep=_d([25, 12, 6], [6, 6, 4], 6)),
        'Bosphorus_bayer': _s([50, 25, 12, 6], [8, 4, 4, 4], 2e-6, 2, 8, deep=_d([25, 12, 6], [4, 4, 2], 8)),
        'Jockey_bayer': _s([25, 12, 6], [16, 8, 4], 2e-6, 2, 16, deep=_d([12, 6], [16, 8])),
        'Runner_bayer': _s([50, 25, 12, 6], [8, 4, 4, 4], 2e-6, 2, 8, deep=_d([25, 12, 6], [8, 8, 4], 10)),
        'ShakeNDry_bayer': _s([50, 25, 12, 6], [8, 4, 4, 4], 2e-6, 2, 10, deep=_d([25, 12, 6], [8, 8, 4])),
        'Traffic_bayer': _s([50, 25], [16, 8], 2e-6, 2, 16, deep=_d([25, 12], [14, 7], 14)),
    },
    # two_stage_ADMM_Online_FastDVD_Warm.py:66-170
    'fastdvd_color': {
        'Beauty_bayer': _s([8], [18], 2e-6, 2, 9, 1, deep=_d([12, 6], [21, 2], 22)),
        'Bosphorus_bayer': _s([12, 6], [24, 12], 2e-7, 2, 12, -1, deep=_d([8, 6], [24, 12], 25)),
        'Jockey_bayer': _s([12], [24], 2e-7, 2, 12, -1, deep=_d([12, 6], [24, 6], 25)),
        'Runner_bayer': _s([14], [24], 2e-7, 2, 12, -1, deep=_d([12, 6], [40, 15], 41)),
        'ShakeNDry_bayer': _s([10], [15], 2e-7, 1, 7, -1, deep=_d([12, 6], [14, 4], 15)),
        'Traffic_bayer': _s([30], [22], 2e-7, 2, 11, -1, deep=_d([25, 12, 6], [36, 6, 2], 43)),
    },
}
TV_SCHEDULE = dict(sigma=[0 / 255], iter_max=[40])          # ADMM_TV_Warm_Start_save.py:36-37


def schedule_for(denoiser, scene_name, deep_demosaicking=False):
    """sigma / iter_max / lr / update_per_iter / interval_iter / update_times of the reference driver for a scene."""
    base = SCHEDULES[denoiser].get(scene_name)
    if base is None:
        raise KeyError(f'no reference schedule for scene {scene_name!r}; pass an explicit schedule')
    s = {k: copy.copy(v) for k, v in base.items() if k != 'deep'}
    if deep_demosaicking:
        s.update(copy.deepcopy(base['deep']))
    return s


# ------------------------------------------------------------------------------------------------ on-disk formats
class Scene:
    """meas (H,W,nmea), mask (H,W,nmask), orig_bayer (H,W,nmask*nmea) float32 in [0,255] units; orig_real as stored."""

    def __init__(self, name, meas, mask, orig_bayer, orig_real=None):
        meas = np.float32(meas)
        if meas.ndim == 2:
            meas = meas[:, :, None]
        self.name, self.meas, self.mask = name, meas, np.float32(mask)
        self.orig_bayer = None if orig_bayer is None else np.float32(orig_bayer)
        self.orig_real = orig_real
        H, W, self.nmea = self.meas.shape
        self.nmask = self.mask.shape[2]
        if self.mask.shape[:2] != (H, W):
            raise ValueError(f'mask {self.mask.shape} does not match meas {self.meas.shape}')
        if self.orig_bayer is not None and self.orig_bayer.shape != (H, W, self.nmask * self.nmea):
            raise ValueError(f'orig_bayer {self.orig_bayer.shape} != {(H, W, self.nmask * self.nmea)}')

    def measurement(self, i):
        """(y (H,W), orig (H,W,nmask) or None) of measurement i, scaled to [0,1] units (reference :244-248)."""
        y = self.meas[:, :, i] / MAXB
        o = None if self.orig_bayer is None else self.orig_bayer[:, :, i * self.nmask:(i + 1) * self.nmask] / MAXB
        return np.ascontiguousarray(y, np.float32), None if o is None else np.ascontiguousarray(o, np.float32)


def _is_hdf5(path):
    sig = b'\x89HDF\r\n\x1a\n'
    with open(path, 'rb') as f:               # MATLAB v7.3 puts a 512-byte text header in front of the HDF5 superblock
        head = f.read(520)
    return head[:8] == sig or head[512:520] == sig


def load_scene(path):
    """Read a scene file: MATLAB v7.3 (HDF5: h5py if installed, else the built-in minimal reader), MATLAB <= v7.2
    (scipy.io) or .npz."""
    name = os.path.splitext(os.path.basename(path))[0]
    if path.endswith('.npz'):
        d = np.load(path)
        return Scene(name, d['meas_bayer'], d['mask_bayer'], d['orig_bayer'] if 'orig_bayer' in d else None,
                     d['orig'] if 'orig' in d else None)
    if _is_hdf5(path):
        want = ('meas_bayer', 'mask_bayer', 'orig_bayer', 'orig')
        try:
            import h5py
        except ImportError:
            h5py = None
        if h5py is not None:
            with h5py.File(path, 'r') as f:
                d = {k: np.array(f[k]) for k in want if k in f}
        else:
            # no h5py here: the package's own reader of the HDF5 subset MATLAB writes (hdf5_min.py) -- same arrays,
            # same (reversed) axis order as h5py returns
            from .hdf5_min import Hdf5Unsupported, read_mat73
            try:
                d = read_mat73(path, want)
            except Hdf5Unsupported as e:
                raise RuntimeError(f'{path}: MATLAB v7.3 (HDF5) file outside what the built-in reader handles ({e}); '
                                   'install h5py or convert the file to MATLAB v7 / .npz') from e
        if 'meas_bayer' not in d or 'mask_bayer' not in d:
            raise RuntimeError(f'{path}: no meas_bayer / mask_bayer variables')
        meas, mask, orig, real = d['meas_bayer'], d['mask_bayer'], d.get('orig_bayer'), d.get('orig')
        # h5py hands MATLAB arrays over with reversed axes (reference :189-195)
        mask = np.float32(mask).transpose((2, 1, 0))
        meas = np.float32(meas).transpose((1, 0)) if meas.ndim < 3 else np.float32(meas).transpose((2, 1, 0))
        orig = None if orig is None else np.float32(orig).transpose((2, 1, 0))
        return Scene(name, meas, mask, orig, real)
    import scipy.io as sio
    d = sio.loadmat(path)
    return Scene(name, d['meas_bayer'], d['mask_bayer'], d.get('orig_bayer'), d.get('orig'))


def save_warm_start(path, v, psnr, ssim):
    import scipy.io as sio
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    sio.savemat(path, {'v_Admm_tv_denoise': v, 'psnr_Admm_tv_denoise': psnr, 'ssim_Admm_tv_denoise': ssim})


def load_warm_start(path):
    import scipy.io as sio
    return np.float32(sio.loadmat(path)['v_Admm_tv_denoise'])


def warm_start_path(results_dir, scene):
    """'./results/savedmat/_Admm_tv_<scene><nmask>.mat' (ADMM_TV_Warm_Start_save.py:174, two_stage_...:168)"""
    return os.path.join(results_dir, 'savedmat', '_Admm_tv_{}{:d}.mat'.format(scene.name, scene.nmask))


class _Tee:
    def __init__(self, logf, echo):
        self.logf, self.echo = logf, echo

    def write(self, s):
        if self.logf is not None:
            self.logf.write(s)

    def say(self, s):
        if self.echo:
            print(s)
        self.write(s + ' \n')


# ------------------------------------------------------------------------------------------------ drivers
def _tv_warm_start_batched(scene, sch, log, v, psnr, ssim, psnr_all):
    """the measurements of a scene as ONE unit batch (solver.AdmmRun(units=nmea)): they are independent ADMM-TV problems on the same
    masks, so one launch sequence steps all of them (about 3x the per-measurement throughput at 256 x 256 x 8);
    results bit-identical to the loop, the log text the loop would have written (measurement after measurement), the running
    time reported per measurement = the batch's time / nmea"""
    from . import ops
    from .solver import AdmmRun, _as_lists, iteration_log_line
    H, W, nmea = scene.meas.shape
    nmask = scene.nmask
    pairs = [scene.measurement(i) for i in range(nmea)]
    have = pairs[0][1] is not None
    sigma, iter_max = _as_lists(sch['sigma'], sch['iter_max'])
    t0 = time.time()
    run = AdmmRun([p[0] for p in pairs], [scene.mask] * nmea, 'tv', False, X_orig=[p[1] for p in pairs] if have else None,
                  _lambda=1, gamma=0.01, units=nmea)
    nsigs = []
    for nsig, iters in zip(sigma, iter_max):
        for _ in range(iters):
            run.step(nsig)
            nsigs.append(nsig)
    mosaics, pall, reports = run.result_mosaic(), run.psnr_all(), (run.final_report() if have else None)
    xs = [ops.to_host(m) for m in mosaics]
    dt = (time.time() - t0) / nmea
    for i in range(nmea):
        log.write('Measurement Frame {}.\n'.format(i))
        log.write('tv_denoiser start.\n')
        for k, val in enumerate(pall[i]):
            if (k + 1) % 2 == 0:
                line, tail = iteration_log_line('TV', k, nsigs[k], val, False)
                print(line)                                  # (as the solver does while it runs)
                log.write(line + tail)
        sl = slice(i * nmask, (i + 1) * nmask)
        v[:, :, sl] = xs[i]
        if have:
            p, s_ = reports[i]
            psnr[sl, 0], ssim[sl, 0] = p, s_
            log.say('ADMM-{} PSNR {:2.2f} dB, SSIM {:.4f}, running time {:.1f} seconds.'.format('TV', mean(p), mean(s_), dt))
        psnr_all.append(pall[i])


def run_tv_warm_start(scene, logf=None, schedule=None, save_to=None, echo=True, batch=False):
    """ADMM-TV on every measurement of a scene (ADMM_TV_Warm_Start_save.py:112-178).  Returns
    dict(v (H,W,nmask*nmea) in [0,1] units, psnr, ssim (nmask*nmea,1), psnr_all per measurement).
    batch=True: the measurements are stepped together as one unit batch (same results and log text, see _tv_warm_start_batched);
    needs H*W to be a multiple of 2048 when the scene carries ground truth (per-iteration PSNR of every unit)."""
    sch = schedule or TV_SCHEDULE
    H, W, nmea = scene.meas.shape
    nmask = scene.nmask
    log = _Tee(logf, echo)
    v = np.zeros([H, W, nmask * nmea], np.float32)
    psnr = np.zeros([nmask * nmea, 1], np.float32)
    ssim = np.zeros([nmask * nmea, 1], np.float32)
    psnr_all = []
    log.write(scene.name + ':\n')
    log.write('tv_denoiser start...\n')
    if batch and nmea > 1:
        _tv_warm_start_batched(scene, sch, log, v, psnr, ssim, psnr_all)
        if save_to:
            save_warm_start(save_to, v, psnr, ssim)
        return dict(v=v, psnr=psnr, ssim=ssim, psnr_all=psnr_all)
    for i in range(nmea):
        log.write('Measurement Frame {}.\n'.format(i))
        y, orig = scene.measurement(i)
        log.write('tv_denoiser start.\n')
        t0 = time.time()
        x, p, s, pall = admm_denoise_bayer_demosaic_pre(y, scene.mask, 1, 0.01, 'tv', sch['iter_max'], False, sch['sigma'],
                                                        x0_bayer=None, X_orig=orig, model=None, show_iqa=True, logf=log)
        dt = time.time() - t0
        sl = slice(i * nmask, (i + 1) * nmask)
        v[:, :, sl] = x
        if orig is not None:
            psnr[sl, 0], ssim[sl, 0] = p, s
            log.say('ADMM-{} PSNR {:2.2f} dB, SSIM {:.4f}, running time {:.1f} seconds.'.format('TV', mean(p), mean(s), dt))
        psnr_all.append(pall)
    if save_to:
        save_warm_start(save_to, v, psnr, ssim)
    return dict(v=v, psnr=psnr, ssim=ssim, psnr_all=psnr_all)


_SHORT = {'ffdnet_color': 'ffd', 'fastdvd_color': 'fastdvd'}
_START = {'ffdnet_color': 'FFDnet-rgb-demosaic start.\n', 'fastdvd_color': 'fastdvdnet-rgb-demosaic start.\n'}


def run_two_stage(scene, warm, denoiser, model_denoise, model_demosaic=None, schedule=None, update=True,
                  reuse_model=True, logf=None, save_dir=None, echo=True, shard=False):
    """Two-stage PnP-ADMM on every measurement of a scene, warm-started from the ADMM-TV result `warm`
    (H,W,nmask*nmea) (two_stage_ADMM_Online_FFD_Warm.py:241-330 / ..._FastDVD_Warm.py:262-345).

    Model carry-over (`:270-275`): with `update and reuse_model` the finetuned denoiser of measurement i seeds
    measurement i+1; otherwise every measurement starts from the weights `model_denoise` had at entry.
    Returns dict(v, rgb (nmea,H,W,3,nmask), psnr, ssim, psnr_all, model)."""
    if denoiser not in _SHORT:
        raise ValueError('Unsupported denoiser {}!'.format(denoiser))
    sch = schedule or schedule_for(denoiser, scene.name, model_demosaic is not None)
    H, W, nmea = scene.meas.shape
    nmask = scene.nmask
    carry = bool(update and reuse_model)
    if shard and carry and nmea > 1:
        raise ValueError('measurements are not independent when the finetuned model is carried over '
                         '(update and reuse_model): run unsharded or pass reuse_model=False')
    log = _Tee(logf, echo)
    pristine = copy.deepcopy(model_denoise.state_dict())
    extra = {'update_times': sch['update_times']} if denoiser == 'fastdvd_color' else {}

    def solve(i, model):
        y, orig = scene.measurement(i)
        v_tv = np.ascontiguousarray(warm[:, :, i * nmask:(i + 1) * nmask], np.float32)
        return twoStageAdmm_denoise_bayer(y, scene.mask, 1, 0.01, denoiser, sch['iter_max'], False, sch['sigma'],
                                          x0_bayer=v_tv, X_orig=orig, model_denoise=model, model_demosaic=model_demosaic,
                                          show_iqa=True, demosaic_method='malvar2004', lr_=sch['lr'],
                                          interval_iter=sch['interval_iter'], logf=log, update_=update,
                                          update_per_iter=sch['update_per_iter'], **extra)

    v = np.zeros([H, W, nmask * nmea], np.float32)
    rgb = np.zeros([nmea, H, W, 3, nmask], np.float32)
    psnr = np.zeros([nmask * nmea, 1], np.float32)
    ssim = np.zeros([nmask * nmea, 1], np.float32)
    psnr_all = [None] * nmea
    log.write(scene.name + ':\n')

    def record(i, res, dt):
        sl = slice(i * nmask, (i + 1) * nmask)
        rgb[i], v[:, :, sl], psnr_all[i] = res[0], res[1], res[4]
        if scene.orig_bayer is not None:
            psnr[sl, 0], ssim[sl, 0] = res[2], res[3]
            log.say('ADMM-{}--{}-{} PSNR {:2.2f} dB, SSIM {:.4f}, running time {:.1f} seconds.'.format(
                denoiser.upper(), scene.name, i, mean(res[2]), mean(res[3]), dt))

    if shard:
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        dev = torch.device('cuda', torch.cuda.current_device())
        local, local_rgb, stats = {}, {}, {}
        mine = _shard.partition(nmea, world, rank)

        def one(i):
            model = copy.deepcopy(model_denoise)             # every unit starts from the pristine weights
            model.load_state_dict(pristine, strict=True)
            t0 = time.time()
            res = solve(i, model)
            return i, res, time.time() - t0

        # this rank's measurements on two host threads / HIP streams: the second stream covers the host-side steps
        # (uploads, engine construction, read-backs, finetune events) of the first
        import concurrent.futures as cf
        cur = torch.cuda.current_stream()
        # (the FastDVDnet finetune draws noise from the process-global NumPy generator: keep its units in order)
        n_streams = 1 if (denoiser == 'fastdvd_color' and update) else 2
        pool_streams = [torch.cuda.Stream() for _ in range(min(n_streams, max(1, len(mine))))]

        def worker(j):
            out = []
            with torch.cuda.stream(pool_streams[j]):
                for i in mine[j::len(pool_streams)]:
                    out.append(one(i))
            return out

        for st_ in pool_streams:
            st_.wait_stream(cur)
        with cf.ThreadPoolExecutor(len(pool_streams)) as ex:
            for part in ex.map(worker, range(len(pool_streams))):
                for i, res, dt in part:
                    stats[i] = (res, dt)
                    local[i] = torch.from_numpy(res[1]).to(dev)
                    local_rgb[i] = torch.from_numpy(res[0]).to(dev)
        for st_ in pool_streams:
            cur.wait_stream(st_)
        # ONE collective for the job: mosaic and colour cube of a unit travel in one (H, W, 4, nmask) slab
        slabs = {i: torch.cat([local_rgb[i], local[i][:, :, None, :]], 2).contiguous() for i in local}
        got = _shard.gather_units(slabs, nmea, (H, W, 4, nmask), dev)
        if got is not None:
            for i, slab in enumerate(got):
                rgb[i], v[:, :, i * nmask:(i + 1) * nmask] = slab[:, :, :3].cpu().numpy(), slab[:, :, 3].cpu().numpy()
        for i, (res, dt) in stats.items():          # PSNR / SSIM lines of this rank's units only
            sl = slice(i * nmask, (i + 1) * nmask)
            psnr_all[i] = res[4]
            if scene.orig_bayer is not None:
                psnr[sl, 0], ssim[sl, 0] = res[2], res[3]
    else:
        for i in range(nmea):
            log.write('Measurement Frame {}.\n'.format(i))
            log.write(_START[denoiser])
            t0 = time.time()
            res = solve(i, model_denoise)
            if carry:
                model_denoise = res[5]
            else:
                model_denoise.load_state_dict(pristine, strict=True)
            record(i, res, time.time() - t0)

    out = dict(v=v, rgb=rgb, psnr=psnr, ssim=ssim, psnr_all=psnr_all, model=model_denoise, schedule=sch)
    if save_dir:
        import scipy.io as sio
        os.makedirs(os.path.join(save_dir, 'savedmat'), exist_ok=True)
        short = _SHORT[denoiser]
        path = os.path.join(save_dir, 'savedmat', 'twoStageAdmm_{}_{}{:d}_sigma{:d}_all7_log.mat'.format(
            denoiser.lower(), scene.name, nmask, int(sch['sigma'][-1] * MAXB)))
        sio.savemat(path, {f'v_twoStageAdmm_{short}_gray_bayer': v, f'psnr_{short}_gray': psnr, f'ssim_{short}_gray': ssim,
                           'psnr_all_iter': [p for p in psnr_all if p is not None],
                           'orig_real': np.zeros(0) if scene.orig_real is None else scene.orig_real,
                           'meas_bayer': scene.meas})
        out['saved'] = path
    return out


def worker_init_fn(pid=0):
    """the reference drivers' seeding (utilspy.py:22-25, called as worker_init_fn(0) at the top of every driver): the
    FastDVDnet finetune draws its noise from the global NumPy generator seeded here"""
    np.random.seed(42 + pid)
    torch.manual_seed(42 + pid)


# ------------------------------------------------------------------------------------------------ CLI
def _load_model(denoiser, weights):
    if denoiser == 'ffdnet_color':
        from .nets import FFDNet
        net = FFDNet()
    else:
        from .fastdvd import FastDVDnet
        net = torch.nn.DataParallel(FastDVDnet())
    if weights:
        sd = ({k: torch.from_numpy(v) for k, v in np.load(weights).items()} if weights.endswith('.npz')
              else torch.load(weights, map_location='cpu'))
        net.load_state_dict(sd.get('state_dict', sd) if isinstance(sd, dict) else sd, strict=True)
    return net


def main(argv=None):
    ap = argparse.ArgumentParser(description='ADMM-TV warm start + two-stage adaptive PnP-ADMM on one scene file')
    ap.add_argument('scene', help='.mat (v7.3 / HDF5 or v7) or .npz with meas_bayer, mask_bayer, orig_bayer')
    ap.add_argument('--denoiser', default='ffdnet_color', choices=sorted(_SHORT))
    ap.add_argument('--weights', help='denoiser checkpoint (.pth state dict or .npz)')
    ap.add_argument('--ddnet-weights', help='DDnet checkpoint -> deep demosaicking instead of Malvar')
    ap.add_argument('--results', default='./results')
    ap.add_argument('--no-update', action='store_true', help='disable the online finetune')
    ap.add_argument('--no-reuse-model', action='store_true')
    ap.add_argument('--batch-tv', action='store_true', help='ADMM-TV warm start of all measurements as one unit batch (same results, '
                                                           '~3x the throughput on small scenes)')
    args = ap.parse_args(argv)
    worker_init_fn(0)
    _lib.cap_host_threads()
    scene = load_scene(args.scene)
    os.makedirs(args.results, exist_ok=True)
    with open(os.path.join(args.results, 'log.txt'), 'a') as f:
        f.write('cacti midscale bayer: \n')
        wpath = warm_start_path(args.results, scene)
        if os.path.exists(wpath):
            warm = load_warm_start(wpath)
        else:
            warm = run_tv_warm_start(scene, f, save_to=wpath, batch=args.batch_tv)['v']
        dd = None
        if args.ddnet_weights:
            from .ddnet import DDnet
            dd = torch.nn.DataParallel(DDnet())
            dd.load_state_dict(torch.load(args.ddnet_weights, map_location='cpu')['state_dict'], strict=True)
        net = _load_model(args.denoiser, args.weights)
        try:
            sch = schedule_for(args.denoiser, scene.name, dd is not None)
        except KeyError:
            sch = schedule_for(args.denoiser, 'Beauty_bayer', dd is not None)
        out = run_two_stage(scene, warm, args.denoiser, net, dd, sch, update=not args.no_update,
                            reuse_model=not args.no_reuse_model, logf=f, save_dir=args.results)
    if scene.orig_bayer is not None:
        print(round(float(out['psnr'].mean()), 2), round(float(out['ssim'].mean()), 4), sep=', ')
    return 0


if __name__ == '__main__':
    raise SystemExit(main())

"""Import shim that makes the READ-ONLY reference at /root/reference importable on a
GPU-less CPU box (python 3.10 / torch CPU).  Used ONLY by tools/make_golden.py in the
build container to generate the golden vectors under tests/golden/.  Nothing in tests/,
bench.py or the package imports this file, and /root/reference never travels to the GPU box.

What it does (no reference code is copied, only third-party modules are stubbed):
  1. provides `skimage.restoration.denoise_tv_chambolle` and `structural_similarity` by forwarding
     to the REAL scikit-image 0.18.3 (pinned dependency of the reference is 0.18.1, readme.md:15)
     of the py3.9 conda env of this image through a worker process (tools/skimage_tv_server.py),
     and `peak_signal_noise_ratio` with skimage's formula;
  2. registers empty stub modules for imports the hot path never calls (cv2, h5py, imageio,
     torchvision, tensorboardX, colour);
  3. turns `.cuda()` into the identity so the reference's hard-coded CUDA placement runs on CPU.
"""
import os
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

REF = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class SkimageServer:
    """The REAL scikit-image 0.18.3 (+ numpy 1.26, the reference's NumPy generation) lives only in
    the py3.9 conda env, which has no torch; keep one worker process of it and ship arrays over /tmp."""

    def __init__(self):
        here = os.path.dirname(os.path.abspath(__file__))
        self.proc = subprocess.Popen(['/opt/conda/bin/python3.9', os.path.join(here, 'skimage_tv_server.py')],
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd='/tmp')
        self.dir = tempfile.mkdtemp(prefix='skimg_')

    def tv(self, image, weight, n_iter_max, multichannel):
        a, b = os.path.join(self.dir, 'in.npy'), os.path.join(self.dir, 'out.npy')
        np.save(a, image)
        self.proc.stdin.write(f'{a} {b} {weight!r} {n_iter_max} {int(bool(multichannel))}\n')
        self.proc.stdin.flush()
        assert self.proc.stdout.readline().strip() == 'ok'
        return np.load(b)

    def ssim(self, x, y):
        a, b = os.path.join(self.dir, 'a.npy'), os.path.join(self.dir, 'b.npy')
        np.save(a, x)
        np.save(b, y)
        self.proc.stdin.write(f'ssim {a} {b}\n')
        self.proc.stdin.flush()
        return float(self.proc.stdout.readline().strip())


SERVER = None


def install():
    global SERVER
    if SERVER is None:
        SERVER = SkimageServer()

    def denoise_tv_chambolle(image, weight=0.1, eps=2e-4, n_iter_max=200, multichannel=False):
        assert eps == 2e-4
        return SERVER.tv(image, weight, n_iter_max, multichannel)

    def psnr(a, b, data_range=None):
        ft = np.result_type(a.dtype, b.dtype, np.float32)
        err = np.mean((a.astype(ft) - b.astype(ft)) ** 2, dtype=np.float64)
        return 10 * np.log10(data_range ** 2 / err)

    def ssim_stub(a, b, data_range=None, **k):
        assert data_range == 1.
        return SERVER.ssim(a, b)

    sk = _mod('skimage', __version__='0.18.3')
    sk.restoration = _mod('skimage.restoration', denoise_tv_chambolle=denoise_tv_chambolle)
    _mod('skimage.metrics', peak_signal_noise_ratio=psnr, structural_similarity=ssim_stub)
    _mod('skimage.metrics.simple_metrics', peak_signal_noise_ratio=psnr)
    _mod('skimage.metrics._structural_similarity', structural_similarity=ssim_stub)
    for n in ('imageio', 'cv2', 'h5py'):
        _mod(n)
    tv = _mod('torchvision')
    tv.__path__ = []
    _mod('torchvision.utils', make_grid=None)
    _mod('tensorboardX', SummaryWriter=object)
    cu = _mod('colour.utilities',
              as_float_array=lambda a, dtype=None: np.asarray(a, np.float64),
              tstack=lambda a: np.concatenate([np.asarray(x)[..., None] for x in a], -1),
              tsplit=lambda a: np.array([np.asarray(a)[..., i] for i in range(np.asarray(a).shape[-1])]),
              ANCILLARY_COLOUR_SCIENCE_PACKAGES={})
    _mod('colour', utilities=cu)
    try:
        import matplotlib  # noqa: F401
    except Exception:
        mp = _mod('matplotlib')
        mp.__path__ = []
        _mod('matplotlib.pyplot')
    try:
        import six  # noqa: F401
    except Exception:
        _mod('six', u=lambda s: s)
    try:
        import tqdm  # noqa: F401
    except Exception:
        _mod('tqdm', tqdm=lambda it, *a, **k: it)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)
    return denoise_tv_chambolle, psnr


def import_solver():
    install()
    import dvp_linear_inv_2_stage_ADMM_tensor_online as R  # noqa: E402
    # the reference enables anomaly mode globally at import (test_ffdnet_ipol.py:26); results are
    # unaffected, it only makes backward ~10x slower -> switch it off for golden generation.
    torch.autograd.set_detect_anomaly(False)
    return R

"""Runs under /opt/conda/bin/python3.9 (the only interpreter in the build image that has the REAL
scikit-image 0.18.3 + numpy 1.26).  tools/ref_shim.py talks to it over stdin/stdout so that the
imported reference calls the genuine third-party `denoise_tv_chambolle` when golden vectors are
generated.  Protocol: one request per line  `<in.npy> <out.npy> <weight> <n_iter_max> <multichannel:0|1>`,
answered by `ok`.  Also serves `ssim <a.npy> <b.npy>` -> the float printed on one line.
"""
import sys
import numpy as np
from skimage.restoration import denoise_tv_chambolle
from skimage.metrics import structural_similarity

for line in sys.stdin:
    f = line.split()
    if not f:
        continue
    if f[0] == 'ssim':
        a, b = np.load(f[1]), np.load(f[2])
        print(repr(float(structural_similarity(a, b, data_range=1.))), flush=True)
        continue
    x = np.load(f[0])
    out = denoise_tv_chambolle(x, float(f[2]), n_iter_max=int(f[3]), multichannel=bool(int(f[4])))
    np.save(f[1], out)
    print('ok', flush=True)

#!/opt/conda/bin/python3.9
"""Write tests/golden/scene_v73*.mat: scene files in MATLAB's v7.3 container (HDF5 behind a 512-byte MAT-file header) with the
reference's variable names (two_stage_ADMM_Online_FFD_Warm.py:164-197), produced with the genuine HDF5 library (h5py of the
py3.9 environment of the build container; the main interpreter has no h5py) the way MATLAB's `save -v7.3` lays them out:
superblock version 0 behind the user block, old-style groups, double arrays with the axes reversed (MATLAB is column-major),
chunked + deflate for the larger ones, MATLAB_class attributes.  Run:  /opt/conda/bin/python3.9 tools/make_v73_fixture.py"""
import os
import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
rng = np.random.default_rng(73)
H, W, nmask, nmea = 12, 16, 4, 2
orig = np.round(rng.uniform(0, 255, (H, W, nmask * nmea)))                # [0,255] units like the reference's data
mask = (rng.uniform(size=(H, W, nmask)) > 0.5).astype(np.float64)
meas = np.stack([(orig[:, :, i * nmask:(i + 1) * nmask] * mask).sum(2) for i in range(nmea)], 2)
real = rng.uniform(0, 255, (H, W, 3, nmask * nmea))
hdr = (b'MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: Sat Oct  3 2026 HDF5 schema 1.00 .').ljust(116) + \
      b'\x00' * 8 + b'\x00\x02' + b'IM'
hdr = hdr.ljust(512, b'\x00')


def write(path, **opts):
    with h5py.File(path, 'w', userblock_size=512, libver='earliest') as f:
        for name, arr in (('meas_bayer', meas), ('mask_bayer', mask), ('orig_bayer', orig), ('orig', real)):
            a = np.ascontiguousarray(arr.transpose(tuple(reversed(range(arr.ndim)))))        # MATLAB stores column-major
            kw = dict(opts) if a.size > 256 else {}
            d = f.create_dataset(name, data=a, **kw)
            d.attrs['MATLAB_class'] = np.bytes_('double')
        s = f.create_dataset('single_var', data=np.float32(rng.uniform(size=(3, 5))).T.copy())
        s.attrs['MATLAB_class'] = np.bytes_('single')
        u = f.create_dataset('u8_var', data=np.arange(20, dtype=np.uint8).reshape(4, 5).T.copy(), chunks=(5, 2), compression='gzip', shuffle=True)
        u.attrs['MATLAB_class'] = np.bytes_('uint8')
    with open(path, 'r+b') as f:
        f.write(hdr)


write(os.path.join(GOLD, 'scene_v73_chunked.mat'), chunks=True, compression='gzip', compression_opts=3)
write(os.path.join(GOLD, 'scene_v73_plain.mat'))
np.savez(os.path.join(GOLD, 'scene_v73_expected.npz'), meas_bayer=meas, mask_bayer=mask, orig_bayer=orig, orig=real)
for n in ('scene_v73_chunked.mat', 'scene_v73_plain.mat', 'scene_v73_expected.npz'):
    print(n, os.path.getsize(os.path.join(GOLD, n)), 'bytes')

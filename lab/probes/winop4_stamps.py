#!/usr/bin/env python3
"""GPU box: s_memtime stamps of the producer / consumer F(4x4,3x3) kernel (scipnp_conv3x3_c8wp_stamped) on a FastDVDnet layer shape
(WP_SHAPE = "n,c,h,w", default 8,64,256,256): where a workgroup's life goes, from a consumer's (wave 0) and a producer's (wave 8)
point of view.  Units of 100 shader cycles."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adaptivepnp_sci_amd import _lib, ops
sys.path.insert(0, os.path.join(ROOT, 'lab'))
import lablib as diaglib  # noqa: E402  (lab/libscipnp_lab.so)
lib = diaglib.load()
n, c, h, w = (int(v) for v in os.environ.get('WP_SHAPE', '8,64,256,256').split(','))
g = torch.Generator().manual_seed(0)
x8 = ops.to_c8(torch.randn(n, c, h, w, generator=g).cuda())
pk = ops.pack_conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.randn(c, generator=g), Cin=c, Cout=c, device='cuda')
p4 = ops.pack_conv3x3_wino4(pk, c, c)
out, ref = torch.empty_like(x8), torch.empty_like(x8)
nwg = ((w + 63) // 64) * ((h + 7) // 8) * n * (c // 64)
st = torch.zeros(nwg * 128, dtype=torch.int64, device='cuda')
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
OFF = int(os.environ.get('WP_OFF', '0'))   # timing experiments (wrong results): 1 no V stores, 2 no raw staging, 4 no transform
FL = 1 | (OFF << 12)
print('parts switched off:', OFF)
for _ in range(3):
    _lib.check(lib.scipnp_conv3x3_c8wp_stamped(P(x8), P(p4), P(out), n, c, c, h, w, FL, P(st), _lib.stream_ptr()), 'stamped')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.check(lib.scipnp_conv3x3_c8wp_stamped(P(x8), P(p4), P(out), n, c, c, h, w, FL, P(st), _lib.stream_ptr()), 'stamped')
e1.record()
torch.cuda.synchronize()
ops.conv3x3_c8w4(x8, p4, c, relu=True, out=ref)
print(f'{c} -> {c} on {n} x {h} x {w}: stamped launch {e0.elapsed_time(e1) * 1e3:.1f} us; equals scipnp_conv3x3_c8w4: {bool(torch.equal(out, ref))}')
s = st.cpu().numpy().reshape(nwg, 128).astype(np.float64) / 100.0
CG, K = c // 8, 2 * (c // 8)
life = s[:, 6] - s[:, 0]
print(f'workgroups {nwg}; life mean {life.mean():.1f} (min {life.min():.1f}, max {life.max():.1f}) [x 100 cycles]; matrix-pipe time of the loop per SIMD: {K * 2 * 36 * 32 / 100:.1f}')
print(f'  consumer: entry -> loop start {np.mean(s[:, 1] - s[:, 0]):.1f} | loop {np.mean(s[:, 3] - s[:, 1]):.1f} | output transform + image {np.mean(s[:, 4] - s[:, 3]):.1f} | '
      f'second round + stores issued {np.mean(s[:, 5] - s[:, 4]):.1f} | acknowledged {np.mean(s[:, 6] - s[:, 5]):.1f}')
print(f'  producer: opening tiles landed {np.mean(s[:, 56] - s[:, 0]):.1f} after entry | first transform {np.mean(s[:, 57] - s[:, 56]):.1f}')
ci = np.array([s[:, 8 + 2 * k] - (s[:, 1] if k == 0 else s[:, 9 + 2 * (k - 1)]) for k in range(K)])       # MFMAs issued
cw = np.array([s[:, 9 + 2 * k] - s[:, 8 + 2 * k] for k in range(K)])                                      # wait + barrier
pi = np.array([s[:, 64 + 2 * k] - (s[:, 57] if k == 0 else s[:, 65 + 2 * (k - 1)]) for k in range(K)])
pwt = np.array([s[:, 65 + 2 * k] - s[:, 64 + 2 * k] for k in range(K)])
print(f'per k-step (mean over workgroups; 36 MFMAs of one wave = 11.52, two consumers on a SIMD = 23.04):')
print('  consumer issue   : ' + ' '.join(f'{v:5.1f}' for v in ci.mean(axis=1)))
print('  consumer barrier : ' + ' '.join(f'{v:5.1f}' for v in cw.mean(axis=1)))
print('  producer work    : ' + ' '.join(f'{v:5.1f}' for v in pi.mean(axis=1)))
print('  producer barrier : ' + ' '.join(f'{v:5.1f}' for v in pwt.mean(axis=1)))
print(f'  k-step total (consumer) mean {np.mean(ci + cw):.2f}; even k-steps {np.mean((ci + cw)[0::2]):.2f}, odd {np.mean((ci + cw)[1::2]):.2f}')

#!/usr/bin/env python3
"""GPU box: where the persistent Winograd kernel's time goes -- the body layer (96 -> 96, 8 frames of 256 x 256) with parts
of the kernel switched off (scipnp_conv3x3_c8p_diag; timing only, results are wrong by construction)."""
import ctypes as C, os, sys
import torch
os.environ['SCIPNP_WINO_F4'] = '0'                              # classic = the F(2x2) kernel here
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import _lib, ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'lab'))
import lablib as diaglib  # noqa: E402  (lab/libscipnp_lab.so)
lib = diaglib.load()
n, c, h, w = 8, 96, 256, 256
g = torch.Generator().manual_seed(0)
x8 = ops.to_c8(torch.randn(n, c, h, w, generator=g).cuda())
pk = ops.pack_conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.randn(c, generator=g), Cin=c, Cout=c, device='cuda')
pb = ops.pack_conv3x3_wino_both(pk, c, c)
pwinop = diaglib.pack_winop(pk, c, c)
out = torch.empty_like(x8)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
NAMES = {1: 'no U LDS-DMA', 2: 'no transform jobs', 4: 'no raw staging', 8: 'no epilogue', 16: 'no V/U fragment reads', 32: 'no barrier'}


def run(diag):
    _lib.check(lib.scipnp_conv3x3_c8p_diag(P(x8), P(pwinop), P(out), n, c, c, h, w, 1, diag, _lib.stream_ptr()), 'diag')


def timed(fn, reps=5, inner=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    return sorted(ts)[len(ts) // 2]


print(f'classic kernel              {timed(lambda: ops.conv3x3_c8w(x8, pb.w, c, relu=True, out=out)):7.1f} us')
print(f'persistent (product)        {timed(lambda: diaglib.conv3x3_c8p(x8, pwinop, c, relu=True, out=out)):7.1f} us')
for d in (0, 1, 2, 4, 8, 16, 32, 1 | 4, 1 | 2 | 4, 1 | 2 | 4 | 8, 1 | 2 | 4 | 8 | 16, 63, 32 | 16):
    name = ' + '.join(NAMES[b] for b in NAMES if d & b) or 'diag build, nothing off'
    print(f'diag {d:2d}: {timed(lambda: run(d)):7.1f} us   {name}')

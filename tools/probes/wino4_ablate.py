#!/usr/bin/env python3
"""GPU box: where the F(4x4,3x3) Winograd kernel's time goes -- the body layer (96 -> 96, 8 frames of 256 x 256) with parts
of the kernel switched off (scipnp_conv3x3_c8w4_diag; timing only, results are wrong by construction)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import _lib, ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
import diaglib  # noqa: E402  (libscipnp_diag.so: the laboratory entries)
lib = diaglib.load()
n, c, h, w = 8, 96, 256, 256
g = torch.Generator().manual_seed(0)
x8 = ops.to_c8(torch.randn(n, c, h, w, generator=g).cuda())
pk = ops.pack_conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.randn(c, generator=g), Cin=c, Cout=c, device='cuda')
p4, p2 = ops.pack_conv3x3_wino4(pk, c, c), ops.pack_conv3x3_wino(pk, c, c)
out = torch.empty_like(x8)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
NAMES = {1: 'no transform', 2: 'no raw staging', 4: 'no U LDS-DMA', 8: 'no barriers', 16: 'no MFMAs', 32: 'no epilogue',
         128: 'matrix work as v_mfma_f32_32x32x2_f32 on the same registers', 256: 'raw requests of 1 KB contiguous memory', 512: 'output stores of 1 KB contiguous memory', 1024: 'stores with the nt hint (isolated layer only: see conv_wino4.hip)', 2048: 'stores with the sc1 hint', 4096: 'classic per-lane store epilogue (correct results)'}


KER = os.environ.get('W4_KERNEL', '4')     # '6' / 'n': the not-adopted kernels of lab/ (needs `make -C lab`; masks 1, 2, 4, 8, 16, 6, 7, 15, 48, 49)
lablib = None
if KER in ('6', 'n') or os.environ.get('W4_WITH_LAB') == '1':
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'lab'))
    import lablib  # noqa: E402
diag_fn = lablib.load().scipnp_conv3x3_c8w6_diag if KER == '6' else lablib.load().scipnp_conv3x3_c8wn_diag if KER == 'n' else lib.scipnp_conv3x3_c8w4_diag
if KER == 'n':
    p4 = lablib.repack_wino4n(p4, c, c)


def run(diag):
    _lib.check(diag_fn(P(x8), P(p4), P(out), n, c, c, h, w, 1, diag, _lib.stream_ptr()), 'diag')


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


print(f'F(2x2) kernel               {timed(lambda: ops.conv3x3_c8w(x8, p2, c, relu=True, out=out)):7.1f} us')
p4_32 = ops.pack_conv3x3_wino4(pk, c, c)
print(f'F(4x4) kernel (product)     {timed(lambda: ops.conv3x3_c8w4(x8, p4_32, c, relu=True, out=out)):7.1f} us')
if lablib is not None:
    print(f'F(4x4) 12-wave workgroups   {timed(lambda: lablib.conv3x3_c8w6(x8, p4_32, c, relu=True, out=out)):7.1f} us')
    pn_ = lablib.repack_wino4n(p4_32, c, c)
    print(f'F(4x4) 16-co, 3 WGs per CU  {timed(lambda: lablib.conv3x3_c8wn(x8, pn_, c, relu=True, out=out)):7.1f} us')
print('ablations of scipnp_conv3x3_c8w' + KER)
MASKS = [int(v) for v in os.environ['W4_MASKS'].split(',')] if os.environ.get('W4_MASKS') else \
    (1, 2, 4, 8, 16, 32, 3, 5, 9, 10, 12, 14, 6, 7, 15, 39, 47, 48, 49, 55, 63, 128)
for d in MASKS:
    name = ' + '.join(NAMES[b] for b in NAMES if d & b)
    print(f'diag {d:2d}: {timed(lambda: run(d)):7.1f} us   {name}', flush=True)

#!/usr/bin/env python3
"""GPU box: the F(4x4,3x3) kernel's PERSISTENT form (csrc/conv_wino4.hip PERSIST = true: workgroups walk the units, the next unit's
opening requests are issued from inside the current unit's epilogue; classic per-lane stores; 95 spilled registers at two waves per
SIMD) against the product form (one unit per workgroup, whole-line stores) on the layers whose workgroups spend most of their life at
the unit boundary -- the narrow-input layers -- and on the body layer.  scipnp_conv3x3_c8w4_diag, diag 8192 (2 workgroups per CU) /
8193 (1 per CU) / 4096 (classic stores, one unit per workgroup).  Results are bit-identical in every form."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from adaptivepnp_sci_amd import _lib, ops  # noqa: E402
import diaglib  # noqa: E402

lib = diaglib.load()
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


def timed(fn, inner=20, reps=5):
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


g = torch.Generator().manual_seed(0)
for name, n, cin, cout, h, w in (('FFDNet head 16 -> 96 @256^2 x 8', 8, 16, 96, 256, 256), ('FastDVDnet 16 -> 96 @512^2 x 8', 8, 16, 96, 512, 512),
                                 ('DDnet 8 -> 96 @512^2 x 24', 24, 8, 96, 512, 512), ('DDnet 16 -> 96 @512^2 x 16', 16, 16, 96, 512, 512),
                                 ('FastDVDnet 32 -> 32 @512^2 x 8', 8, 32, 32, 512, 512), ('FFDNet body 96 -> 96 @256^2 x 8', 8, 96, 96, 256, 256)):
    x8 = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    pk = ops.pack_conv3x3(wt, torch.randn(cout, generator=g), Cin=cin, Cout=cout, device='cuda')
    p4 = ops.pack_conv3x3_wino4(pk, cin, cout)
    out = torch.empty(n, (cout + 7) // 8, h, w, 8, device='cuda')
    ref = ops.conv3x3_c8w4(x8, p4, cout, relu=True).clone()

    def diag(d):
        _lib.check(lib.scipnp_conv3x3_c8w4_diag(P(x8), P(p4), P(out), n, cin, cout, h, w, 1, d, _lib.stream_ptr()), 'diag')
    same = []
    for d in (8192, 8193, 4096):
        out.fill_(-7.0)
        diag(d)
        same.append(bool(torch.equal(out, ref)))
    t0 = timed(lambda: ops.conv3x3_c8w4(x8, p4, cout, relu=True, out=out))
    tc = timed(lambda: diag(4096))
    t2 = timed(lambda: diag(8192))
    t1 = timed(lambda: diag(8193))
    print(f'{name:34s} product {t0:7.1f} us | classic stores {tc:7.1f} | persistent, 2 per CU {t2:7.1f} | persistent, 1 per CU {t1:7.1f} | bit-identical {same}')

#!/usr/bin/env python3
"""GPU box: the F(4x4) kernel's stores with the nt hint (laboratory instantiation, diag 1024) against plain stores on layers whose
OUTPUT is far larger than the 256 MB Infinity Cache (DDnet's 8 -> 96 / 16 -> 96 on 24 evaluations of 512 x 512: 2.4 GB) -- isolated
and followed by a consumer layer that reads that output"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adaptivepnp_sci_amd import _lib, ops
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import diaglib  # noqa: E402
lib = diaglib.load()
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
g = torch.Generator().manual_seed(0)


def timed(fn, inner=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner * 1e3


for (n, cin, cmid, cout, h, w) in ((24, 8, 96, 24, 512, 512), (16, 16, 96, 24, 512, 512), (8, 16, 96, 32, 512, 512), (24, 40, 40, 40, 256, 256),
                                   (8, 96, 96, 96, 256, 256)):
    x = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
    pk1 = ops.pack_conv3x3(torch.randn(cmid, cin, 3, 3, generator=g) * 0.05, torch.randn(cmid, generator=g), Cin=cin, Cout=cmid, device='cuda')
    pk2 = ops.pack_conv3x3(torch.randn(cout, cmid, 3, 3, generator=g) * 0.05, torch.randn(cout, generator=g), Cin=cmid, Cout=cout, device='cuda')
    p1, p2 = ops.pack_conv3x3_wino4(pk1, cin, cmid), ops.pack_conv3x3_wino4(pk2, cmid, cout)
    mid = torch.empty(n, cmid // 8, h, w, 8, device='cuda')
    out = torch.empty(n, (cout + 7) // 8, h, w, 8, device='cuda')
    res = {}
    for name, d in (('plain', 0), ('nt', 1024)):
        if d:
            l1 = lambda: _lib.check(lib.scipnp_conv3x3_c8w4_diag(P(x), P(p1), P(mid), n, cin, cmid, h, w, 1, d, _lib.stream_ptr()), 'diag')  # noqa: E731
        else:
            l1 = lambda: ops.conv3x3_c8w4(x, p1, cmid, relu=True, out=mid)  # noqa: E731
        def pair():
            l1()
            ops.conv3x3_c8w4(mid, p2, cout, relu=True, out=out)
        res[name] = (timed(l1), timed(pair))
    mb = mid.numel() * 4 / 1e6
    print(f'{n:2d} x {cin:3d} -> {cmid:3d} (-> {cout:3d}) @ {h}x{w}, output {mb:6.0f} MB: ' +
          '  '.join(f'{k}: alone {v[0]:7.1f} us, with its consumer {v[1]:7.1f} us' for k, v in res.items()), flush=True)
print('done')

#!/usr/bin/env python3
"""GPU box: the F(4x4,3x3) Winograd kernel (csrc/conv_wino4.hip) against float64 on a few shapes, then timed against the
F(2x2,3x3) kernel on the layer shapes of the denoisers."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import ops

g = torch.Generator().manual_seed(4)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for n, cin, cout, h, w in ((1, 8, 32, 8, 64), (1, 32, 32, 8, 64), (2, 96, 96, 20, 70), (1, 16, 96, 37, 129), (3, 64, 24, 5, 7),
                           (1, 40, 72, 64, 64)):
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), padding=1)
    pk = ops.pack_conv3x3(wt, b, Cin=cin, Cout=cout, device='cuda')
    p4, p2 = ops.pack_conv3x3_wino4(pk, cin, cout), ops.pack_conv3x3_wino(pk, cin, cout)
    xc = ops.to_c8(x.cuda())
    got = ops.from_c8(ops.conv3x3_c8w4(xc, p4, cout)).cpu()
    g2 = ops.from_c8(ops.conv3x3_c8w(xc, p2, cout)).cpu()
    got_r = ops.from_c8(ops.conv3x3_c8w4(xc, p4, cout, relu=True, residual=ops.to_c8(res.cuda()))).cpu()
    print(f'{(n, cin, cout, h, w)}: F(4x4) vs float64 {rel(got, ref):.3e}   F(2x2) {rel(g2, ref):.3e}   relu+residual '
          f'{rel(got_r, torch.relu(ref + res.double())):.3e}   max abs err {float((got.double() - ref).abs().max()):.2e}', flush=True)

for name, n, cin, cout, h, w in (('FFDNet body', 8, 96, 96, 256, 256), ('FastDVDnet 64->64', 8, 64, 64, 256, 256),
                                 ('FastDVDnet 128->128', 8, 128, 128, 128, 128), ('FastDVDnet 96->32', 8, 96, 32, 512, 512),
                                 ('FastDVDnet 32->32', 8, 32, 32, 512, 512), ('tile body', 16, 96, 96, 128, 128),
                                 ('FFDNet head 16->96', 8, 16, 96, 256, 256), ('FFDNet tail 96->16', 8, 96, 16, 256, 256),
                                 ('FastDVDnet inc 16->96', 8, 16, 96, 512, 512), ('FastDVDnet out 32->8', 8, 32, 8, 512, 512)):
    x8 = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    pk = ops.pack_conv3x3(wt, torch.randn(cout, generator=g), Cin=cin, Cout=cout, device='cuda')
    p4, p2 = ops.pack_conv3x3_wino4(pk, cin, cout), ops.pack_conv3x3_wino(pk, cin, cout)
    o4 = torch.empty(n, cout // 8, h, w, 8, device='cuda')
    o2 = torch.empty_like(o4)
    od = torch.empty_like(o4)
    variants = {'F(2x2)': lambda: ops.conv3x3_c8w(x8, p2, cout, relu=True, out=o2),
                'F(4x4)': lambda: ops.conv3x3_c8w4(x8, p4, cout, relu=True, out=o4),
                'direct': lambda: ops.conv3x3_c8(x8, pk, cout, relu=True, out=od)}
    for f in variants.values():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    out = [f'{name:22s} F(4x4) vs F(2x2) rel-L2 {rel(o4, o2):.2e}']
    for k, f in variants.items():
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        us = sorted(ts)[2]
        mf = (36 if k == 'F(4x4)' else 64 if k == 'F(2x2)' else 576) / 16.0 * 2 * cin * ((cout + 31) // 32 * 32) * h * w * n / 4
        out.append(f'{k} {us:7.1f} us (matrix-pipe duty {mf / us / 1e6 / 157.3:4.2f})')
    out.append(f'tensors {4e-6 * n * h * w * (cin + cout):.0f} MB = {4e-6 * n * h * w * (cin + cout) / us:.2f} TB/s at the fastest' if False else
               f'in + out {4e-6 * n * h * w * (cin + cout):.0f} MB')
    print('   '.join(out), flush=True)

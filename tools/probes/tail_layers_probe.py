#!/usr/bin/env python3
"""GPU box: the layers with FEW OUTPUT channels (read-dominated: FFDNet tail 96 -> 12, FastDVDnet outc 32 -> 3 (8), DDnet 24 -> 8 and
8 -> 8) on the kernels that can run them -- F(2x2,3x3) (csrc/conv_wino.hip: what the library dispatches for fewer than 24 outputs),
F(4x4,3x3) with 32-channel workgroups (csrc/conv_wino4.hip, half or three quarters of its output block padding) and F(4x4,3x3) with
16-channel workgroups, three per CU (lab/csrc/conv_wino4n.hip; needs `make -C lab`) -- microseconds per launch and HBM bytes moved."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'lab'))
from adaptivepnp_sci_amd import ops  # noqa: E402
import lablib  # noqa: E402


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
for name, n, cin, cout, h, w in (('FFDNet tail 96 -> 12 @256^2 x 8', 8, 96, 16, 256, 256), ('FastDVDnet outc 32 -> 8 @512^2 x 8', 8, 32, 8, 512, 512),
                                 ('DDnet 24 -> 8 @512^2 x 24', 24, 24, 8, 512, 512), ('DDnet 8 -> 8 @512^2 x 24', 24, 8, 8, 512, 512),
                                 ('FFDNet head 16 -> 96 @256^2 x 8', 8, 16, 96, 256, 256)):
    x8 = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    pk = ops.pack_conv3x3(wt, torch.randn(cout, generator=g), Cin=cin, Cout=cout, device='cuda')
    p2, p4 = ops.pack_conv3x3_wino(pk, cin, cout), ops.pack_conv3x3_wino4(pk, cin, cout)
    pn = lablib.repack_wino4n(p4, cin, cout)
    out = torch.empty(n, (cout + 7) // 8, h, w, 8, device='cuda')
    ref = ops.conv3x3_c8w4(x8, p4, cout, relu=True).clone()
    got = lablib.conv3x3_c8wn(x8, pn, cout, relu=True)
    mb = (x8.numel() + out.numel()) * 4 / 1e6
    t2 = timed(lambda: ops.conv3x3_c8w(x8, p2, cout, relu=True, out=out))
    t4 = timed(lambda: ops.conv3x3_c8w4(x8, p4, cout, relu=True, out=out))
    tn = timed(lambda: lablib.conv3x3_c8wn(x8, pn, cout, relu=True, out=out))
    print(f'{name:36s} {mb:7.1f} MB | F(2x2) {t2:7.1f} us ({mb / t2 / 1e3:5.2f} TB/s) | F(4x4) 32-co workgroups {t4:7.1f} us ({mb / t4 / 1e3:5.2f} TB/s) | '
          f'F(4x4) 16-co workgroups x3 per CU {tn:7.1f} us ({mb / tn / 1e3:5.2f} TB/s) | 16-co == 32-co result: {bool(torch.equal(got, ref))}')

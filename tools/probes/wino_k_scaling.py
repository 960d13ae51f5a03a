#!/usr/bin/env python3
"""GPU box: fp32 Winograd conv time against the number of input-channel groups at fixed output shape (96 output channels,
8 frames of 256 x 256): slope = cost of one channel group per unit, intercept = per-unit fixed cost (prologue, output
transform, stores)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import ops

n, h, w, cout = 8, 256, 256, 96
g = torch.Generator().manual_seed(0)
for per_unit in (True,):
    pts = []
    for cin in (8, 24, 48, 96, 144, 192):
        x8 = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
        wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        pw = ops.pack_conv3x3_wino(ops.pack_conv3x3(wt, None, Cin=cin, Cout=cout, device='cuda'), cin, cout)
        out = torch.empty(n, cout // 8, h, w, 8, device='cuda')
        f = lambda: ops.conv3x3_c8w(x8, pw, cout, relu=True, out=out)
        for _ in range(3):
            f()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        pts.append((cin // 8, sorted(ts)[2]))
    (g0, t0), (g1, t1) = pts[2], pts[-1]
    slope = (t1 - t0) / (g1 - g0)
    print(' '.join(f'CG={c}: {t:.1f}us' for c, t in pts),
          f'| slope {slope:.2f} us/group, intercept {t1 - slope * g1:.1f} us; MFMA-only slope would be {246.0 / 12:.2f}')

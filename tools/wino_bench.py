#!/usr/bin/env python3
"""GPU box: FFDNet body layer (96 -> 96, 8 frames of 256 x 256): fp32 direct MFMA vs fp32 Winograd F(2x2,3x3) / F(4x4,3x3) MFMA."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

n, c, h, w = 8, 96, 256, 256
flop = 2.0 * 9 * c * c * h * w * n
g = torch.Generator().manual_seed(0)
x = torch.randn(n, c, h, w, generator=g).cuda()
wt = torch.randn(c, c, 3, 3, generator=g) * 0.05
b = torch.randn(c, generator=g)
x8 = ops.to_c8(x)
pk = ops.pack_conv3x3(wt, b, Cin=c, Cout=c, device='cuda')
pw = ops.pack_conv3x3_wino(pk, c, c)
p4 = ops.pack_conv3x3_wino4(pk, c, c)
o84 = torch.empty_like(x8)
o8, o8w, o8v, o8p = torch.empty_like(x8), torch.empty_like(x8), torch.empty_like(x8), torch.empty_like(x8)
variants = {'fp32 direct': lambda: ops.conv3x3_c8(x8, pk, c, relu=True, out=o8),
            'fp32 winograd': lambda: ops.conv3x3_c8w(x8, pw, c, relu=True, out=o8w),
            'fp32 wino 16row': lambda: ops.conv3x3_c8w(x8, pw, c, relu=True, out=o8v, rows16=True),
            'fp32 wino F(4x4)': lambda: ops.conv3x3_c8w4(x8, p4, c, relu=True, out=o84)}
for f in variants.values():
    for _ in range(3):
        f()
torch.cuda.synchronize()
print('winograd vs direct rel-L2:', float((o8w - o8).norm() / o8.norm()), '16-row:', float((o8v - o8).norm() / o8.norm()),
      'bitwise equal variants:', bool(torch.equal(o8w, o8v)), 'persistent == classic:', bool(torch.equal(o8w, o8p)),
      'F(4x4) vs direct rel-L2:', float((o84 - o8).norm() / o8.norm()))
res = {k: [] for k in variants}
for r in range(5):
    for k, f in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
for k, v in res.items():
    us = sorted(v)[len(v) // 2]
    print(f'{k:16s} median {us:8.1f} us  min {min(v):8.1f} us   {flop / us / 1e6:7.1f} TFLOP/s (algorithmic)  {flop / us / 1e6 / 157.3:5.2f} of fp32 MFMA peak')

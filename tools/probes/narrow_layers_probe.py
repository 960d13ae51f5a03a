#!/usr/bin/env python3
"""GPU box: the fp32 layers that ops.wino_f4_shape keeps on the F(2x2,3x3) kernel (fewer than 32 output or 16 input channels) at
the sizes DDnet / FastDVDnet run them: F(2x2) against F(4x4) (whose workgroup computes 32 output channels: a narrower layer pays
for the padding) against the direct kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import ops
g = torch.Generator().manual_seed(0)
shapes = [(24, 96, 24, 512, 512), (16, 96, 24, 512, 512), (24, 24, 24, 512, 512), (24, 96, 24, 256, 256), (24, 24, 24, 256, 256),
          (24, 24, 8, 512, 512), (24, 8, 8, 512, 512), (24, 8, 96, 512, 512), (24, 16, 96, 512, 512), (24, 16, 96, 256, 256),
          (8, 96, 16, 256, 256), (24, 40, 24, 512, 512)]
for (n, cin, cout, h, w) in shapes:
    x = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
    pk = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) * 0.05, torch.randn(cout, generator=g), Cin=cin, Cout=cout, device='cuda')
    pw = ops.pack_conv3x3_wino(pk, cin, cout)
    p4 = ops.pack_conv3x3_wino4(pk, cin, cout)
    outs = [torch.empty(n, (cout + 7) // 8, h, w, 8, device='cuda') for _ in range(3)]
    fns = {'direct': lambda: ops.conv3x3_c8(x, pk, cout, relu=True, out=outs[0]),
           'F(2x2)': lambda: ops.conv3x3_c8w(x, pw, cout, relu=True, out=outs[1]),
           'F(4x4)': lambda: ops.conv3x3_c8w4(x, p4, cout, relu=True, out=outs[2])}
    res = {}
    for k, f in fns.items():
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        res[k] = e0.elapsed_time(e1) / 10 * 1e3
    d24 = float((outs[2] - outs[0]).norm() / outs[0].norm())
    print(f'{n:3d} x {cin:3d} -> {cout:3d} @ {h}x{w}: ' + '  '.join(f'{k} {v:8.1f} us' for k, v in res.items()) + f'   F(4x4) vs direct rel-L2 {d24:.1e}', flush=True)
print('done')

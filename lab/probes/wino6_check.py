#!/usr/bin/env python3
"""GPU box: the three-waves-per-SIMD F(4x4,3x3) kernel (scipnp_conv3x3_c8w6) against the two-wave kernel (scipnp_conv3x3_c8w4):
bit-identity on ragged shapes and every epilogue, then the FFDNet body layer's time (HIP events, 50 launches each, A/B/A/B)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'lab'))
import lablib as diaglib  # noqa: E402  (lab/libscipnp_lab.so)

g = torch.Generator().manual_seed(1)
ok = True
for (n, cin, cout, h, w) in ((1, 16, 32, 8, 64), (2, 24, 40, 13, 70), (1, 96, 96, 37, 131), (3, 32, 16, 20, 64), (1, 8, 96, 9, 9), (2, 64, 64, 64, 64)):
    x = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
    pk = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) * 0.1, torch.randn(cout, generator=g), Cin=cin, Cout=cout, device='cuda')
    p4 = ops.pack_conv3x3_wino4(pk, cin, cout)
    res = ops.to_c8(torch.randn(n, cout, h, w, generator=g).cuda())
    msk = ops.to_c8(torch.randn(n, cout, h, w, generator=g).cuda())
    for kw in ({}, {'relu': True}, {'relu': True, 'residual': res, 'head': True}, {'mask_src': msk, 'residual': res}):
        a = ops.conv3x3_c8w4(x, p4, cout, **kw)
        b = diaglib.conv3x3_c8w6(x, p4, cout, **kw)
        kwn = {k: v for k, v in kw.items() if k != 'head'}
        cn = diaglib.conv3x3_c8wn(x, diaglib.repack_wino4n(p4, cin, cout), cout, **kwn)
        same = bool(torch.equal(a, b)) and bool(torch.equal(a, cn))
        ok &= same
        if not same:
            d = (a - b).abs()
            print('MISMATCH', (n, cin, cout, h, w), sorted(kw), 'max', float(d.max()), 'rel', float((a - b).norm() / a.norm()),
                  'bad rows', sorted(set((d.sum(dim=(0, 1, 3, 4)) > 0).nonzero().flatten().tolist()))[:12])
print('bit-identical to scipnp_conv3x3_c8w4 on every shape and epilogue:', ok)

n, c, h, w = 8, 96, 256, 256
x8 = ops.to_c8(torch.randn(n, c, h, w, generator=g).cuda())
pk = ops.pack_conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.randn(c, generator=g), Cin=c, Cout=c, device='cuda')
p4 = ops.pack_conv3x3_wino4(pk, c, c)
o4, o6, on = torch.empty_like(x8), torch.empty_like(x8), torch.empty_like(x8)
pn = diaglib.repack_wino4n(p4, c, c)
fns = {'c8w4 (product: 2 x 4 waves / CU, 32 co)  ': lambda: ops.conv3x3_c8w4(x8, p4, c, relu=True, out=o4),
       'c8w6 (1 x 12 waves / CU, lockstep, 32 co)': lambda: diaglib.conv3x3_c8w6(x8, p4, c, relu=True, out=o6),
       'c8wn (3 x 4 waves / CU, 16 co)           ': lambda: diaglib.conv3x3_c8wn(x8, pn, c, relu=True, out=on)}
for f in fns.values():
    for _ in range(200):
        f()
torch.cuda.synchronize()
for rnd in range(3):
    for name, f in fns.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f'round {rnd}  {name}: {us:7.1f} us   {21.743 / us * 1e3 / 1e3:6.1f} TFLOP/s executed = {21.743e9 / (us * 1e-6) / 157.3e12:.3f} of the fp32 MFMA peak')
print('body layer equal:', bool(torch.equal(o4, o6)), bool(torch.equal(o4, on)))

#!/usr/bin/env python3
"""Micro-benchmark of the FFDNet body-layer convolution (96->96, 8 frames of 256x256) on the GPU box:
fp32 MFMA (conv.hip) vs split-fp16 MFMA (conv_split.hip).  Interleaved rounds in one process."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import ops

n, c, h, w = 8, 96, 256, 256
flop = 2.0 * 9 * c * c * h * w * n
g = torch.Generator().manual_seed(0)
x = torch.randn(n, c, h, w, generator=g).cuda()
wt = torch.randn(c, c, 3, 3, generator=g) * 0.05
b = torch.randn(c, generator=g)
x8 = ops.to_c8(x)
xs = ops.c8_to_c8s(x8)
pk = ops.pack_conv3x3(wt, b, Cin=c, Cout=c, device='cuda')
pks = ops.pack_conv3x3_split(wt, b, Cin=c, Cout=c, device='cuda')
o8 = torch.empty_like(x8)
os_ = torch.empty_like(xs)
variants = {'fp32': lambda: ops.conv3x3_c8(x8, pk, c, relu=True, out=o8),
            'f16x3': lambda: ops.conv3x3_c8s(xs, pks, c, relu=True, out=os_),
            'f16x3 linear tile order (no XCD remap)': lambda: ops.conv3x3_c8s(xs, pks, c, relu=True, out=os_, variant=0x800),
            'f16x3 W double-buffered, 2 WG/CU': lambda: ops.conv3x3_c8s(xs, pks, c, relu=True, out=os_, variant=0x200),
            'f16x3 16x16x32 MFMA': lambda: ops.conv3x3_c8s(xs, pks, c, relu=True, out=os_, variant=0x400)}
variants['f16x3']()
torch.cuda.synchronize()
ref_out = None
for name, f in variants.items():
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    if name.startswith('f16x3'):
        if ref_out is None:
            ref_out = os_.clone()
        else:
            d = (ops.c8s_to_float(os_) - ops.c8s_to_float(ref_out)).norm() / ops.c8s_to_float(ref_out).norm()
            print(f'{name}: output identical to default: {bool(torch.equal(os_, ref_out))}, rel-L2 {float(d):.2e}')
torch.cuda.synchronize()
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
    print(f'{k:26s} median {us:8.1f} us  min {min(v):8.1f} us   {flop / us / 1e6:7.1f} TFLOP/s (algorithmic)')

if os.environ.get('CONV_BENCH_SHORT'):
    sys.exit(0)

# FastDVDnet DenBlock layer shapes (512x512 frames): 64 ch at 256x256, 128 ch at 128x128, stride-2 and PixelShuffle layers
def layer(cin, cout, hh, ww, **kw):
    xx = ops.c8_to_c8s(ops.to_c8(torch.randn(n, cin, hh, ww, generator=g).cuda()))
    pp = ops.pack_conv3x3_split(torch.randn(cout, cin, 3, 3, generator=g) * 0.05, None, Cin=cin, Cout=cout, device='cuda')
    out = ops.conv3x3_c8s(xx, pp, cout, relu=not kw.get('shuffle'), **kw)
    ho, wo = (hh // 2, ww // 2) if kw.get('stride2') else (hh, ww)
    return (lambda: ops.conv3x3_c8s(xx, pp, cout, relu=not kw.get('shuffle'), out=out, **kw)), 2.0 * 9 * cin * cout * ho * wo * n
for name, args, kw in [('96->32 512^2', (96, 32, 512, 512), {}), ('32->64 s2 512^2', (32, 64, 512, 512), dict(stride2=True)),
                       ('64->64 256^2', (64, 64, 256, 256), {}), ('64->128 s2 256^2', (64, 128, 256, 256), dict(stride2=True)),
                       ('128->128 128^2', (128, 128, 128, 128), {}), ('128->256 shuf 128^2', (128, 256, 128, 128), dict(shuffle=True)),
                       ('64->128 shuf 256^2', (64, 128, 256, 256), dict(shuffle=True)),
                       ('128->256 shuf->c8s', (128, 256, 128, 128), dict(shuffle='c8s')),
                       ('64->128 shuf->c8s', (64, 128, 256, 256), dict(shuffle='c8s')), ('32->32 512^2', (32, 32, 512, 512), {})]:
    f, fl = layer(*args, **kw)
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
    us = sorted(ts)[2]
    print(f'f16x3 {name:22s} median {us:8.1f} us   {fl / us / 1e6:7.1f} TFLOP/s (algorithmic)')
    if not kw:                       # the same layer on the double-buffered two-workgroup form
        f, fl = layer(*args, variant=0x200)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f'      {"(W double-buffered)":22s}        {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us')

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
            'f16x3': lambda: ops.conv3x3_c8s(xs, pks, c, relu=True, out=os_)}
import ctypes as C
from adaptivepnp_sci_amd import _lib
lib = _lib.load()
def raw(flags):
    def f():
        _lib.check(lib.scipnp_conv3x3_c8s(C.c_void_p(xs.data_ptr()), C.c_void_p(pks.data_ptr()), C.c_void_p(os_.data_ptr()),
                                          n, c, c, h, w, 1 | flags, C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'x')
    return f
variants['f16x3 late-fill'] = raw(0x2000)
variants['f16x3 no-stage'] = raw(0x1000)
variants['f16x3 no-stage no-barrier'] = raw(0x1000 | 0x4000)
variants['f16x3 register staging'] = raw(0x20000)
variants['nostage nopkmul'] = raw(0x1000 | 0x8000)
variants['nostage noldsread'] = raw(0x1000 | 0x10000)
variants['nostage nopkmul noldsread'] = raw(0x1000 | 0x18000)
variants['f16x3']()
torch.cuda.synchronize()
ref_out = os_.clone()
for name, f in variants.items():
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    if name.startswith('f16x3') and 'no-' not in name:
        print(f'{name:26s} output identical to default: {bool(torch.equal(os_, ref_out))}')
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

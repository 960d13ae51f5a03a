#!/usr/bin/env python3
"""GPU box: where the time of the F(4x4)-domain weight-gradient kernel goes (csrc/wgrad_wino4.hip, laboratory instantiation in
libscipnp_diag.so): timing-only ablations at the FFDNet body layer, 96 -> 96 on 8 frames of 256 x 256, with the slab
reduction and the finish kernel included in every figure; then the SIMD every wave of a workgroup lands on."""
import collections
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diaglib  # noqa: E402
from adaptivepnp_sci_amd import _lib  # noqa: E402
lib, diag = _lib.load(), diaglib.load()
n, c, h, w = 8, 96, 256, 256
ns = int(os.environ.get('WW4_SLABS', 28))
torch.manual_seed(0)
act = torch.randn(n, c // 8, h, w, 8, device='cuda')
dz = torch.randn(n, c // 8, h, w, 8, device='cuda')
ws = torch.zeros(lib.scipnp_conv3x3_wgrad_wino4_workspace_floats(c, c, ns), device='cuda')
dW = torch.empty(c, c, 3, 3, device='cuda')
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


def run(dbg):
    _lib.check(diag.scipnp_diag_conv3x3_wgrad_wino4(p(act), p(dz), p(dW), p(ws), ns, n, c, c, c, c, h, w, dbg, _lib.stream_ptr()), 'diag')


def timed(dbg):
    for _ in range(3):
        run(dbg)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(dbg)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    return sorted(ts)[2]


chunks = n * (h // 4) * (w // 32) / ns
rows = ((0, 'the kernel'), (2, 'no raw loads / LDS stores'), (32, 'loads, no LDS stores of them'), (64, 'LDS stores, no loads'),
        (6, 'MFMAs only (phase B)'), (3, 'transform only (phase A)'), (1, 'no MFMAs'), (7, 'barriers, slab stores, reduction only'))
base = None
print(f'F(4x4) weight gradient 96 -> 96, {n} x {h} x {w}, {ns} slabs x 9 blocks, {chunks:.1f} chunks per workgroup')
for dbg, what in rows:
    us = timed(dbg)
    print(f'  dbg {dbg:2d}  {what:40s} {us:7.1f} us')
run(8)
torch.cuda.synchronize()
ids = ws[:ns * 9 * 12].view(torch.int32).cpu().numpy().reshape(-1, 12)
pat = collections.Counter(tuple((int(v) >> 4) & 3 for v in row) for row in ids)
print('SIMD of waves 0..11 -> number of workgroups:', dict(pat))

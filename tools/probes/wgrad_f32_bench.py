#!/usr/bin/env python3
"""GPU box: fp32 weight-gradient kernel (csrc/finetune.hip), FFDNet body layer 96 -> 96, 8 frames of 256x256."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import _lib
lib = _lib.load()
n, c, h, w, nslab = 8, 96, 256, 256, 256
torch.manual_seed(0)
act = torch.randn(n, c // 8, h, w, 8, device='cuda')
dz = torch.randn(n, c // 8, h, w, 8, device='cuda')
dW = torch.empty(c, c, 3, 3, device='cuda')
ws = torch.empty(lib.scipnp_conv3x3_wgrad_workspace_floats(c, c, nslab), device='cuda')
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
f = lambda: _lib.check(lib.scipnp_conv3x3_wgrad(p(act), p(dz), p(dW), p(ws), nslab, n, c, c, c, c, h, w, _lib.stream_ptr()), 'wgrad')  # noqa: E731
for _ in range(3):
    f()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
us = sorted(ts)[2]
flop = 2.0 * 9 * c * c * h * w * n
print(f'fp32 wgrad 96->96: {us:.1f} us (incl. the slab reduction)  {flop / us / 1e6:.1f} TFLOP/s = {flop / us / 1e6 / 157.3:.2f} of the fp32 MFMA peak; checksum {float(dW.double().sum()):.6e} {float(dW.double().abs().sum()):.6e}')

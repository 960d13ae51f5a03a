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
nsw = int(os.environ.get('WW_SLABS', 85))
wsw = torch.empty(lib.scipnp_conv3x3_wgrad_wino_workspace_floats(c, c, nsw), device='cuda')
dWw = torch.empty(c, c, 3, 3, device='cuda')
fw = lambda: _lib.check(lib.scipnp_conv3x3_wgrad_wino(C.c_void_p(act.data_ptr()), C.c_void_p(dz.data_ptr()), C.c_void_p(dWw.data_ptr()), C.c_void_p(wsw.data_ptr()), nsw, n, c, c, c, c, h, w, _lib.stream_ptr()), 'wgrad wino')  # noqa: E731
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

for _ in range(3):
    fw()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fw()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
us = sorted(ts)[2]
print(f'fp32 Winograd wgrad 96->96 ({nsw} slabs): {us:.1f} us  {flop / us / 1e6:.1f} TFLOP/s algorithmic = {flop / us / 1e6 / 157.3:.2f} of the fp32 MFMA peak; '
      f'rel-L2 vs direct {float((dWw - dW).norm() / dW.norm()):.2e}')
ns4 = int(os.environ.get('WW4_SLABS', 28))
ws4 = torch.empty(lib.scipnp_conv3x3_wgrad_wino4_workspace_floats(c, c, ns4), device='cuda')
dW4 = torch.empty(c, c, 3, 3, device='cuda')
f4 = lambda: _lib.check(lib.scipnp_conv3x3_wgrad_wino4(p(act), p(dz), p(dW4), p(ws4), ns4, n, c, c, c, c, h, w, _lib.stream_ptr()), 'wgrad wino4')  # noqa: E731
for _ in range(3):
    f4()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f4()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
us = sorted(ts)[2]
print(f'fp32 Winograd F(4x4) wgrad 96->96 ({ns4} slabs): {us:.1f} us  {flop / us / 1e6:.1f} TFLOP/s algorithmic = {flop / us / 1e6 / 157.3:.2f} of the fp32 MFMA peak; '
      f'rel-L2 vs direct {float((dW4 - dW).norm() / dW.norm()):.2e}')
# fp64 reference on a small problem
n2, c2, h2, w2 = 2, 40, 19, 23
a = torch.randn(n2, c2, h2, w2, dtype=torch.float64, requires_grad=False)
g = torch.randn(n2, 24, h2, w2, dtype=torch.float64)
wt = torch.zeros(24, c2, 3, 3, dtype=torch.float64, requires_grad=True)
torch.nn.functional.conv2d(a, wt, padding=1).backward(g)
from adaptivepnp_sci_amd import ops
a8, g8 = ops.to_c8(a.float().cuda()), ops.to_c8(g.float().cuda())
cin_p, cout_p = a8.shape[1] * 8, g8.shape[1] * 8
out = torch.empty(24, c2, 3, 3, device='cuda')
ws2 = torch.empty(lib.scipnp_conv3x3_wgrad_wino_workspace_floats(cin_p, cout_p, 7), device='cuda')
_lib.check(lib.scipnp_conv3x3_wgrad_wino(C.c_void_p(a8.data_ptr()), C.c_void_p(g8.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(ws2.data_ptr()), 7, n2, c2, 24, cin_p, cout_p, h2, w2, _lib.stream_ptr()), 'small')
ref = wt.grad
print(f'small ragged problem vs fp64 autograd: rel-L2 {float((out.cpu().double() - ref).norm() / ref.norm()):.2e}')

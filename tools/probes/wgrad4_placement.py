#!/usr/bin/env python3
"""GPU box: which SIMD does each wave of a conv3x3_wgrad_wino4_kernel workgroup land on?  (SCIPNP_W4G_DBG=8: every wave writes its
HW_ID register; bits 3:0 wave slot, 5:4 SIMD, 11:8 CU, 12 SH, 15:13 SE)"""
import ctypes as C, os, sys
os.environ['SCIPNP_W4G_DBG'] = '8'
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import _lib
lib = _lib.load()
n, c, h, w, ns = 8, 96, 256, 256, 28
act = torch.randn(n, c // 8, h, w, 8, device='cuda'); dz = torch.randn_like(act)
ws = torch.zeros(lib.scipnp_conv3x3_wgrad_wino4_workspace_floats(c, c, ns), device='cuda')
dW = torch.empty(c, c, 3, 3, device='cuda')
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
lib.scipnp_conv3x3_wgrad_wino4(p(act), p(dz), p(dW), p(ws), ns, n, c, c, c, c, h, w, _lib.stream_ptr())
torch.cuda.synchronize()
ids = ws[:ns * 9 * 8].view(torch.int32).cpu().numpy().reshape(-1, 8)
for wg in list(range(6)) + [100, 251]:
    print(f'workgroup {wg:3d}: ' + '  '.join(f'w{k}: simd {(int(v) >> 4) & 3} slot {int(v) & 15} cu {(int(v) >> 8) & 15} se {(int(v) >> 13) & 7}' for k, v in enumerate(ids[wg])))
import collections
pat = collections.Counter(tuple((int(v) >> 4) & 3 for v in row) for row in ids)
print('SIMD pattern of waves 0..7 -> workgroups:', dict(pat))

#!/usr/bin/env python3
"""GPU box, under rocprofv3 --pmc: the FFDNet body layer on the classic Winograd kernel, the persistent one and the
persistent one with everything but the MFMAs switched off (diag 63), 12 launches each -- for clock (GRBM_GUI_ACTIVE / time)
and matrix-pipe duty (SQ_VALU_MFMA_BUSY_CYCLES) per variant."""
import ctypes as C, os, sys
import torch
os.environ['SCIPNP_WINO_PERSISTENT'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import _lib, ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'lab'))
import lablib as diaglib  # noqa: E402  (lab/libscipnp_lab.so)
lib = diaglib.load()
n, c, h, w = 8, 96, 256, 256
g = torch.Generator().manual_seed(0)
x8 = ops.to_c8(torch.randn(n, c, h, w, generator=g).cuda())
pk = ops.pack_conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.randn(c, generator=g), Cin=c, Cout=c, device='cuda')
pb = ops.pack_conv3x3_wino_both(pk, c, c)
pwinop = diaglib.pack_winop(pk, c, c)
out = torch.empty_like(x8)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for _ in range(40):                                            # clocks up
    ops.conv3x3_c8w(x8, pb.w, c, relu=True, out=out)
torch.cuda.synchronize()
for _ in range(12):
    ops.conv3x3_c8w(x8, pb.w, c, relu=True, out=out)          # classic
for _ in range(12):
    diaglib.conv3x3_c8p(x8, pwinop, c, relu=True, out=out)            # persistent
for d in (63, 31, 15, 7):
    for _ in range(12):
        _lib.check(lib.scipnp_conv3x3_c8p_diag(P(x8), P(pwinop), P(out), n, c, c, h, w, 1, d, _lib.stream_ptr()), 'diag')
torch.cuda.synchronize()
print('done')

#!/usr/bin/env python3
"""GPU box: weight gradient of FFDNet's head (16 -> 96) and tail (96 -> 16) layers, F(2x2)-domain kernel (what the fp32 trainer
uses for them) against the F(4x4)-domain kernel (which pads the narrow side to its 32-channel block), at the tile's and the
full cube's size"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import _lib, ops, finetune
lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
g = torch.Generator().manual_seed(0)
for (n, h, w) in ((16, 128, 128), (8, 256, 256)):
    for (ci_r, co_r, ci, co) in ((13, 96, 16, 96), (96, 12, 96, 16), (96, 96, 96, 96)):
        x8 = ops.to_c8(torch.relu(torch.randn(n, ci_r, h, w, generator=g)).cuda())
        dz8 = ops.to_c8(torch.randn(n, co_r, h, w, generator=g).cuda())
        res = {}
        for name, fn, wsf, slabs in (('F(2x2)', lib.scipnp_conv3x3_wgrad_wino, lib.scipnp_conv3x3_wgrad_wino_workspace_floats, finetune._wino_slabs(ci)),
                                     ('F(4x4)', lib.scipnp_conv3x3_wgrad_wino4, lib.scipnp_conv3x3_wgrad_wino4_workspace_floats, finetune._wino4_slabs(ci, co))):
            ws = torch.empty(wsf(ci, co, slabs), device='cuda')
            dW = torch.empty(co_r, ci_r, 3, 3, device='cuda')
            for _ in range(3):
                _lib.check(fn(p(x8), p(dz8), p(dW), p(ws), slabs, n, ci_r, co_r, ci, co, h, w, s), name)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _lib.check(fn(p(x8), p(dz8), p(dW), p(ws), slabs, n, ci_r, co_r, ci, co, h, w, s), name)
            e1.record(); torch.cuda.synchronize()
            res[name] = (e0.elapsed_time(e1) / 10 * 1e3, slabs, dW.clone())
        d = float((res['F(2x2)'][2] - res['F(4x4)'][2]).norm() / res['F(2x2)'][2].norm())
        print(f'{n} x {ci:3d} -> {co:3d} @ {h}x{w}: ' + '  '.join(f'{k} {v[0]:7.1f} us ({v[1]} slabs)' for k, v in res.items()) + f'   rel diff {d:.1e}', flush=True)
print('done')

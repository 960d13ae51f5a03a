#!/usr/bin/env python3
"""GPU box, SCIPNP_LIB=build/variants/libscipnp_tvstamps.so (make -C adaptivepnp_sci_amd/csrc tvstamps): clock stamps of one
workgroup of the banded TV kernel (candidate form) on the ADMM-TV shape -- where the time of one Chambolle iteration goes (phase A:
divergence + `out`, barrier, phase B: gradient + dual update, barrier) for the workgroup's first and last wave, and the shader
clock against the 100 MHz reference clock."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import ops
dev = torch.device('cuda', 0)
for (C_, M, N) in ((32, 128, 128), (8, 256, 256)):
    x = torch.rand(C_, M, N, device=dev)
    b = torch.randn(C_, M, N, device=dev) * 0.1
    out = torch.empty_like(x)
    plan = ops.TvPlan(M, N, C_, 5, dev)
    warm = torch.zeros(64 + 2 * 2 * 64 + 64, dtype=torch.int32, device=dev)
    plan.stop_iter = warm
    for _ in range(20):
        ops.tv_chambolle(x, b, -1.0, out, plan, 0.1, kernel=4)
    torch.cuda.synchronize()
    # ONE stamped launch into a fresh buffer (scalar stores of several launches to the same words do not land in order)
    plan.stop_iter = torch.zeros(64 + 2 * 2 * 64 + 64, dtype=torch.int32, device=dev)      # [0,64): stop iterations; then 2 x 64 stamps
    torch.cuda.synchronize()
    ops.tv_chambolle(x, b, -1.0, out, plan, 0.1, kernel=4)
    torch.cuda.synchronize()
    st = plan.stop_iter[64:64 + 256].cpu().numpy().view(np.uint64).astype(np.int64)
    print(f'== {C_} planes of {M}x{N}')
    for w, name in ((0, 'first wave'), (1, 'last wave')):
        s = st[64 * w:64 * w + 64]
        t0 = s[0]
        ref = (s[61] - s[60]) / 100.0                       # us on the 100 MHz reference clock
        cyc = s[53] - s[0]
        print(f'{name}: kernel body {cyc} clocks = {ref:.2f} us -> {cyc / ref / 1e3:.3f} GHz; loads done +{s[1] - t0}')
        for it in range(5):
            a = s[8 + 8 * it: 8 + 8 * it + 5]
            if it < 4:
                print(f'  it {it}: start +{a[0] - t0:6d} | phase A {a[1] - a[0]:5d} | barrier {a[2] - a[1]:5d} | phase B {a[3] - a[2]:5d} | '
                      f'barrier {a[4] - a[3]:5d} | total {a[4] - a[0]:5d}')
            else:
                print(f'  it {it}: start +{a[0] - t0:6d} | phase A {a[1] - a[0]:5d}')
        print(f'  after loop +{s[50] - t0} | reduction {s[51] - s[50]} | barrier + sum {s[52] - s[51]} | end +{s[53] - t0}')
print('done')

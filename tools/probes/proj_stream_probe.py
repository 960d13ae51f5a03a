#!/usr/bin/env python3
"""GPU box: the plane-major projection kernel on an HBM-resident state (2048 x 2048 x 8: 570 MB per launch), hipGraph-free event timing of
40 back-to-back launches; run once per library variant (SCIPNP_LIB=build/variants/libscipnp_tvprojnt{1,2,3}.so: non-temporal loads /
stores / both)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adaptivepnp_sci_amd import ops  # noqa: E402

lib = os.environ.get('SCIPNP_LIB', 'product library')
for B, M in ((8, 1024), (8, 724), (8, 512), (8, 384), (16, 512), (16, 384), (8, 256)):
    th = torch.rand(B, 4, M, M, device='cuda')
    bb, ph = torch.rand_like(th), (torch.rand_like(th) > 0.5).float()
    yy, ps = torch.rand(4, M, M, device='cuda') * B / 2, torch.full((4, M, M), B / 2.0, device='cuda')
    xo = torch.empty_like(th)
    nbytes = 16.0 * (2 * M) ** 2 * B + 8.0 * (2 * M) ** 2
    ts = []
    for rep in range(5):
        for _ in range(5):
            ops.pm_project(th, bb, ph, yy, ps, 0, 1.0, 1.0, out=xo)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            ops.pm_project(th, bb, ph, yy, ps, 0, 1.0, 1.0, out=xo)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 40 * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print(f'{lib:44s} pm_project on {2 * M}x{2 * M}x{B} ({nbytes / 1e6:6.1f} MB per launch, back to back): {t:7.1f} us = {nbytes / t / 1e6:6.3f} TB/s')
    del th, bb, ph, yy, ps, xo

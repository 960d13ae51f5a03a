#!/usr/bin/env python3
"""GPU box (optionally under rocprofv3 --kernel-trace): the banded TV kernel in its candidate form on the ADMM-TV shape
(256x256x8 cube = 32 quarter-resolution planes of 128x128) for 2..5 inner iterations, 100 calls each -- the slope is the cost
of one Chambolle iteration, the intercept the kernel's fixed latency.  With a trace directory as argv[1] the script is the
post-processor: average duration of the band kernel per group of 100 calls."""
import csv, glob, os, sys
if len(sys.argv) > 1:
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    for key in ('tv_band_kernel', 'tv_stop', 'tv_select', 'pm_dual_project'):
        sel = [(e - s) / 1e3 for s, e, n in rows if key in n]
        groups = [sel[i:i + 100] for i in range(0, len(sel), 100)]
        print(key, len(sel), ' '.join(f'{sum(g) / len(g):.2f}' for g in groups if g))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import ops
dev = torch.device('cuda', 0)
shapes = ((32, 128, 128), (64, 128, 128), (8, 256, 256))
for (C_, M, N) in shapes:
    x = torch.rand(C_, M, N, device=dev)
    b = torch.randn(C_, M, N, device=dev) * 0.1
    out = torch.empty_like(x)
    for n_iter in (2, 3, 4, 5):
        plan = ops.TvPlan(M, N, C_, n_iter, dev)
        for _ in range(5):
            ops.tv_chambolle(x, b, -1.0, out, plan, 0.1, kernel=4)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(95):
                ops.tv_chambolle(x, b, -1.0, out, plan, 0.1, kernel=4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print(f'{C_} planes of {M}x{N}, n_iter={n_iter}: band candidates + selection {e0.elapsed_time(e1) / 95 * 1e3:.2f} us per call',
              flush=True)
        del g
print('done')

#!/usr/bin/env python3
"""GPU box: ADMM-TV (one-stage, 256x256x8) as unit batches of U = 1, 2, 4, 8, 16 cubes -- us per iteration of the whole batch,
per unit, and the speed-up per unit over the single-unit run (round 4 target: >= 3x at U = 8)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun

H = W = int(os.environ.get('TV_HW', 256))
B = 8
iqa = os.environ.get('TV_IQA', '1') == '1'
only = os.environ.get('TV_UNITS')
pr = [synth.make_problem(H, W, B, seed=i) for i in range(16)]
base = None
for U in ([int(only)] if only else [1, 2, 4, 8, 16]):
    kw = dict(X_orig=[p[2] for p in pr[:U]]) if iqa else {}
    run = (AdmmRun(pr[0][0], pr[0][1], 'tv', False, X_orig=pr[0][2] if iqa else None) if U == 1 else
           AdmmRun([p[0] for p in pr[:U]], [p[1] for p in pr[:U]], 'tv', False, units=U, **kw))
    for _ in range(10):
        run.step(0)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(100):
            run.step(0)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100 * 1e6)
    base = base or best
    print(f'U = {U:2d}: {best:7.1f} us per iteration of the batch, {best / U:6.2f} us per unit, {base / (best / U):5.2f}x per unit '
          f'(iqa {int(iqa)}, {H}x{W}x{B})', flush=True)

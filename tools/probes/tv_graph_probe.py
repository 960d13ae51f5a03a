#!/usr/bin/env python3
"""GPU box: the ADMM-TV iteration (256x256x8) as a host loop against 50 iterations captured into ONE hipGraph (bench.graph_timed)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
y, Phi, orig = synth.make_problem(256, 256, 8, 0)
for two in (False, True):
    for iqa in (True, False):
        run = AdmmRun(y, Phi, 'tv', two, X_orig=orig if iqa else None)
        for _ in range(10):
            run.step(0)
        g, h = bench.graph_timed(lambda: run.step(0), 50)
        print(f'{"two" if two else "one"}-stage, PSNR partials {int(iqa)}: hipGraph replay {g * 1e6:.2f} us per iteration, host loop {h * 1e6:.2f} us', flush=True)
print('done')

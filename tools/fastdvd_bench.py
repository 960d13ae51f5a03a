#!/usr/bin/env python3
"""GPU box: N two-stage ADMM + FastDVDnet iterations at 512x512x8 (for rocprofv3 kernel traces)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
from adaptivepnp_sci_amd.synth import synth_fastdvdnet as synth_fastdvdnet_weights
y, Phi, orig = synth.make_problem(512, 512, 8, 0)
run = AdmmRun(y, Phi, 'fastdvd_color', True, X_orig=orig, model=torch.nn.DataParallel(synth_fastdvdnet_weights(0)))
for _ in range(3):
    run.step(8 / 255)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get('FD_STEPS', 10))
for _ in range(n):
    run.step(8 / 255)
torch.cuda.synchronize()
print(f'FastDVDnet {run.eng.precision}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/iteration')

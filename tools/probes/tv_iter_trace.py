#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace: 200 ADMM-TV iterations at 256x256x8 with the dual update deferred (two launches per
iteration) and 200 without (four), for the per-kernel durations and the gaps between them."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
y, Phi, orig = synth.make_problem(256, 256, 8, 0)
for defer in ('1', '0'):
    os.environ['SCIPNP_TV_DEFER'] = defer
    run = AdmmRun(y, Phi, 'tv', True, X_orig=orig)
    for _ in range(200):
        run.step(0)
    run.flush()
    torch.cuda.synchronize()
print('done')

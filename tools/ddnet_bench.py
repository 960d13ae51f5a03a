#!/usr/bin/env python3
"""GPU box: N two-stage ADMM + FFDNet iterations with DDnet deep demosaicking at 512x512x8 (for rocprofv3 kernel traces)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
from adaptivepnp_sci_amd.nets import FFDNet
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'ffdnet_color_weights.npz'))
net = FFDNet(); net.load_state_dict({k: torch.from_numpy(g[k]) for k in g.files})
y, Phi, orig = synth.make_problem(512, 512, 8, 0)
run = AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=net, model_demosaic=synth.synth_ddnet(0))
for _ in range(3):
    run.step(25 / 255)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get('DD_STEPS', 10))
for _ in range(n):
    run.step(25 / 255)
torch.cuda.synchronize()
print(f'FFDNet + DDnet {run.eng.precision}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/iteration')

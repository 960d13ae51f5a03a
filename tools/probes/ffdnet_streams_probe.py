#!/usr/bin/env python3
"""GPU box: two-stage ADMM + FFDNet iteration time (512x512x8) with the network pass on one vs two HIP streams."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
from adaptivepnp_sci_amd.nets import FFDNet
g = np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden', 'ffdnet_color_weights.npz'))
net = FFDNet(); net.load_state_dict({k: torch.from_numpy(g[k]) for k in g.files})
y, Phi, orig = synth.make_problem(512, 512, 8, 0)
for prec in ('f16x3', 'f32'):
    for streams in ('1', '2', '1', '2'):
        os.environ['SCIPNP_STREAMS'] = streams
        run = AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=net, conv_precision=prec)
        for _ in range(30):
            run.step(25 / 255)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            run.step(25 / 255)
        torch.cuda.synchronize()
        print(f'{prec} streams={streams}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/iteration', run.psnr_all()[-1])

#!/usr/bin/env python3
"""GPU box, run once with the product library and once with SCIPNP_LIB=build/variants/libscipnp_w4nt.so: the FFDNet ADMM iteration
at sizes whose body-layer output per launch runs from 25 MB to 310 MB -- where do nt stores start to pay?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
from adaptivepnp_sci_amd.nets import FFDNet
g = np.load(os.path.join(ROOT, 'tests/golden/ffdnet_color_weights.npz'))
net = FFDNet(); net.load_state_dict({k: torch.from_numpy(g[k]) for k in g.files})
for (H, W, B) in ((256, 256, 8), (256, 256, 16), (384, 384, 8), (448, 448, 8), (512, 512, 8), (512, 512, 12), (640, 640, 8), (768, 768, 8), (1024, 1024, 8)):
    y, Phi, orig = synth.make_problem(H, W, B, 0)
    run = AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=net)
    for _ in range(8):
        run.step(25 / 255)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15):
            run.step(25 / 255)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 15 * 1e3)
    mb = B * 96 * (H // 2) * (W // 2) * 4 / 1e6
    print(f'{H}x{W}x{B}: body-layer output {mb:6.0f} MB ({mb / 2:5.0f} per launch): {best:8.3f} ms per iteration', flush=True)
    del run

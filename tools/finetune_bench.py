#!/usr/bin/env python3
"""GPU box: one FFDNet online-finetune event (2 Adam steps) at 512x512x8, for rocprofv3 kernel traces."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
from adaptivepnp_sci_amd.nets import FFDNet
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'ffdnet_color_weights.npz'))
sd = {k: torch.from_numpy(g[k]) for k in g.files}
H_, W_, B_ = (int(v) for v in os.environ.get('FT_SHAPE', '512,512,8').split(','))       # FT_SHAPE=256,256,16: a tile of configs[4]
y, Phi, orig = synth.make_problem(H_, W_, B_, 0)
if os.environ.get('FT_DENOISER', 'ffdnet') == 'fastdvd':
    from adaptivepnp_sci_amd.synth import synth_fastdvdnet as synth_fastdvdnet_weights
    net = torch.nn.DataParallel(synth_fastdvdnet_weights(0))
    run = AdmmRun(y, Phi, 'fastdvd_color', True, X_orig=orig, model=net, update_=True, lr_=2e-6, update_per_iter=2,
                  inital_iter=0, interval_iter=1)
else:
    net = FFDNet(); net.load_state_dict(sd)
    run = AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=net, update_=True, lr_=2e-6, update_per_iter=2,
                  inital_iter=0, interval_iter=1)
run.step(25 / 255)
ts = []
for _ in range(int(os.environ.get('FT_REPS', 2))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run.step(25 / 255)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('iteration with finetune: ' + ' '.join(f'{t:.1f}' for t in ts) + f' ms  (median {sorted(ts)[len(ts) // 2]:.1f})')

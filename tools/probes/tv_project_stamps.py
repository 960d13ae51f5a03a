#!/usr/bin/env python3
"""GPU box, build/variants/libscipnp_tvstamps.so (make -C adaptivepnp_sci_amd/csrc tvstamps): clock stamps of one wave of
pm_dual_project_spec_kernel inside the running ADMM-TV iteration at 256x256x8 -- where its 8 us go (loads issued, loads back,
stop test, arithmetic + stores, stores acknowledged).  The stamp buffer's address travels in SCIPNP_STAMP_PTR, which the
library reads at its first launch of the kernel: this script re-executes itself as a child once it has the buffer."""
import os, subprocess, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
buf = torch.zeros(64, dtype=torch.int64, device='cuda')
os.environ['SCIPNP_STAMP_PTR'] = hex(buf.data_ptr())
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
y, Phi, orig = synth.make_problem(256, 256, 8, 0)
for two in (False, True):
    run = AdmmRun(y, Phi, 'tv', two, X_orig=orig)
    for _ in range(30):
        run.step(0)
    torch.cuda.synchronize()
    buf.zero_()
    torch.cuda.synchronize()
    run.step(0)
    torch.cuda.synchronize()
    s = buf.cpu().numpy()
    names = ['entry', 'every load issued', 'loads back', 'stop test', 'arithmetic, stores issued', 'error sum', 'stores acknowledged']
    print(f'== {"two" if two else "one"}-stage: ' + ' | '.join(f'{n} +{int(s[i] - s[0])}' for i, n in enumerate(names)))
print('done')

#!/usr/bin/env python3
"""GPU box: cProfile of one FastDVDnet online-finetune event (host-side cost of the launch sequence)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
from adaptivepnp_sci_amd.synth import synth_fastdvdnet as synth_fastdvdnet_weights
y, Phi, orig = synth.make_problem(512, 512, 8, 0)
net = torch.nn.DataParallel(synth_fastdvdnet_weights(0))
run = AdmmRun(y, Phi, 'fastdvd_color', True, X_orig=orig, model=net, update_=True, lr_=2e-6, update_per_iter=2,
              inital_iter=0, interval_iter=1)
run.step(8 / 255); run.step(8 / 255); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
run.step(8 / 255)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)

import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from adaptivepnp_sci_amd import synth, twoStageAdmm_denoise_bayer
y, Phi, orig = synth.make_problem(512, 512, 8, 0)
def once():
    net = torch.nn.DataParallel(synth.synth_fastdvdnet(0))
    return twoStageAdmm_denoise_bayer(y, Phi, denoiser='fastdvd_color', iter_max=[18], sigma=[8 / 255], X_orig=orig, model_denoise=net,
                                      logf=io.StringIO(), lr_=2e-6, interval_iter=9, update_=True, update_per_iter=2, update_times=1)
once(); once()
pr = cProfile.Profile(); pr.enable(); t0=time.perf_counter(); once(); dt=time.perf_counter()-t0; pr.disable()
print('wall', dt)
pstats.Stats(pr).sort_stats('tottime').print_stats(14)

import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from adaptivepnp_sci_amd import synth, twoStageAdmm_denoise_bayer
from adaptivepnp_sci_amd.nets import FFDNet
g = np.load('/root/repo/tests/golden/ffdnet_color_weights.npz')
sd = {k: torch.from_numpy(g[k]) for k in g.files}
y, Phi, orig = synth.make_problem(256, 256, 16, 1)
def once(update=True):
    net = FFDNet(); net.load_state_dict(sd)
    return twoStageAdmm_denoise_bayer(y, Phi, denoiser='ffdnet_color', iter_max=[15, 6, 4], sigma=[25 / 255, 12 / 255, 6 / 255], X_orig=orig,
                                      model_denoise=net, logf=io.StringIO(), lr_=2e-6, interval_iter=15, update_=update, update_per_iter=2)
once(); once()
for upd in (False, True):
    torch.cuda.synchronize(); t0=time.perf_counter(); once(upd); torch.cuda.synchronize(); print('update', upd, 'wall %.1f ms' % ((time.perf_counter()-t0)*1e3))
pr = cProfile.Profile(); pr.enable(); once(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)

#!/usr/bin/env python3
"""GPU box: cProfile of one whole FFDNet / FastDVDnet reconstruction call (host-side fixed costs)."""
import cProfile, io, os, pstats, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import synth, twoStageAdmm_denoise_bayer
from adaptivepnp_sci_amd.nets import FFDNet
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'ffdnet_color_weights.npz'))
sd = {k: torch.from_numpy(g[k]) for k in g.files}
y, Phi, orig = synth.make_problem(512, 512, 8, 0)
which = os.environ.get('FT_DENOISER', 'ffdnet')


def once():
    if which == 'ffdnet':
        net = FFDNet(); net.load_state_dict(sd)
        return twoStageAdmm_denoise_bayer(y, Phi, denoiser='ffdnet_color', iter_max=[15, 6, 4], sigma=[25 / 255, 12 / 255, 6 / 255],
                                          X_orig=orig, model_denoise=net, logf=io.StringIO())
    from adaptivepnp_sci_amd.synth import synth_fastdvdnet as synth_fastdvdnet_weights
    net = torch.nn.DataParallel(synth_fastdvdnet_weights(0))
    return twoStageAdmm_denoise_bayer(y, Phi, denoiser='fastdvd_color', iter_max=[18], sigma=[8 / 255], X_orig=orig,
                                      model_denoise=net, logf=io.StringIO())


once(); once()
pr = cProfile.Profile()
pr.enable()
once()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)

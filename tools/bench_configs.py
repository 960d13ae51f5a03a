#!/usr/bin/env python3
"""Secondary timings on the GPU box: ms per ADMM iteration for the BASELINE configs other than the bench's
(config 0 ADMM-TV 256x256x8, config 2 FastDVDnet 512x512x8, config 4 tile 256x256x16), plus finetune events."""
import io, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
from adaptivepnp_sci_amd.nets import FFDNet


def timeit(run, sig, n, warm=3):
    for _ in range(warm):
        run.step(sig)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run.step(sig)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'ffdnet_color_weights.npz'))
sd = {k: torch.from_numpy(g[k]) for k in g.files}
for (H, W, B) in ((256, 256, 8), (512, 512, 8)):
    y, Phi, orig = synth.make_problem(H, W, B, 0)
    for two in (False, True):
        run = AdmmRun(y, Phi, 'tv', two, X_orig=orig)
        print(f'ADMM-TV {"two" if two else "one"}-stage {H}x{W}x{B}: {timeit(run, 0, 50):.3f} ms/iteration')
y, Phi, orig = synth.make_problem(512, 512, 8, 0)
tv = AdmmRun(y, Phi, 'tv', False)
for _ in range(20):
    tv.step(0)
warm = tv.result_mosaic()
for prec in ('f32', 'f16x3'):
    os.environ['SCIPNP_CONV_PRECISION'] = prec
    net = FFDNet(); net.load_state_dict(sd)
    run = AdmmRun(y, Phi, 'ffdnet_color', True, x0_bayer=warm, X_orig=orig, model=net)
    print(f'FFDNet {prec} 512x512x8: {timeit(run, 25/255, 20):.3f} ms/iteration')
    run = AdmmRun(y, Phi, 'ffdnet_color', True, x0_bayer=warm, X_orig=orig, model=net, update_=True, lr_=2e-6,
                  update_per_iter=2, inital_iter=0, interval_iter=1)
    run.step(25 / 255); torch.cuda.synchronize()
    t0 = time.perf_counter(); run.step(25 / 255); torch.cuda.synchronize()
    print(f'FFDNet {prec} iteration WITH online finetune (2 Adam steps): {(time.perf_counter() - t0) * 1e3:.1f} ms')
from adaptivepnp_sci_amd.synth import synth_fastdvdnet as synth_fastdvdnet_weights
fnet = torch.nn.DataParallel(synth_fastdvdnet_weights(0))
for prec in ('f32', 'f16x3'):
    os.environ['SCIPNP_CONV_PRECISION'] = prec
    run = AdmmRun(y, Phi, 'fastdvd_color', True, x0_bayer=warm, X_orig=orig, model=fnet)
    print(f'FastDVDnet {prec} 512x512x8: {timeit(run, 8/255, 5, 1):.3f} ms/iteration')
    run = AdmmRun(y, Phi, 'fastdvd_color', True, x0_bayer=warm, X_orig=orig, model=fnet, update_=True, lr_=2e-6,
                  update_per_iter=2, inital_iter=0, interval_iter=1)
    run.step(8 / 255); torch.cuda.synchronize()
    t0 = time.perf_counter(); run.step(8 / 255); torch.cuda.synchronize()
    print(f'FastDVDnet {prec} iteration WITH online finetune (2 Adam steps): {(time.perf_counter() - t0) * 1e3:.1f} ms')
from adaptivepnp_sci_amd.synth import synth_ddnet as synth_ddnet_weights
dd = synth_ddnet_weights(0)
os.environ['SCIPNP_CONV_PRECISION'] = 'f16x3'
net = FFDNet(); net.load_state_dict(sd)
run = AdmmRun(y, Phi, 'ffdnet_color', True, x0_bayer=warm, X_orig=orig, model=net, model_demosaic=dd)
print(f'FFDNet f16x3 + DDnet deep demosaicking 512x512x8: {timeit(run, 25/255, 10):.3f} ms/iteration')
y, Phi, orig = synth.make_problem(256, 256, 16, 0)
net = FFDNet(); net.load_state_dict(sd)
run = AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=net)
print(f'FFDNet f16x3 tile 256x256x16: {timeit(run, 25/255, 20):.3f} ms/iteration')

# whole adaptive reconstructions with the reference drivers' schedules (Beauty scene): wall time of the solver call
from adaptivepnp_sci_amd import twoStageAdmm_denoise_bayer
y, Phi, orig = synth.make_problem(512, 512, 8, 0)
os.environ['SCIPNP_CONV_PRECISION'] = 'f16x3'
for name, kw in (('FFDNet [15,6,4] its, finetune at k=15 (2 Adam steps)',
                  dict(denoiser='ffdnet_color', iter_max=[15, 6, 4], sigma=[25 / 255, 12 / 255, 6 / 255], lr_=2e-6,
                       interval_iter=15, update_=True, update_per_iter=2)),
                 ('FFDNet [15,6,4] its, no finetune',
                  dict(denoiser='ffdnet_color', iter_max=[15, 6, 4], sigma=[25 / 255, 12 / 255, 6 / 255])),
                 ('FastDVDnet [18] its, finetune at k=9 (2 Adam steps, update_times=1)',
                  dict(denoiser='fastdvd_color', iter_max=[18], sigma=[8 / 255], lr_=2e-6, interval_iter=9, update_=True,
                       update_per_iter=2, update_times=1)),
                 ('FastDVDnet [18] its, no finetune', dict(denoiser='fastdvd_color', iter_max=[18], sigma=[8 / 255]))):
    ts = []
    for rep in range(3):
        if kw['denoiser'] == 'ffdnet_color':
            model = FFDNet(); model.load_state_dict(sd)
        else:
            model = torch.nn.DataParallel(synth_fastdvdnet_weights(0))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        twoStageAdmm_denoise_bayer(y, Phi, x0_bayer=warm, X_orig=orig, model_denoise=model, logf=io.StringIO(), **kw)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f'whole reconstruction 512x512x8, {name}: {min(ts):.1f} ms (runs: {" ".join(f"{t:.0f}" for t in ts)})')

# configs[0]: ADMM-TV 256x256x8, 50 iterations, whole call; hipGraph replay of the iteration vs eager launches
from adaptivepnp_sci_amd import admm_denoise_bayer_demosaic_pre
y0, Phi0, orig0 = synth.make_problem(256, 256, 8, 0)
for mode in ('1', '0'):
    os.environ['SCIPNP_HIPGRAPH'] = mode
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        admm_denoise_bayer_demosaic_pre(y0, Phi0, 1, 0.01, 'tv', [50], False, [0], X_orig=orig0, logf=io.StringIO())
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f'whole ADMM-TV call 256x256x8, 50 iterations, hipGraph={mode}: {min(ts):.2f} ms (runs: {" ".join(f"{t:.1f}" for t in ts)})')

# configs[4] on one GPU: four 256x256x16 tiles of a 512x512x16 cube, FFDNet [15,6,4] with the online finetune (per-tile model
# copies), sequential vs two host threads / HIP streams
from adaptivepnp_sci_amd import shard
yt, Phit, origt = synth.make_problem(512, 512, 16, 1)
os.environ['SCIPNP_CONV_PRECISION'] = 'f16x3'


def solve_tile(args, model):
    y_t, Phi_t, _x0, orig_t = args
    res = twoStageAdmm_denoise_bayer(np.ascontiguousarray(y_t), np.ascontiguousarray(Phi_t), denoiser='ffdnet_color',
                                     iter_max=[15, 6, 4], sigma=[25 / 255, 12 / 255, 6 / 255], X_orig=np.ascontiguousarray(orig_t),
                                     model_denoise=model, logf=io.StringIO(), lr_=2e-6, interval_iter=15, update_=True,
                                     update_per_iter=2)
    return torch.from_numpy(res[1]).cuda()


model = FFDNet(); model.load_state_dict(sd)
outs = {}
for nstreams in (1, 2, 1, 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs[nstreams] = shard.reconstruct_tiled(yt, Phit, 256, solve_tile, torch.device('cuda'), orig=origt, model=model, streams=nstreams)
    torch.cuda.synchronize()
    print(f'four 256x256x16 tiles, FFDNet 25 its + finetune, {nstreams} stream(s): {(time.perf_counter() - t0) * 1e3:.1f} ms')
print('identical results:', bool(torch.equal(outs[1], outs[2])))

# configs[3] with several cubes per GPU: four independent 512x512x8 cubes, FFDNet [15,6,4] without finetune, one after
# the other vs two host threads / HIP streams
cubes = [synth.make_problem(512, 512, 8, s) for s in range(4)]


def solve_cube(args, model):
    y_c, Phi_c, orig_c = args
    res = twoStageAdmm_denoise_bayer(y_c, Phi_c, denoiser='ffdnet_color', iter_max=[15, 6, 4], sigma=[25 / 255, 12 / 255, 6 / 255],
                                     X_orig=orig_c, model_denoise=model, logf=io.StringIO())
    return torch.from_numpy(res[1]).cuda()


for nstreams in (1, 2, 1, 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    shard.reconstruct_sharded(cubes, solve_cube, (512, 512, 8), torch.device('cuda'), model=model, streams=nstreams)
    torch.cuda.synchronize()
    print(f'four 512x512x8 cubes, FFDNet 25 its, {nstreams} stream(s): {(time.perf_counter() - t0) * 1e3:.1f} ms')

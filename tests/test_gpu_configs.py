"""-m gpu: the BASELINE.json configurations at (or near) their full sizes, HIP path vs the CPU oracle on the same
seeded inputs, with the oracle leg bounded to seconds (few iterations)."""
import io

import numpy as np
import pytest
import torch

from conftest import load_gold, rel_l2

pytestmark = pytest.mark.gpu
REL_TOL, PSNR_TOL = 1e-5, 1e-4


class Trace:
    def __init__(self):
        self.it = []

    def __call__(self, k, mosaic):
        self.it.append(mosaic.cpu().numpy())


@pytest.fixture()
def solver():
    from adaptivepnp_sci_amd import solver as S
    yield S
    S.ITERATE_HOOK = None


def test_config0_admm_tv_256x256x8_50_iterations(solver):
    """configs[0]: ADMM-TV warm start (ADMM_TV_Warm_Start_save.py), 256x256x8, 50 iterations, every iterate."""
    from adaptivepnp_sci_amd import synth
    from oracle import solver as OS
    y, Phi, orig = synth.make_problem(256, 256, 8, seed=0)
    tr = Trace()
    solver.ITERATE_HOOK = tr
    xb, psnr_, ssim_, psnr_all = solver.admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [50], False, [0],
                                                                        X_orig=orig, logf=io.StringIO())
    o = OS.one_stage_admm(y, Phi, 1, 0.01, 'tv', [50], [0], X_orig=orig)
    worst = max(rel_l2(tr.it[k], o['x_iterates'][k]) for k in range(50))
    assert worst <= REL_TOL, worst
    assert np.abs(np.array(psnr_all) - np.array(o['psnr_all'])).max() <= PSNR_TOL


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_config1_ffdnet_512x512x8(solver, ffdnet_state_dict, precision, monkeypatch):
    monkeypatch.setenv('SCIPNP_FFDNET_PRECISION', precision)
    """configs[1]: two-stage ADMM + FFDNet-colour on one 512x512x8 cube (3 iterations against the oracle)."""
    from adaptivepnp_sci_amd import synth
    from adaptivepnp_sci_amd.nets import FFDNet
    from oracle import nets as ON
    from oracle import solver as OS
    y, Phi, orig = synth.make_problem(512, 512, 8, seed=0)
    warm = solver.admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [10], False, [0], logf=io.StringIO())[0]
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    onet = ON.OracleFFDNet()
    onet.load_state_dict(ffdnet_state_dict)
    onet.eval()
    tr = Trace()
    solver.ITERATE_HOOK = tr
    res = solver.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'ffdnet_color', [2, 1], False, [25 / 255, 12 / 255],
                                            x0_bayer=warm, X_orig=orig, model_denoise=net, logf=io.StringIO())
    with torch.no_grad():
        o = OS.two_stage_admm(y, Phi, 'ffdnet_color', [2, 1], [25 / 255, 12 / 255], x0_bayer=warm, X_orig=orig,
                              model_denoise=onet)
    for k in range(3):
        assert rel_l2(tr.it[k], o['theta_iterates'][k]) <= REL_TOL, k
    assert rel_l2(res[0], o['rgb']) <= REL_TOL
    assert np.abs(np.array(res[4]) - np.array(o['psnr_all'])).max() <= PSNR_TOL


def test_config3_eight_512x512x8_cubes_as_one_unit_batch(solver, ffdnet_state_dict):
    """configs[3] as written: 8 independent 512x512x8 cubes (seeds 0..7) solved as ONE unit batch on one GPU
    (what a rank does with its share at 1 GPU; `bench.py --cubes 8`): every unit bit-identical to its single-unit run for 3
    iterations, and unit 0 within the gates of the CPU oracle (reference loop two_stage_ADMM_Online_FFD_Warm.py:241-275)."""
    from adaptivepnp_sci_amd import synth
    from adaptivepnp_sci_amd.nets import FFDNet
    from adaptivepnp_sci_amd.solver import AdmmRun
    from oracle import nets as ON
    from oracle import solver as OS
    U = 8
    pr = [synth.make_problem(512, 512, 8, seed=i) for i in range(U)]
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    warm = []
    for y, Phi, _ in pr:                                     # TV warm start per cube, as the drivers do (:259-263)
        tv = AdmmRun(y, Phi, 'tv', False)
        for _ in range(10):
            tv.step(0)
        warm.append(tv.result_mosaic())
    sig = 25 / 255
    batch = AdmmRun([p[0] for p in pr], [p[1] for p in pr], 'ffdnet_color', True, x0_bayer=warm, X_orig=[p[2] for p in pr],
                    model=net, conv_precision='f32', units=U)
    its = []
    for k in range(3):
        batch.step(sig, last=(k == 2))
        its.append([m.clone() for m in batch.result_mosaic()])
    ps = batch.psnr_all()
    del batch
    for u in range(U):
        r = AdmmRun(pr[u][0], pr[u][1], 'ffdnet_color', True, x0_bayer=warm[u], X_orig=pr[u][2], model=net, conv_precision='f32')
        for k in range(3):
            r.step(sig, last=(k == 2))
            assert torch.equal(its[k][u], r.result_mosaic()), (u, k)
        assert np.abs(np.array(ps[u]) - np.array(r.psnr_all())).max() < 1e-9, u
        del r
    onet = ON.OracleFFDNet()
    onet.load_state_dict(ffdnet_state_dict)
    onet.eval()
    y, Phi, orig = pr[0]
    with torch.no_grad():
        o = OS.two_stage_admm(y, Phi, 'ffdnet_color', [3], [sig], x0_bayer=warm[0].cpu().numpy(), X_orig=orig, model_denoise=onet)
    worst = max(rel_l2(its[k][0].cpu().numpy(), o['theta_iterates'][k]) for k in range(3))
    assert worst <= REL_TOL, worst
    assert np.abs(np.array(ps[0]) - np.array(o['psnr_all'])).max() <= PSNR_TOL


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_config2_fastdvdnet_512x512x8(solver, precision, monkeypatch):
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', precision)
    """configs[2]: two-stage ADMM + FastDVDnet (5-frame temporal window), 512x512x8, rho = 0.55; 2 iterations."""
    from adaptivepnp_sci_amd import synth
    from oracle import solver as OS
    from oracle.nets import cpu_data_parallel, synth_fastdvdnet_weights
    y, Phi, orig = synth.make_problem(512, 512, 8, seed=1)
    warm = solver.admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [10], False, [0], logf=io.StringIO())[0]
    net = cpu_data_parallel(synth_fastdvdnet_weights(1))
    tr = Trace()
    solver.ITERATE_HOOK = tr
    res = solver.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'fastdvd_color', [2], False, [8 / 255], x0_bayer=warm,
                                            X_orig=orig, model_denoise=net, logf=io.StringIO())
    o = OS.two_stage_admm(y, Phi, 'fastdvd_color', [2], [8 / 255], x0_bayer=warm, X_orig=orig, model_denoise=net)
    for k in range(2):
        assert rel_l2(tr.it[k], o['theta_iterates'][k]) <= REL_TOL, k
    assert rel_l2(res[0], o['rgb']) <= REL_TOL


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_config2_fastdvdnet_driver_schedule_vs_reference_golden(solver, precision, monkeypatch):
    """The reference driver's own FastDVDnet schedule, free-running, against iterates captured FROM THE REFERENCE
    (tests/golden/fastdvdadmm_long_64x64x8.npz, tools/make_golden.py fastdvdlong): sigma 8/255 x 18 iterations, rho 0.55, online
    finetune lr 2e-6 x 2 Adam steps firing once at k = 9 (two_stage_ADMM_Online_FastDVD_Warm.py:68-75), 64 x 64 x 8, seeded
    synthetic weights -- EVERY one of the 18 iterates within 1e-5 relative L2, every per-iteration PSNR within 1e-4 dB,
    through the weight update.  (The full-size 18-iteration run below is checked through the agreement of the three
    convolution forms; this test is the parity evidence for the long schedule.)"""
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', precision)
    from oracle.nets import cpu_data_parallel, synth_fastdvdnet_weights
    g = load_gold('fastdvdadmm_long_64x64x8')
    net = cpu_data_parallel(synth_fastdvdnet_weights(0))
    np.random.seed(42)                                           # worker_init_fn(0) of the reference (utilspy.py:22-25)
    tr = Trace()
    solver.ITERATE_HOOK = tr
    res = solver.twoStageAdmm_denoise_bayer(g['y'], g['Phi'], 1, 0.01, 'fastdvd_color', [18], False, [8 / 255], x0_bayer=g['warm'],
                                            X_orig=g['orig'], model_denoise=net, logf=io.StringIO(), lr_=2e-6, inital_iter=1,
                                            interval_iter=9, update_=True, update_per_iter=2, update_times=1)
    assert len(tr.it) == 18
    errs = [rel_l2(tr.it[k], g['theta'][k]) for k in range(18)]
    assert max(errs) <= REL_TOL, errs
    assert np.abs(np.array(res[4]) - g['psnr_all']).max() <= PSNR_TOL
    assert rel_l2(res[1], g['final']) <= REL_TOL and rel_l2(res[0], g['rgb']) <= REL_TOL
    assert np.abs(np.array(res[2]) - g['psnr_frames']).max() <= PSNR_TOL
    # the one finetune event changed the weights by what the reference's Adam steps changed them
    w0 = synth_fastdvdnet_weights(0).state_dict()
    sd = net.state_dict()
    for key in ('temp1.inc.convblock.0.weight', 'temp2.outc.convblock.3.weight', 'temp2.downc0.convblock.0.weight'):
        d = float(torch.norm(sd['module.' + key].float() - w0[key].float()))
        want = float(g[key.replace('.', '_') + '_dnorm'])
        assert want > 0 and abs(d / want - 1) < 2e-2, (key, d, want)


def test_config2_fastdvdnet_full_driver_schedule(solver, monkeypatch):
    """configs[2] with the reference driver's whole schedule at full size (two_stage_ADMM_Online_FastDVD_Warm.py:68-75:
    sigma 8/255 x 18 iterations, rho 0.55, online finetune lr 2e-6 x 2 steps firing once at k = 9, update_times = 1) --
    too long for the CPU oracle inside the suite (18 x 6.5 s), so the full run is checked through size-independent
    properties: the three convolution forms (fp32 Winograd, fp32 direct, split-fp16), each already pinned per iterate
    against the oracle on the first iterations (test above) and on the 64x64x8 goldens incl. the finetune, must stay
    together over all 18 free-running iterations and through the weight update; PSNR must not collapse."""
    from adaptivepnp_sci_amd import synth
    from oracle.nets import cpu_data_parallel, synth_fastdvdnet_weights
    y, Phi, orig = synth.make_problem(512, 512, 8, seed=1)
    warm = solver.admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [10], False, [0], logf=io.StringIO())[0]
    runs = {}
    for form in ('f32-winograd', 'f32-direct', 'f16x3'):
        monkeypatch.setenv('SCIPNP_CONV_PRECISION', form.split('-')[0])
        monkeypatch.setenv('SCIPNP_F32_CONV', form.split('-')[1] if '-' in form else 'winograd')
        net = cpu_data_parallel(synth_fastdvdnet_weights(1))
        np.random.seed(42)                                       # the finetune's noise comes from the global NumPy RNG
        res = solver.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'fastdvd_color', [18], False, [8 / 255], x0_bayer=warm,
                                                X_orig=orig, model_denoise=net, logf=io.StringIO(), lr_=2e-6, inital_iter=1,
                                                interval_iter=9, update_=True, update_per_iter=2, update_times=1)
        runs[form] = (res[1], np.array(res[4]), {k: v.clone() for k, v in net.state_dict().items()})
    base = runs['f32-direct']
    assert len(base[1]) == 18 and np.isfinite(base[1]).all()
    for form in ('f32-winograd', 'f16x3'):
        mosaic, psnr, sd = runs[form]
        assert rel_l2(mosaic, base[0]) <= REL_TOL, (form, rel_l2(mosaic, base[0]))
        assert np.abs(psnr - base[1]).max() <= PSNR_TOL, (form, np.abs(psnr - base[1]).max())
    # the finetune event did change the weights, in all three runs alike (Adam steps ~ lr * sign(g))
    w0 = synth_fastdvdnet_weights(1).state_dict()
    key = 'temp2.outc.convblock.3.weight'
    d = {f: (runs[f][2]['module.' + key] - w0[key]).double() for f in runs}
    assert float(d['f32-direct'].abs().max()) > 0
    for f in ('f32-winograd', 'f16x3'):
        assert float((d[f] - d['f32-direct']).norm() / d['f32-direct'].norm()) < 5e-2, f


def test_config4_tile_256x256x16_with_online_finetune(solver, ffdnet_state_dict):
    """configs[4]: one 256x256 tile of the 1024x1024x16 colour cube (16 frames), FFDNet with online finetune firing
    once (gate at k = 2), per-tile model copy as in shard.reconstruct_sharded."""
    from adaptivepnp_sci_amd import synth
    from adaptivepnp_sci_amd.nets import FFDNet
    from oracle import nets as ON
    from oracle import solver as OS
    y, Phi, orig = synth.make_problem(256, 256, 16, seed=3)
    warm = solver.admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [10], False, [0], logf=io.StringIO())[0]
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    onet = ON.OracleFFDNet()
    onet.load_state_dict(ffdnet_state_dict)
    onet.eval()
    tr = Trace()
    solver.ITERATE_HOOK = tr
    kw = dict(lr_=2e-6, inital_iter=1, interval_iter=2, update_=True, update_per_iter=1)
    res = solver.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'ffdnet_color', [3], False, [25 / 255], x0_bayer=warm,
                                            X_orig=orig, model_denoise=net, logf=io.StringIO(), **kw)
    o = OS.two_stage_admm(y, Phi, 'ffdnet_color', [3], [25 / 255], x0_bayer=warm, X_orig=orig, model_denoise=onet,
                          lr=2e-6, inital_iter=1, interval_iter=2, update=True, update_per_iter=1)
    for k in range(3):
        assert rel_l2(tr.it[k], o['theta_iterates'][k]) <= REL_TOL, k
    assert res[1].shape == (256, 256, 16)


def test_config4_tiled_cube_matches_per_tile_oracle(solver, ffdnet_state_dict):
    """configs[4] end to end at reduced size: a 128x128x8 cube cut into four 64x64 patches, every patch reconstructed
    independently (own model copy, online finetune firing once), gathered and stitched; oracle = the reference solver
    called per patch."""
    from adaptivepnp_sci_amd import shard, synth
    from adaptivepnp_sci_amd.nets import FFDNet
    from oracle import nets as ON
    from oracle import solver as OS
    y, Phi, orig = synth.make_problem(128, 128, 8, seed=4)
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    kw = dict(lr_=2e-6, inital_iter=0, interval_iter=2, update_=True, update_per_iter=1)

    def solve(args, model):
        y_t, Phi_t, _x0, orig_t = args
        res = solver.twoStageAdmm_denoise_bayer(np.ascontiguousarray(y_t), np.ascontiguousarray(Phi_t), 1, 0.01,
                                                'ffdnet_color', [3], False, [25 / 255], X_orig=np.ascontiguousarray(orig_t),
                                                model_denoise=model, logf=io.StringIO(), **kw)
        return torch.from_numpy(res[1]).cuda()

    out = shard.reconstruct_tiled(y, Phi, 64, solve, torch.device('cuda'), orig=orig, model=net).cpu().numpy()
    assert out.shape == (128, 128, 8)
    for k0, w0 in ffdnet_state_dict.items():                      # the caller's model is untouched (per-tile copies)
        assert torch.equal(net.state_dict()[k0], w0)
    for (r, c), (y_t, Phi_t, _x0, orig_t) in zip(shard.tile_grid(128, 128, 64), shard.tile_cube(y, Phi, 64, orig=orig)):
        onet = ON.OracleFFDNet()
        onet.load_state_dict(ffdnet_state_dict)
        onet.eval()
        o = OS.two_stage_admm(np.ascontiguousarray(y_t), np.ascontiguousarray(Phi_t), 'ffdnet_color', [3], [25 / 255],
                              X_orig=np.ascontiguousarray(orig_t), model_denoise=onet, lr=2e-6, inital_iter=0,
                              interval_iter=2, update=True, update_per_iter=1)
        assert rel_l2(out[r:r + 64, c:c + 64], o['x_bayer']) <= REL_TOL, (r, c)


def test_config4_full_1024x1024x16_cube_16_tiles_driver_schedule(solver, ffdnet_state_dict):
    """configs[4] AS WRITTEN, on one GPU: the 1024x1024x16 colour cube cut into 16 patches of 256x256, every patch
    reconstructed with the reference driver's schedule (sigma [25,12,6]/255 x [15,6,4] iterations, lr 2e-6, 2 Adam steps
    per event, interval 15 -> one online-finetune event at k = 15) on its own copy of the model, gathered and stitched.
    Two of the 16 patches (a corner and an interior one) are checked against the oracle = the reference solver called per
    patch; the stitch is checked on every patch boundary by reconstructing the same patches alone."""
    from adaptivepnp_sci_amd import shard, synth
    from adaptivepnp_sci_amd.nets import FFDNet
    from oracle import nets as ON
    from oracle import solver as OS
    Hc, tile, Bc = 1024, 256, 16
    y, Phi, orig = synth.make_problem(Hc, Hc, Bc, seed=5)
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    sig, its = [25 / 255, 12 / 255, 6 / 255], [15, 6, 4]
    kw = dict(lr_=2e-6, inital_iter=1, interval_iter=15, update_=True, update_per_iter=2)
    finetuned = []

    def solve(args, model):
        y_t, Phi_t, _x0, orig_t = (None if a is None else np.ascontiguousarray(a) for a in args)
        res = solver.twoStageAdmm_denoise_bayer(y_t, Phi_t, 1, 0.01, 'ffdnet_color', its, False, sig, X_orig=orig_t,
                                                model_denoise=model, logf=io.StringIO(), **kw)
        finetuned.append(float((model.state_dict()['model.10.weight'] - ffdnet_state_dict['model.10.weight']).abs().max()))
        return torch.from_numpy(res[1]).cuda()

    out = shard.reconstruct_tiled(y, Phi, tile, solve, torch.device('cuda'), orig=orig, model=net).cpu().numpy()
    assert out.shape == (Hc, Hc, Bc) and len(finetuned) == 16 and min(finetuned) > 0      # every tile had its event
    for k0, w0 in ffdnet_state_dict.items():                      # the caller's model is untouched (per-tile copies)
        assert torch.equal(net.state_dict()[k0], w0)
    grid = shard.tile_grid(Hc, Hc, tile)
    units = shard.tile_cube(y, Phi, tile, orig=orig)
    for j in (0, 6):                                              # corner patch, interior patch
        (r, c), (y_t, Phi_t, _x0, orig_t) = grid[j], units[j]
        onet = ON.OracleFFDNet()
        onet.load_state_dict(ffdnet_state_dict)
        onet.eval()
        o = OS.two_stage_admm(np.ascontiguousarray(y_t), np.ascontiguousarray(Phi_t), 'ffdnet_color', its, sig,
                              X_orig=np.ascontiguousarray(orig_t), model_denoise=onet, lr=2e-6, inital_iter=1,
                              interval_iter=15, update=True, update_per_iter=2)
        assert rel_l2(out[r:r + tile, c:c + tile], o['x_bayer']) <= REL_TOL, (j, rel_l2(out[r:r + tile, c:c + tile], o['x_bayer']))
    assert np.isfinite(out).all() and 0.0 <= float(out.min()) and float(out.max()) <= 1.0      # (cold start: the iterates are clipped)


def test_largest_cube_1024x1024x16_untiled(solver, ffdnet_state_dict):
    """configs[4]'s cube reconstructed in one piece (no tiling): 16.8 M-element state tensors, 64 MiB per tensor --
    index arithmetic, grid limits and the summation-order emulation for B = 16 at full size, per iterate vs the oracle"""
    from adaptivepnp_sci_amd import synth
    from oracle import nets as ON
    from oracle import solver as OS
    y, Phi, orig = synth.make_problem(1024, 1024, 16, seed=11)
    tr = Trace()
    solver.ITERATE_HOOK = tr
    warm = solver.admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [2], False, [0], X_orig=orig, logf=io.StringIO())[0]
    ot = OS.one_stage_admm(y, Phi, 1, 0.01, 'tv', [2], [0], X_orig=orig)
    for k in range(2):
        assert rel_l2(tr.it[k], ot['x_iterates'][k]) <= REL_TOL, ('tv', k)
    onet = ON.OracleFFDNet()
    onet.load_state_dict(ffdnet_state_dict)
    onet.eval()
    tr = Trace()
    solver.ITERATE_HOOK = tr
    from adaptivepnp_sci_amd.nets import FFDNet
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    res = solver.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'ffdnet_color', [2], False, [25 / 255], x0_bayer=warm,
                                            X_orig=orig, model_denoise=net, logf=io.StringIO())
    with torch.no_grad():
        o = OS.two_stage_admm(y, Phi, 'ffdnet_color', [2], [25 / 255], x0_bayer=warm, X_orig=orig, model_denoise=onet)
    for k in range(2):
        assert rel_l2(tr.it[k], o['theta_iterates'][k]) <= REL_TOL, ('ffdnet', k, rel_l2(tr.it[k], o['theta_iterates'][k]))
    assert np.abs(np.array(res[4]) - np.array(o['psnr_all'])).max() <= PSNR_TOL


def test_config1_full_reference_schedule_with_online_finetune(solver, ffdnet_state_dict):
    """configs[1](ii) exactly as the reference driver runs it on a mid-scale scene (two_stage_ADMM_Online_FFD_Warm.py:71-76):
    sigma [25,12,6]/255 x [15,6,4] iterations, lr 2e-6, update_per_iter 2, interval_iter 15 -> one finetune event at k = 15,
    512x512x8, free-running against the CPU oracle: final iterate, every PSNR of the trace, the finetuned weights"""
    from adaptivepnp_sci_amd import synth
    from adaptivepnp_sci_amd.nets import FFDNet
    from oracle import nets as ON
    from oracle import solver as OS
    y, Phi, orig = synth.make_problem(512, 512, 8, seed=2)
    warm = solver.admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [40], False, [0], logf=io.StringIO())[0]
    sched = dict(sig=[25 / 255, 12 / 255, 6 / 255], its=[15, 6, 4])
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    res = solver.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'ffdnet_color', sched['its'], False, sched['sig'], x0_bayer=warm,
                                            X_orig=orig, model_denoise=net, logf=io.StringIO(), lr_=2e-6, interval_iter=15,
                                            update_=True, update_per_iter=2)
    onet = ON.OracleFFDNet()
    onet.load_state_dict(ffdnet_state_dict)
    onet.eval()
    o = OS.two_stage_admm(y, Phi, 'ffdnet_color', sched['its'], sched['sig'], x0_bayer=warm, X_orig=orig, model_denoise=onet,
                          lr=2e-6, inital_iter=1, interval_iter=15, update=True, update_per_iter=2)
    assert rel_l2(res[1], o['x_bayer']) <= REL_TOL, rel_l2(res[1], o['x_bayer'])
    assert len(res[4]) == 25 and np.abs(np.array(res[4]) - np.array(o['psnr_all'])).max() <= PSNR_TOL
    sd, osd = net.state_dict(), o['model'].state_dict()
    k0 = 'model.10.weight'
    d_got, d_ref = (sd[k0] - ffdnet_state_dict[k0]).numpy(), (osd[k0] - ffdnet_state_dict[k0]).numpy()
    assert np.abs(d_ref).max() > 0 and rel_l2(d_got, d_ref) < 2e-2


def _full512_case(name):
    """inputs of a tests/golden/full512_*.npz run: the synthetic problem of its seed and the oracle's 40-iteration ADMM-TV warm start
    (what tools/make_golden.py handed the reference; checked against the digest the fixture carries)"""
    import hashlib
    from adaptivepnp_sci_amd import synth
    from oracle import solver as OS
    g = load_gold(name)
    y, Phi, orig = synth.make_problem(512, 512, 8, seed=int(g['seed']))
    warm = OS.one_stage_admm(y, Phi, 1, 0.01, 'tv', [40], [0])['x_bayer']
    assert hashlib.sha256(np.ascontiguousarray(warm).tobytes()).digest() == bytes(g['warm_sha']), 'warm start differs from the one the reference was given'
    return g, y, Phi, orig, warm


def test_config1_full_schedule_vs_reference_capture(solver, ffdnet_state_dict):
    """configs[1](ii) at BASELINE's size against THE REFERENCE ITSELF (tests/golden/full512_ffdnet_schedule.npz: captured by
    tools/make_golden.py full512ffd from the imported reference on CPU): 512x512x8, sigma [25,12,6]/255 x [15,6,4], online
    finetune lr 2e-6 x 2 steps firing once at k = 15 (two_stage_ADMM_Online_FFD_Warm.py:62-76,260-269) -- the final mosaic
    within 1e-5 relative L2, the PSNR of every one of the 25 free-running iterations and of every frame within 1e-4 dB"""
    from adaptivepnp_sci_amd.nets import FFDNet
    g, y, Phi, orig, warm = _full512_case('full512_ffdnet_schedule')
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    res = solver.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'ffdnet_color', [int(v) for v in g['its']], False, [float(v) for v in g['sig']],
                                            x0_bayer=warm, X_orig=orig, model_denoise=net, logf=io.StringIO(), lr_=2e-6, inital_iter=1,
                                            interval_iter=15, update_=True, update_per_iter=2)
    assert len(res[4]) == 25
    assert np.abs(np.array(res[4]) - g['psnr_all']).max() <= PSNR_TOL, np.abs(np.array(res[4]) - g['psnr_all']).max()
    assert rel_l2(res[1], g['final']) <= REL_TOL, rel_l2(res[1], g['final'])
    assert np.abs(np.array(res[2]) - g['psnr_frames']).max() <= PSNR_TOL


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_config2_full_schedule_vs_reference_capture(solver, precision, monkeypatch):
    """configs[2] at BASELINE's size against THE REFERENCE ITSELF (tests/golden/full512_fastdvd_schedule.npz, tools/make_golden.py
    full512fastdvd): 512x512x8, sigma 8/255 x 18 iterations, rho 0.55, online finetune lr 2e-6 x 2 Adam steps firing once at
    k = 9, update_times 1 (two_stage_ADMM_Online_FastDVD_Warm.py:68-75,295-304), seeded synthetic weights (model.pth is not in the
    snapshot) -- final mosaic within 1e-5 relative L2, all 18 PSNR values and the per-frame PSNR within 1e-4 dB"""
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', precision)
    from oracle.nets import cpu_data_parallel, synth_fastdvdnet_weights
    g, y, Phi, orig, warm = _full512_case('full512_fastdvd_schedule')
    net = cpu_data_parallel(synth_fastdvdnet_weights(0))
    np.random.seed(42)                                           # worker_init_fn(0) of the reference (utilspy.py:22-25)
    res = solver.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'fastdvd_color', [18], False, [8 / 255], x0_bayer=warm, X_orig=orig,
                                            model_denoise=net, logf=io.StringIO(), lr_=2e-6, inital_iter=1, interval_iter=9,
                                            update_=True, update_per_iter=2, update_times=1)
    assert len(res[4]) == 18
    assert np.abs(np.array(res[4]) - g['psnr_all']).max() <= PSNR_TOL, np.abs(np.array(res[4]) - g['psnr_all']).max()
    assert rel_l2(res[1], g['final']) <= REL_TOL, rel_l2(res[1], g['final'])
    assert np.abs(np.array(res[2]) - g['psnr_frames']).max() <= PSNR_TOL


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_deep_demosaicking_256x256x8(solver, ffdnet_state_dict, precision, monkeypatch):
    """the reference drivers' default mode (deep_demosaicking=True) at a size with many tiles per layer: DDnet + FFDNet,
    three iterations, per iterate against the oracle (synthetic DDnet weights)"""
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', precision)
    from adaptivepnp_sci_amd import synth
    from adaptivepnp_sci_amd.nets import FFDNet
    from oracle import nets as ON
    from oracle import solver as OS
    y, Phi, orig = synth.make_problem(256, 256, 8, seed=6)
    warm = solver.admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [10], False, [0], logf=io.StringIO())[0]
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    tr = Trace()
    solver.ITERATE_HOOK = tr
    res = solver.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'ffdnet_color', [2, 1], False, [25 / 255, 12 / 255], x0_bayer=warm,
                                            X_orig=orig, model_denoise=net, model_demosaic=synth.synth_ddnet(0),
                                            logf=io.StringIO())
    onet = ON.OracleFFDNet()
    onet.load_state_dict(ffdnet_state_dict)
    onet.eval()
    with torch.no_grad():
        o = OS.two_stage_admm(y, Phi, 'ffdnet_color', [2, 1], [25 / 255, 12 / 255], x0_bayer=warm, X_orig=orig, model_denoise=onet,
                              model_demosaic=ON.synth_ddnet_weights(0))
    for k in range(3):
        assert rel_l2(tr.it[k], o['theta_iterates'][k]) <= REL_TOL, (k, rel_l2(tr.it[k], o['theta_iterates'][k]))
    assert rel_l2(res[0], o['rgb']) <= REL_TOL

"""-m gpu: UNIT BATCHES (round 4) -- U independent problems of one shape stepped by ONE launch sequence (AdmmRun(units=U),
state layout [B][U][4][M][N]; the reference loops its measurements one after the other,
two_stage_ADMM_Online_FFD_Warm.py:241-275).  The bar: every unit's iterate is BIT-IDENTICAL to its own single-unit run
(same kernels, same per-pixel / per-frame / per-plane arithmetic), per-iteration PSNR within 1e-9 dB (only the association of
the fp64 partial sums may differ), through split() and an online-finetune event on per-unit weights."""
import copy
import io

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def make_ffdnet(sd):
    from adaptivepnp_sci_amd.nets import FFDNet
    net = FFDNet()
    net.load_state_dict(sd)
    return net


def problems(n, H, W, B, seed0=0):
    from adaptivepnp_sci_amd import synth
    return [synth.make_problem(H, W, B, seed=seed0 + i) for i in range(n)]


@pytest.mark.parametrize('two_stage', [False, True])
@pytest.mark.parametrize('shape,U', [((64, 64, 8), 3), ((256, 256, 8), 8), ((64, 128, 5), 2), ((128, 128, 3), 6)])
def test_admm_tv_unit_batch_equals_single_unit_runs(two_stage, shape, U):
    """ADMM-TV, both solvers: the batch's two-launch iteration (fused dual update + projection over 4 M N U pixels, banded TV over
    4 B U planes) against U single-unit runs, iterate by iterate; deferred and flushed rows both appear (flush at k = 4)"""
    from adaptivepnp_sci_amd.solver import AdmmRun
    H, W, B = shape
    pr = problems(U, H, W, B, seed0=3)
    single = [AdmmRun(y, Phi, 'tv', two_stage, X_orig=orig) for y, Phi, orig in pr]
    batch = AdmmRun([p[0] for p in pr], [p[1] for p in pr], 'tv', two_stage, X_orig=[p[2] for p in pr], units=U)
    for k in range(9):
        for r in single:
            r.step(0)
        batch.step(0)
        if k in (4, 8):
            got = batch.result_mosaic()
            for u in range(U):
                assert torch.equal(got[u], single[u].result_mosaic()), (k, u)
    ps = batch.psnr_all()
    for u in range(U):
        want = single[u].psnr_all()
        assert len(ps[u]) == 9 and np.abs(np.array(ps[u]) - np.array(want)).max() < 1e-9, u
        fp, fs = batch.final_report()[u], single[u].final_report()
        assert np.allclose(fp[0], fs[0], atol=1e-9) and np.allclose(fp[1], fs[1], atol=1e-12)


def test_admm_tv_unit_batch_without_ground_truth_and_one_call_per_iteration():
    """no X_orig: no partial rows at all; the batch still issues ONE scipnp_admm_tv_iterate per iteration for all units"""
    from adaptivepnp_sci_amd.solver import AdmmRun
    pr = problems(4, 96, 64, 8, seed0=11)
    single = [AdmmRun(y, Phi, 'tv', False) for y, Phi, _ in pr]
    batch = AdmmRun([p[0] for p in pr], [p[1] for p in pr], 'tv', False, units=4)
    for _ in range(6):
        for r in single:
            r.step(0)
        batch.step(0)
    got = batch.result_mosaic()
    for u in range(4):
        assert torch.equal(got[u], single[u].result_mosaic())
    assert batch.psnr_all() == [[], [], [], []]


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_ffdnet_unit_batch_equals_single_unit_runs(ffdnet_state_dict, precision):
    """two-stage ADMM + FFDNet-colour, 3 units of 64 x 96 x 8 sharing the weights: projection over the units' pixels, Malvar /
    network / dual updates over B U frames -- iterates bit-identical per unit, PSNR rows cut at unit boundaries"""
    from adaptivepnp_sci_amd.solver import AdmmRun
    U = 3
    pr = problems(U, 64, 96, 8, seed0=21)
    net = make_ffdnet(ffdnet_state_dict)
    warm = []
    for y, Phi, _ in pr:
        tv = AdmmRun(y, Phi, 'tv', False)
        for _ in range(8):
            tv.step(0)
        warm.append(tv.result_mosaic())
    single = [AdmmRun(y, Phi, 'ffdnet_color', True, x0_bayer=warm[i], X_orig=orig, model=net, conv_precision=precision)
              for i, (y, Phi, orig) in enumerate(pr)]
    batch = AdmmRun([p[0] for p in pr], [p[1] for p in pr], 'ffdnet_color', True, x0_bayer=warm, X_orig=[p[2] for p in pr],
                    model=net, conv_precision=precision, units=U)
    for k, sig in enumerate([50 / 255, 50 / 255, 25 / 255, 12 / 255]):
        for r in single:
            r.step(sig, last=(k == 3))
        batch.step(sig, last=(k == 3))
        got = batch.result_mosaic()
        for u in range(U):
            assert torch.equal(got[u], single[u].result_mosaic()), (k, u)
    ps = batch.psnr_all()
    for u in range(U):
        assert np.abs(np.array(ps[u]) - np.array(single[u].psnr_all())).max() < 1e-9
    batch.check_overflow()
    # the denoised RGB frames of the last iteration, frame f = t U + u
    rgb = batch.out_rgb.view(8, U, 3, 64, 96)
    for u in range(U):
        assert torch.equal(rgb[:, u], single[u].out_rgb)


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
@pytest.mark.parametrize('close_form', [False, True])
def test_fastdvdnet_unit_batch_equals_single_unit_runs(precision, close_form):
    """two-stage ADMM + FastDVDnet, 3 units of 64 x 64 x 8 sharing the (synthetic) weights: the 5-frame temporal windows stay
    inside a unit (frames t U + u, neighbours +- U and +- 2U through the two DenBlock stages) -- iterates, PSNR rows and
    denoised frames bit-identical per unit; then split() and two more iterations on per-unit runs"""
    from adaptivepnp_sci_amd.solver import AdmmRun
    from oracle.nets import cpu_data_parallel, synth_fastdvdnet_weights
    U = 3
    pr = problems(U, 64, 64, 8, seed0=41)
    net = cpu_data_parallel(synth_fastdvdnet_weights(0))
    kw = dict(model=net, conv_precision=precision, close_form_demosaic=close_form)
    single = [AdmmRun(y, Phi, 'fastdvd_color', True, X_orig=orig, **kw) for y, Phi, orig in pr]
    batch = AdmmRun([p[0] for p in pr], [p[1] for p in pr], 'fastdvd_color', True, X_orig=[p[2] for p in pr], units=U, **kw)
    for k in range(3):
        for r in single:
            r.step(8 / 255)
        batch.step(8 / 255)
        got = batch.result_mosaic()
        for u in range(U):
            assert torch.equal(got[u], single[u].result_mosaic()), (k, u)
    parts = batch.split()
    for k in range(3, 5):
        for r in single:
            r.step(8 / 255, last=(k == 4))
        for p in parts:
            p.step(8 / 255, last=(k == 4))
    for u in range(U):
        assert torch.equal(parts[u].result_mosaic(), single[u].result_mosaic()), u
        assert torch.equal(parts[u].out_rgb, single[u].out_rgb), u
        assert np.abs(np.array(parts[u].psnr_all()) - np.array(single[u].psnr_all())).max() < 1e-9
    batch.check_overflow()


@pytest.mark.parametrize('denoiser', ['ffdnet_color', 'fastdvd_color'])
def test_deep_demosaicking_unit_batch_equals_single_unit_runs(ffdnet_state_dict, denoiser):
    """model_demosaic= (DDnet, the reference drivers' default) in a unit batch: stage 1's temporal triplets are gathered inside
    a unit (DDnetEngine(units=): only the index tables change), frames [B][U] -- bit-identical per unit for both CNN denoisers"""
    from adaptivepnp_sci_amd.solver import AdmmRun
    from oracle.nets import cpu_data_parallel, synth_ddnet_weights, synth_fastdvdnet_weights
    U = 2
    pr = problems(U, 64, 64, 8, seed0=51)
    net = make_ffdnet(ffdnet_state_dict) if denoiser == 'ffdnet_color' else cpu_data_parallel(synth_fastdvdnet_weights(0))
    dd = synth_ddnet_weights(0)
    kw = dict(model=net, model_demosaic=dd, conv_precision='f32')
    single = [AdmmRun(y, Phi, denoiser, True, X_orig=orig, **kw) for y, Phi, orig in pr]
    batch = AdmmRun([p[0] for p in pr], [p[1] for p in pr], denoiser, True, X_orig=[p[2] for p in pr], units=U, **kw)
    for k in range(3):
        for r in single:
            r.step(12 / 255, last=(k == 2))
        batch.step(12 / 255, last=(k == 2))
        got = batch.result_mosaic()
        for u in range(U):
            assert torch.equal(got[u], single[u].result_mosaic()), (k, u)
    ps = batch.psnr_all()
    for u in range(U):
        assert np.abs(np.array(ps[u]) - np.array(single[u].psnr_all())).max() < 1e-9
        assert torch.equal(batch.out_rgb.view(8, U, 3, 64, 64)[:, u], single[u].out_rgb)


def test_unit_batch_split_before_the_finetune_event(ffdnet_state_dict):
    """BASELINE configs[4] shape of work: the tiles of a cube share the weights until the online finetune fires, then every
    tile trains ITS OWN copy.  Batch for k < 3, split(), event at k = 3 on per-unit models -> identical to single-unit runs
    with the finetune enabled from the start; stepping the batch INTO the event is refused."""
    from adaptivepnp_sci_amd import _lib
    from adaptivepnp_sci_amd.solver import AdmmRun
    U = 2
    pr = problems(U, 64, 64, 8, seed0=31)
    kw = dict(update_=True, lr_=2e-6, update_per_iter=1, inital_iter=1, interval_iter=3)
    nets_single = [make_ffdnet(ffdnet_state_dict) for _ in range(U)]
    single = [AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=nets_single[i], conv_precision='f32', **kw)
              for i, (y, Phi, orig) in enumerate(pr)]
    shared = make_ffdnet(ffdnet_state_dict)
    batch = AdmmRun([p[0] for p in pr], [p[1] for p in pr], 'ffdnet_color', True, X_orig=[p[2] for p in pr], model=shared,
                    conv_precision='f32', units=U, **kw)
    for k in range(3):
        for r in single:
            r.step(25 / 255)
        batch.step(25 / 255)
    with pytest.raises(_lib.ScipnpError, match='split'):
        batch.step(25 / 255)                                   # k = 3 fires the gate
    nets = [copy.deepcopy(shared) for _ in range(U)]
    parts = batch.split(models=nets)
    assert len(parts) == U and all(p.k == 3 and p.U == 1 for p in parts)
    for k in range(3, 5):
        for r in single:
            r.step(25 / 255)
        for p in parts:
            p.step(25 / 255)
    for u in range(U):
        assert torch.equal(parts[u].result_mosaic(), single[u].result_mosaic()), u
        assert np.abs(np.array(parts[u].psnr_all()) - np.array(single[u].psnr_all())).max() < 1e-9
        a, b = nets[u].state_dict(), nets_single[u].state_dict()
        assert all(torch.equal(a[k_], b[k_]) for k_ in a)
        assert not torch.equal(a['model.2.weight'], ffdnet_state_dict['model.2.weight'])      # the event did train


def test_part_lanes_step_the_split_runs_concurrently_with_identical_results(ffdnet_state_dict):
    """solver.PartLanes (round 5): after split() the per-unit runs are stepped on host threads with a HIP stream each (the
    units are independent; the finetune event's loss read-back and weight write-back stall only their own lane) -- every unit
    bit-identical to stepping the parts one after the other, trained weights included; the caller's stream continues behind
    all lanes (results read on it right after)"""
    from adaptivepnp_sci_amd.solver import AdmmRun, PartLanes
    U = 5
    pr = problems(U, 64, 64, 8, seed0=77)
    kw = dict(update_=True, lr_=2e-6, update_per_iter=2, inital_iter=1, interval_iter=2)

    def build():
        shared = make_ffdnet(ffdnet_state_dict)
        batch = AdmmRun([p[0] for p in pr], [p[1] for p in pr], 'ffdnet_color', True, X_orig=[p[2] for p in pr], model=shared,
                        conv_precision='f32', units=U, **kw)
        for _ in range(2):
            batch.step(25 / 255)
        nets = [copy.deepcopy(shared) for _ in range(U)]
        return batch.split(models=nets), nets
    seq, nets_seq = build()
    par, nets_par = build()
    lanes = PartLanes(par, lanes=3)
    assert lanes.n == 3
    for k in range(2, 5):                                      # k = 2 and k = 4 fire the event (two Adam steps each)
        for p in seq:
            p.step(25 / 255)
        lanes.step(25 / 255)
    lanes.close()
    for u in range(U):
        assert torch.equal(par[u].result_mosaic(), seq[u].result_mosaic()), u
        assert np.array_equal(np.array(par[u].psnr_all()), np.array(seq[u].psnr_all()))
        a, b = nets_par[u].state_dict(), nets_seq[u].state_dict()
        assert all(torch.equal(a[k_], b[k_]) for k_ in a)
        assert not torch.equal(a['model.2.weight'], ffdnet_state_dict['model.2.weight'])
    one = PartLanes(seq[:1], lanes=4)                          # a single part: no threads, no streams
    assert one.n == 1 and one.pool is None
    one.step(25 / 255)


def test_unit_batch_argument_errors(ffdnet_state_dict):
    from adaptivepnp_sci_amd.solver import AdmmRun
    from oracle.nets import cpu_data_parallel, synth_fastdvdnet_weights
    pr = problems(2, 32, 32, 4)
    ys, Phis = [p[0] for p in pr], [p[1] for p in pr]
    with pytest.raises(ValueError, match='sequences of 3'):
        AdmmRun(ys, Phis, 'tv', False, units=3)
    with pytest.raises(ValueError, match='unit batches'):
        AdmmRun(ys, Phis, 'tv', False, Phi_sum=torch.ones(32, 32), units=2)
    with pytest.raises(ValueError, match='shape of the first'):
        AdmmRun([ys[0], ys[1][:16]], [Phis[0], Phis[1][:16]], 'tv', False, units=2)
    with pytest.raises(ValueError, match='2048'):               # 24 x 24 mosaic: partial blocks would straddle units
        p3 = problems(2, 24, 24, 4)
        AdmmRun([p[0] for p in p3], [p[1] for p in p3], 'tv', False, X_orig=[p[2] for p in p3], units=2)


def test_a_solve_keeps_its_configuration_whatever_the_environment_says(ffdnet_state_dict, monkeypatch):
    """AdmmRun(config=...) (adaptivepnp_sci_amd.config, round 5): the kernel forms of a solve are fixed at construction and used by
    every step -- a later environment change or an enclosing full configuration does not reach it, an explicit field override
    does unless the constructor pinned that field; the three fp32 forms and f16x3 agree with each other within the gates"""
    from adaptivepnp_sci_amd import config, ops
    from adaptivepnp_sci_amd.solver import AdmmRun
    (y, Phi, orig), = problems(1, 64, 64, 8, seed0=61)
    net = make_ffdnet(ffdnet_state_dict)
    forms = {'f4': config.Config(), 'f2': config.Config(wino_f4=False), 'direct': config.Config(f32_form='direct'),
             'f16x3': config.Config(precision='f16x3')}
    runs = {k: AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=net, config=c) for k, c in forms.items()}
    assert runs['f16x3'].eng.precision == 'f16x3' and runs['f4'].eng.precision == 'f32' and runs['direct'].eng.f32_form == 'direct'
    assert runs['f4'].eng.packed_wino[1].f4 is not None and runs['f2'].eng.packed_wino[1].f4 is None
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', 'f16x3')          # the process default changes under the runs' feet ...
    monkeypatch.setenv('SCIPNP_WINO_F4', '0')
    seen = {}
    try:
        with config.use(config.Config(f32_form='direct')):        # ... and so does the enclosing full configuration
            for k, r in runs.items():
                ops.LAUNCH_LOG = log = []
                r.step(25 / 255)
                seen[k] = {e[0] for e in log if e[0].startswith('conv3x3')}
    finally:
        ops.LAUNCH_LOG = None
    assert seen['f4'] == {'conv3x3_c8w4_kernel', 'conv3x3_c8w_kernel'}                  # head + body on F(4x4), tail on F(2x2)
    assert seen['f2'] == {'conv3x3_c8w_kernel'}
    assert seen['direct'] == {'conv3x3_c8_kernel'}
    assert seen['f16x3'] == {'conv3x3_c8s_kernel'}
    ref = runs['direct'].result_mosaic()
    for k in ('f4', 'f2', 'f16x3'):
        d = float((runs[k].result_mosaic() - ref).norm() / ref.norm())
        assert d < 1e-5, (k, d)
    # a run built WITHOUT config= follows explicit field overrides of the caller
    free = AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=net, conv_precision='f32')
    log2 = []
    ops.LAUNCH_LOG = log2
    try:
        with config.use(wino_f4=True, precision='f16x3'):         # wino_f4 reaches in; precision was pinned by conv_precision=
            free.step(25 / 255)
    finally:
        ops.LAUNCH_LOG = None
    assert 'conv3x3_c8s_kernel' not in {e[0] for e in log2} and free.eng.precision == 'f32'

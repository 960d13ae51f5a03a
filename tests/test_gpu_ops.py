"""-m gpu: every HIP kernel, called through the C ABI (ctypes), against the CPU oracle and the
golden vectors generated from the reference.  Tolerances are written next to each check:
streaming kernels are expected BIT-EXACT (same fp32 operations in the same order as the
reference's PyTorch expressions); the Malvar correlations, TV and the MFMA convolutions differ from
the CPU libraries only in summation order -> tight relative-L2 bounds."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_gold, rel_l2

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def planes_to_state(p):      # (M,N,B,4) -> [B][4][M][N]
    return p.permute(2, 3, 0, 1).contiguous()


def state_to_planes(s):
    return s.permute(2, 3, 0, 1).contiguous()


@pytest.fixture(scope='module')
def ops():
    from adaptivepnp_sci_amd import ops as o
    return o


@pytest.mark.parametrize('tag', ['8x8x8', '32x32x8'])
def test_reference_layout_ops_bit_exact(ops, tag):
    g = load_gold('ops_' + tag)
    theta, b, Phi, y, Ps = (dev(g[k]) for k in ('theta', 'b', 'Phi', 'y', 'Phisum'))
    assert torch.equal(ops.phisum(Phi).cpu(), torch.from_numpy(g['Phisum']))
    assert torch.equal(ops.A_(theta, Phi).cpu(), torch.from_numpy(g['A_theta']))
    assert torch.equal(ops.At_(y, Phi).cpu(), torch.from_numpy(g['At_y']))
    x2 = ops.proj_twostage(theta, b, Phi, y, Ps, 1.0, 1.0)
    assert torch.equal(x2.cpu(), torch.from_numpy(g['x_two_stage']))            # bit-exact
    x2r = ops.proj_twostage(theta, b, Phi, y, Ps, 1 / 0.55, 1 * 0.55)
    assert torch.equal(x2r.cpu(), torch.from_numpy(g['x_two_stage_rho055']))
    x1 = ops.proj_onestage(theta, b, Phi, y, Ps, 1.0, 0.01)
    assert torch.equal(x1.cpu(), torch.from_numpy(g['x_one_stage']))
    # in place (x aliases theta), as the reference's first iteration does
    th2 = theta.clone()
    ops.proj_twostage(th2, b, Phi, y, Ps, 1.0, 1.0, out=th2)
    assert torch.equal(th2.cpu(), torch.from_numpy(g['x_two_stage']))


def test_frame_count_limit(ops):
    """the restated torch summation orders (ATen cascade_sum incl. its level-1 flush at 64 strided addends) hold up to 511
    frames -- the contiguous order takes its next level at 64 eight-float vectors: more is refused"""
    Phi = torch.zeros(2, 2, 512, 4, device='cuda')
    with pytest.raises(ValueError):
        ops.phisum(Phi)
    with pytest.raises(ValueError):
        ops.pm_setup(torch.zeros(512, 4, 2, 2, device='cuda'), torch.zeros(4, 2, 2, device='cuda'))


@pytest.mark.parametrize('B', [1, 2, 3, 5, 6, 7, 8, 9, 11, 12, 15, 16, 17, 23, 24, 31, 32, 33, 39, 40, 41, 47, 48, 49, 55, 56, 57, 63,
                               64, 65, 67, 71, 72, 100, 127, 128, 131, 200, 256, 300, 511])
def test_any_frame_count_bit_exact_against_torch_orders(ops, B):
    """Phi_sum (strided torch.sum), A_ (contiguous torch.sum of the fresh product), At_ and both projections, in the
    reference's (M,N,B,4) layout and in the plane-major engine, for every class of frame count: below one 8-float vector,
    whole vectors, vectors + tail, more than four vectors (ATen row_sum's four accumulators), with real-valued masks so
    that the summation order matters.  Oracle = the reference's own torch expressions on the CPU (oracle/sci_ops.py)."""
    from oracle import sci_ops as OO
    rng = np.random.default_rng(100 + B)
    M, N = 6, 10
    T = torch.from_numpy
    theta = rng.random((M, N, B, 4), np.float32)
    b = (0.2 * rng.standard_normal((M, N, B, 4))).astype(np.float32)
    Phi = (rng.random((M, N, B, 4)) * (rng.random((M, N, B, 4)) < 0.7)).astype(np.float32)
    Phi[0, 0] = 0                                        # an unsampled quad: Phi_sum -> 1
    y = (rng.random((M, N, 4)) * B / 2).astype(np.float32)
    Ps = torch.zeros(M, N, 4)
    for ib in range(4):
        Ps[..., ib] = torch.sum(T(Phi)[..., ib], dim=2)     # dvp...:72
    Ps[Ps == 0] = 1
    A_ref = torch.stack([OO.forward_A(T(theta)[..., ib], T(Phi)[..., ib]) for ib in range(4)], -1)
    At_ref = torch.stack([OO.transpose_At(T(y)[..., ib], T(Phi)[..., ib]) for ib in range(4)], -1)
    two = OO.project_two_stage(T(theta), T(b), T(Phi), T(y), Ps, 0.55, 1.0)
    one = OO.project_one_stage(T(theta), T(b), T(Phi), T(y), Ps, 1.0, 0.01)
    # reference layout
    th_d, b_d, Phi_d, y_d = dev(theta), dev(b), dev(Phi), dev(y)
    Ps_d = ops.phisum(Phi_d)
    assert torch.equal(Ps_d.cpu(), Ps)
    assert torch.equal(ops.A_(th_d, Phi_d).cpu(), A_ref)
    assert torch.equal(ops.At_(y_d, Phi_d).cpu(), At_ref)
    assert torch.equal(ops.proj_twostage(th_d, b_d, Phi_d, y_d, Ps_d, np.float32(1 / 0.55), np.float32(0.55)).cpu(), two)
    assert torch.equal(ops.proj_onestage(th_d, b_d, Phi_d, y_d, Ps_d, 1.0, 0.01).cpu(), one)
    # plane-major engine
    th_s, b_s, Phi_s = (planes_to_state(t) for t in (th_d, b_d, Phi_d))
    y_s = y_d.permute(2, 0, 1).contiguous()
    Ps_s, x0 = ops.pm_setup(Phi_s, y_s)
    assert torch.equal(Ps_s.cpu(), Ps.permute(2, 0, 1))
    assert torch.equal(state_to_planes(x0).cpu(), At_ref)
    out = torch.empty_like(th_s)
    ops.pm_project(th_s, b_s, Phi_s, y_s, Ps_s, 0, np.float32(1 / 0.55), np.float32(0.55), out)
    assert torch.equal(state_to_planes(out).cpu(), two)
    ops.pm_project(th_s, b_s, Phi_s, y_s, Ps_s, 1, 1.0, 0.01, out)
    assert torch.equal(state_to_planes(out).cpu(), one)


@pytest.mark.parametrize('tag', ['8x8x8', '32x32x8', '12x20x5'])
def test_plane_major_projection_bit_exact(ops, tag):
    g = load_gold('ops_' + tag)
    theta, b, Phi = (planes_to_state(dev(g[k])) for k in ('theta', 'b', 'Phi'))
    y = dev(g['y']).permute(2, 0, 1).contiguous()
    Ps_ref = torch.from_numpy(g['Phisum']).permute(2, 0, 1).contiguous()
    Ps, x0 = ops.pm_setup(Phi, y)
    assert torch.equal(Ps.cpu(), Ps_ref)
    assert torch.equal(state_to_planes(x0).cpu(), torch.from_numpy(g['At_y']))
    out = torch.empty_like(theta)
    ops.pm_project(theta, b, Phi, y, Ps, 0, 1.0, 1.0, out)
    assert torch.equal(state_to_planes(out).cpu(), torch.from_numpy(g['x_two_stage']))
    ops.pm_project(theta, b, Phi, y, Ps, 0, 1 / 0.55, 0.55, out)
    assert torch.equal(state_to_planes(out).cpu(), torch.from_numpy(g['x_two_stage_rho055']))
    ops.pm_project(theta, b, Phi, y, Ps, 1, 1.0, 0.01, out)
    assert torch.equal(state_to_planes(out).cpu(), torch.from_numpy(g['x_one_stage']))
    th2 = theta.clone()
    ops.pm_project(th2, b, Phi, y, Ps, 1, 1.0, 0.01, th2)      # in place
    assert torch.equal(state_to_planes(th2).cpu(), torch.from_numpy(g['x_one_stage']))


def test_layout_conversions_bit_exact(ops):
    g = load_gold('bayer_12x20x5')
    mos = dev(g['mosaic'])
    rng = np.random.default_rng(0)
    # reference layout split/merge on an 8-frame cube
    mos8 = dev(rng.uniform(size=(12, 20, 8)).astype(np.float32))
    from oracle import sci_ops as OO
    pl = ops.bayer_split(mos8)
    assert torch.equal(pl.cpu(), OO.bayer_split(mos8.cpu()))
    assert torch.equal(ops.bayer_merge(pl).cpu(), mos8.cpu())
    # plane-major (any B)
    st = ops.mosaic_to_state(mos)
    assert torch.equal(state_to_planes(st).cpu(), torch.from_numpy(g['planes']))
    assert torch.equal(ops.state_to_mosaic(st).cpu(), mos.cpu())
    yy = dev(rng.uniform(size=(12, 20)).astype(np.float32))
    assert torch.equal(ops.y_to_meas(yy).cpu(), OO.bayer_split(yy.cpu()).permute(2, 0, 1))
    cube = dev(rng.uniform(size=(12, 20, 3, 5)).astype(np.float32))
    rgb = ops.cube_to_rgb(cube)
    assert torch.equal(rgb.cpu(), cube.cpu().permute(3, 2, 0, 1))
    assert torch.equal(ops.rgb_to_cube(rgb).cpu(), cube.cpu())
    # more than 64 frames: the conversions walk the frames in chunks of 64
    for B in (64, 65, 130, 200):
        m = dev(rng.uniform(size=(6, 70, B)).astype(np.float32))
        st = ops.mosaic_to_state(m)
        assert torch.equal(state_to_planes(st).cpu(), OO.bayer_split(m.cpu()))
        assert torch.equal(ops.state_to_mosaic(st), m)
        c = dev(rng.uniform(size=(3, 70, 3, B)).astype(np.float32))
        r = ops.cube_to_rgb(c)
        assert torch.equal(r.cpu(), c.cpu().permute(3, 2, 0, 1)) and torch.equal(ops.rgb_to_cube(r), c)


@pytest.mark.parametrize('kernel', [1, 2, 3], ids=['tiled', 'whole-plane', 'banded'])
def test_tv_chambolle_matches_skimage_golden(ops, kernel):
    g = load_gold('tv_chambolle')
    v = dev(g['v']).permute(2, 0, 1).contiguous()     # (C, M, N)
    C_, M, N = v.shape
    for key, w, n in (('w01_n5', 0.1, 5), ('w01_n50', 0.1, 50), ('w003_n5', 0.03, 5)):
        if kernel == 3 and n > 5:
            continue                                  # the banded kernel's halo covers five iterations
        plan = ops.TvPlan(M, N, C_, n, v.device)
        out = torch.empty_like(v)
        ops.tv_chambolle(v, None, 0.0, out, plan, w, kernel=kernel)
        ref = torch.from_numpy(g['out_' + key]).permute(2, 0, 1)
        stop = plan.stop_iter.cpu().numpy()
        assert (stop == g['stop_' + key]).all(), (key, stop, g['stop_' + key])   # incl. early-stopping channels
        # same fp32 operations per pixel as skimage; only the energy sums differ (they do not enter `out`)
        assert rel_l2(out.cpu().numpy(), ref.numpy()) == 0.0, key


@pytest.mark.parametrize('kernel', [1, 2, 3], ids=['tiled', 'whole-plane', 'banded'])
def test_tv_fused_input_and_ragged_size(ops, kernel):
    from oracle.tv_chambolle import tv_chambolle_multichannel
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 2, (37, 45, 6)).astype(np.float32)
    b = rng.normal(0, 0.3, (37, 45, 6)).astype(np.float32)
    coef = np.float32(1 / 0.55)
    ref, stops, _ = tv_chambolle_multichannel(x + coef * b, 0.1, n_iter_max=5, return_info=True)
    xs, bs = dev(x).permute(2, 0, 1).contiguous(), dev(b).permute(2, 0, 1).contiguous()
    plan = ops.TvPlan(37, 45, 6, 5, xs.device)
    out = torch.empty_like(xs)
    ops.tv_chambolle(xs, bs, float(coef), out, plan, 0.1, kernel=kernel)
    assert rel_l2(out.permute(1, 2, 0).cpu().numpy(), ref) == 0.0
    assert (plan.stop_iter.cpu().numpy() == stops).all()


@pytest.mark.parametrize('shape', [(128, 128, 32), (256, 256, 8), (300, 256, 3), (127, 128, 3), (128, 65, 3), (64, 63, 40), (9, 130, 2),
                                   (1, 1, 2), (3, 200, 2), (7, 5, 3), (100, 64, 5), (64, 100, 2), (65, 64, 2), (33, 17, 2),
                                   (16, 255, 2), (17, 129, 3), (48, 64, 70)])
@pytest.mark.parametrize('n_iter', [1, 2, 3, 5])
def test_tv_banded_kernel_equals_the_tiled_kernel(ops, shape, n_iter):
    """planes up to 256 columns run all (<= 5) iterations in one launch of MANY workgroups per channel (csrc/tv.hip
    tv_band_kernel: bands of 16 / 32 rows computed with a 4-row halo, stop test and recomputation of early-stopped channels
    in a second launch): same float32 operations per pixel, so `out` and the stop iterations must be identical to the
    per-iteration tiled kernel at every size, band / strip / wave seam and iteration count"""
    M, N, C_ = shape
    rng = np.random.default_rng(M * 1000 + N)
    x = dev(rng.uniform(0, 1, (C_, M, N)).astype(np.float32))
    x[0] *= 1000.0                                    # large-amplitude channels: the dual field saturates and the energy
    x[-1] *= 50.0                                     # settles within 2e-4 before the last iteration (the recomputation path)
    b = dev(rng.normal(0, 0.1, (C_, M, N)).astype(np.float32))
    plan1, plan2 = ops.TvPlan(M, N, C_, n_iter, x.device), ops.TvPlan(M, N, C_, n_iter, x.device)
    o1, o2 = torch.empty_like(x), torch.full_like(x, -7.0)
    ops.tv_chambolle(x, b, -1.0, o1, plan1, 0.1, kernel=1)
    ops.tv_chambolle(x, b, -1.0, o2, plan2, 0.1, kernel=3)
    assert torch.equal(plan1.stop_iter, plan2.stop_iter), (plan1.stop_iter, plan2.stop_iter)
    assert torch.equal(o1, o2)
    o0 = torch.full_like(x, -3.0)
    ops.tv_chambolle(x, b, -1.0, o0, plan2, 0.1, kernel=0)      # the library's own choice
    assert torch.equal(o0, o1)
    # kernel 4 = the CANDIDATE form (one launch stores every iteration's `out` and the bands' partial sums -- no counters, no
    # atomics, no communication between bands -- nothing is recomputed; here followed by the stop-test / selection launches); a
    # call leaves nothing behind in the workspace that the next one reads, so a plan is reusable at once -- the same plan runs
    # both banded forms, twice
    if n_iter >= 2:
        for kernel in (4, 3, 4, 4):
            o4 = torch.full_like(x, -9.0)
            plan2.stop_iter.fill_(-1)
            ops.tv_chambolle(x, b, -1.0, o4, plan2, 0.1, kernel=kernel)
            assert torch.equal(o4, o1) and torch.equal(plan1.stop_iter, plan2.stop_iter), kernel


def test_tv_banded_kernel_early_stop_is_exercised(ops):
    rng = np.random.default_rng(0)
    x = dev(rng.uniform(0, 1, (6, 96, 128)).astype(np.float32))
    x[1] *= 50.0                                      # large amplitudes saturate the dual field: early stop
    x[4] *= 1000.0
    plan1, plan2 = ops.TvPlan(96, 128, 6, 5, x.device), ops.TvPlan(96, 128, 6, 5, x.device)
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    ops.tv_chambolle(x, None, 0.0, o1, plan1, 0.1, kernel=1)
    ops.tv_chambolle(x, None, 0.0, o2, plan2, 0.1, kernel=3)
    assert torch.equal(plan1.stop_iter, plan2.stop_iter) and torch.equal(o1, o2)
    assert int(plan1.stop_iter.min()) < 4 and int(plan1.stop_iter.max()) == 4


@pytest.mark.parametrize('shape', [(128, 128, 8), (127, 128, 3), (128, 65, 3), (64, 63, 4), (9, 130, 2), (1, 1, 2), (3, 200, 2),
                                   (7, 5, 3), (100, 64, 5), (64, 100, 2), (65, 64, 2), (33, 17, 2)])
@pytest.mark.parametrize('n_iter', [1, 2, 5, 40])
def test_tv_whole_plane_kernel_equals_the_tiled_kernel(ops, shape, n_iter):
    """planes up to 128 x 128 run all iterations in one launch (csrc/tv.hip tv_plane_kernel): same float32 operations per
    pixel, so `out` and the stop iterations must be identical to the per-iteration tiled kernel at every size, seam and
    iteration count; wider planes are refused by kernel=2 and fall to the tiled kernel under kernel=0"""
    M, N, C_ = shape
    rng = np.random.default_rng(M * 1000 + N)
    x = dev(rng.uniform(0, 1, (C_, M, N)).astype(np.float32))
    x[0] = x[0] * 0.02 + 0.5                          # a nearly flat channel: stops early
    b = dev(rng.normal(0, 0.1, (C_, M, N)).astype(np.float32))
    plan1, plan2 = ops.TvPlan(M, N, C_, n_iter, x.device), ops.TvPlan(M, N, C_, n_iter, x.device)
    o1, o2 = torch.empty_like(x), torch.full_like(x, -7.0)
    ops.tv_chambolle(x, b, -1.0, o1, plan1, 0.1, kernel=1)
    if N > 128 or M > 128:
        with pytest.raises(ValueError):              # SCIPNP_EINVAL
            ops.tv_chambolle(x, b, -1.0, o2, plan2, 0.1, kernel=2)
        ops.tv_chambolle(x, b, -1.0, o2, plan2, 0.1, kernel=1 if n_iter > 5 and N > 256 else 0)
    else:
        ops.tv_chambolle(x, b, -1.0, o2, plan2, 0.1, kernel=2)
    assert torch.equal(o1, o2)
    assert torch.equal(plan1.stop_iter, plan2.stop_iter)
    if n_iter == 40 and M * N > 1:
        assert int(plan1.stop_iter.min()) < 39          # the early stop was exercised


@pytest.mark.parametrize('tag', ['16x16', '64x64', '8x24'])
def test_malvar_pre_denoise(ops, tag):
    g = load_gold('malvar')
    cfa = g['cfa_' + tag]
    H, W = cfa.shape
    B = 3
    rng = np.random.default_rng(1)
    mos = np.stack([cfa, rng.uniform(0, 1, (H, W)).astype(np.float32), cfa[::-1].copy()], -1)
    from oracle.malvar import malvar_demosaic_cube
    ref = malvar_demosaic_cube(torch.from_numpy(mos))            # (H,W,3,B)
    assert rel_l2(ref[..., 0].numpy(), g['rgb_' + tag]) == 0.0   # oracle == reference golden
    x = ops.mosaic_to_state(dev(mos))
    x_rgb = torch.empty(B, 3, H, W, device='cuda')
    ops.pm_pre_denoise(x, None, None, x_rgb, None, None, 1.0, 0.0, 0.0)
    got = ops.rgb_to_cube(x_rgb).cpu().numpy()
    # CFA sites are copied exactly; interpolated sites: 5x5 fp32 correlation, summation order only
    assert rel_l2(got, ref.numpy()) < 2e-7
    assert np.abs(got - ref.numpy()).max() < 1e-6


def test_pre_post_denoise_fusion(ops):
    from oracle import sci_ops as OO
    from oracle.malvar import malvar_demosaic_cube
    rng = np.random.default_rng(2)
    M, N, B = 12, 40, 4
    H, W = 2 * M, 2 * N
    xp = torch.from_numpy(rng.uniform(0, 1, (M, N, B, 4)).astype(np.float32))
    bp = torch.from_numpy(rng.normal(0, 0.1, (M, N, B, 4)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.5, (H, W, 3, B)).astype(np.float32))
    inv_rho, inv_tau, sigma = np.float32(1 / 0.55), np.float32(1 / 100), 25 / 255
    mosaic = OO.bayer_merge(xp + float(inv_rho) * bp)
    x_rgb_ref = malvar_demosaic_cube(mosaic)
    in_ref = x_rgb_ref - float(inv_tau) * w
    x, b = planes_to_state(xp.cuda()), planes_to_state(bp.cuda())
    w_d = ops.cube_to_rgb(w.cuda())
    x_rgb = torch.empty(B, 3, H, W, device='cuda')
    rgb_w = torch.empty_like(x_rgb)
    c8 = torch.empty(B, 2, M, N, 8, device='cuda')
    ops.pm_pre_denoise(x, b, w_d, x_rgb, rgb_w, c8, inv_rho, inv_tau, sigma)
    assert rel_l2(ops.rgb_to_cube(x_rgb).cpu().numpy(), x_rgb_ref.numpy()) < 2e-7
    assert rel_l2(ops.rgb_to_cube(rgb_w).cpu().numpy(), in_ref.numpy()) < 2e-7
    # c8 = pixel-unshuffle(rgb_w) ++ sigma map ++ zeros  (reference network_ffdnet.py:61-64)
    nchw = ops.from_c8(c8)                                   # (B,16,M,N)
    un = rgb_w.reshape(B, 3, M, 2, N, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, 12, M, N)
    assert torch.equal(nchw[:, :12], un)
    assert torch.equal(nchw[:, 12], torch.full((B, M, N), np.float32(sigma), device='cuda'))
    assert (nchw[:, 13:] == 0).all()

    # ---- post: theta gather, clip, duals, SSE; both the planar and the c8 (pre pixel-shuffle) sources
    out_cube = torch.from_numpy(rng.uniform(-0.2, 1.2, (H, W, 3, B)).astype(np.float32))
    orig = torch.from_numpy(rng.uniform(0, 1, (M, N, B, 4)).astype(np.float32))
    raw = OO.rgb_to_bayer_planes(out_cube)
    th_ref = torch.clip(raw, 0, 1)
    for alias in (False, True):
        b_ref = bp + ((raw if alias else xp) - th_ref)
        w_ref = w + (x_rgb_ref - out_cube)
        for src in ('rgb', 'c8'):
            theta = torch.empty_like(x)
            b2, w2, x2 = b.clone(), w_d.clone(), x.clone()
            xr = ops.cube_to_rgb(x_rgb_ref.cuda())
            out_rgb = ops.cube_to_rgb(out_cube.cuda())
            part = torch.empty(ops.post_nblocks(M, N, B), dtype=torch.float64, device='cuda')
            store = torch.empty_like(out_rgb)
            if src == 'rgb':
                ops.pm_post_denoise(out_rgb, None, store, x2, xr, theta, b2, w2, alias, planes_to_state(orig.cuda()), part)
            else:
                un = out_rgb.reshape(B, 3, M, 2, N, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, 12, M, N)
                oc8 = ops.to_c8(un)
                ops.pm_post_denoise(None, oc8, store, x2, xr, theta, b2, w2, alias, planes_to_state(orig.cuda()), part)
            assert torch.equal(state_to_planes(theta).cpu(), th_ref)             # bit-exact
            assert torch.equal(state_to_planes(b2).cpu(), b_ref)
            assert torch.equal(ops.rgb_to_cube(w2).cpu(), w_ref)
            assert torch.equal(store, out_rgb)
            if alias:
                assert torch.equal(state_to_planes(x2).cpu(), raw)
            sse_ref = float(((orig - th_ref).double() ** 2).sum())
            assert abs(float(part.sum()) - sse_ref) <= 1e-9 * sse_ref


def test_dual_update_and_sse(ops):
    rng = np.random.default_rng(3)
    B, M, N = 8, 16, 24
    raw = torch.from_numpy(rng.uniform(-0.3, 1.3, (B, 4, M, N)).astype(np.float32))
    x = torch.from_numpy(rng.uniform(0, 1, (B, 4, M, N)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 0.1, (B, 4, M, N)).astype(np.float32))
    orig = torch.from_numpy(rng.uniform(0, 1, (B, 4, M, N)).astype(np.float32))
    th_ref = torch.clip(raw, 0, 1)
    for sign, which in ((+1.0, 0), (-1.0, 1)):
        theta = torch.empty_like(raw).cuda()
        b2 = b.clone().cuda()
        part = torch.empty(ops.sse_nblocks(raw.numel()), dtype=torch.float64, device='cuda')
        ops.pm_dual_update(raw.cuda(), x.cuda(), theta, b2, sign, orig.cuda(), part, which)
        assert torch.equal(theta.cpu(), th_ref)
        assert torch.equal(b2.cpu(), b + (x - th_ref) if sign > 0 else b - (x - th_ref))
        rep = th_ref if which == 0 else x
        ref = float((((orig - rep) ** 2).double()).sum())
        assert abs(float(part.sum()) - ref) <= 1e-12 * ref
    assert abs(ops.sse(orig.cuda(), x.cuda()) - float((((orig - x) ** 2).double()).sum())) < 1e-9


@pytest.mark.parametrize('cin,cout,h,w,n', [(16, 96, 16, 32, 2), (96, 96, 24, 40, 1), (96, 16, 9, 33, 2),
                                            (32, 64, 16, 32, 1), (64, 128, 8, 32, 1), (128, 256, 8, 32, 1),
                                            (8, 32, 5, 7, 3)])
def test_conv3x3_mfma_vs_fp64(ops, cin, cout, h, w, n):
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    packed = ops.pack_conv3x3(wt, bias, Cin=cin, Cout=cout, device='cuda')
    xc = ops.to_c8(x.cuda())
    got = ops.from_c8(ops.conv3x3_c8(xc, packed, cout)).cpu()
    # exact fp32 products, fp32 fmaf-chain accumulation over K = 9*cin <= 1152 terms vs an fp64 reference:
    # relative L2 error grows like sqrt(K)*2^-24 (measured 1.5e-7 .. 4.7e-7); bf16/fp16 operands would give 1e-3
    assert rel_l2(got.numpy(), ref.numpy()) < 1e-6
    got = ops.from_c8(ops.conv3x3_c8(xc, packed, cout, relu=True, residual=ops.to_c8(res.cuda()))).cpu()
    assert rel_l2(got.numpy(), torch.relu(ref + res.double()).numpy()) < 1e-6


def test_conv3x3_identity_asymmetric(ops):
    """A = I check with an asymmetric operand: catches row/col swaps of the MFMA output map."""
    cin = cout = 32
    wt = torch.zeros(cout, cin, 3, 3)
    for c in range(cout):
        wt[c, (c * 7 + 3) % cin, 1, 1] = 1.0          # a channel permutation, centre tap only
    x = torch.arange(1 * cin * 8 * 32, dtype=torch.float32).reshape(1, cin, 8, 32) * 0.001
    packed = ops.pack_conv3x3(wt, None, Cin=cin, Cout=cout, device='cuda')
    got = ops.from_c8(ops.conv3x3_c8(ops.to_c8(x.cuda()), packed, cout)).cpu()
    perm = [(c * 7 + 3) % cin for c in range(cout)]
    assert torch.equal(got, x[:, perm])


@pytest.mark.parametrize('tag,sigmas', [('64x64', (6, 12, 25, 50)), ('128x128', (25,)), ('37x50', (12,))])
def test_ffdnet_forward_vs_reference_golden(ffdnet_state_dict, tag, sigmas):
    from adaptivepnp_sci_amd.nets import FFDNet
    g = load_gold('ffdnet_forward')
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    x = dev(g['in_' + tag])
    for s in sigmas:
        out = net(x, torch.full((1, 1, 1, 1), s / 255.)).cpu().numpy()
        # 12 layers of fp32 MFMA vs oneDNN fp32 on the CPU: summation order only
        assert rel_l2(out, g[f'out_{tag}_s{s}']) < 2e-6, (tag, s)


def test_conv3x3_stride2_and_pixelshuffle(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 12, 36, generator=g)
    wt = torch.randn(64, 32, 3, 3, generator=g) * 0.06
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, stride=2, padding=1)
    packed = ops.pack_conv3x3(wt, None, Cin=32, Cout=64, device='cuda')
    got = ops.from_c8(ops.conv3x3_c8(ops.to_c8(x.cuda()), packed, 64, stride2=True)).cpu()
    assert got.shape == ref.shape and rel_l2(got.numpy(), ref.numpy()) < 1e-6
    # odd input size: out = (h-1)//2+1
    x = torch.randn(1, 8, 9, 35, generator=g)
    wt = torch.randn(32, 8, 3, 3, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, stride=2, padding=1)
    packed = ops.pack_conv3x3(wt, None, Cin=8, Cout=32, device='cuda')
    got = ops.from_c8(ops.conv3x3_c8(ops.to_c8(x.cuda()), packed, 32, stride2=True)).cpu()
    assert got.shape == ref.shape and rel_l2(got.numpy(), ref.numpy()) < 1e-6
    # PixelShuffle(2) epilogue with a residual in the shuffled layout and folded BatchNorm-style scale/shift
    x = torch.randn(2, 64, 10, 33, generator=g)
    wt = torch.randn(128, 64, 3, 3, generator=g) * 0.04
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    res = torch.randn(2, 32, 20, 66, generator=g)
    conv = torch.nn.functional.conv2d(x.double(), wt.double(), None, padding=1) * sc.double()[None, :, None, None] \
        + sh.double()[None, :, None, None]
    ref = torch.nn.functional.pixel_shuffle(conv, 2) + res.double()
    packed = ops.pack_conv3x3(wt, None, sc, sh, Cin=64, Cout=128, device='cuda')
    got = ops.from_c8(ops.conv3x3_c8(ops.to_c8(x.cuda()), packed, 128, shuffle=True, residual=ops.to_c8(res.cuda()))).cpu()
    assert got.shape == ref.shape and rel_l2(got.numpy(), ref.numpy()) < 1e-6


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_fastdvdnet_forward_vs_reference_golden(precision, monkeypatch):
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', precision)
    """Synthetic seeded weights (the reference's model.pth is not in the snapshot); circular-window edge frames
    0,1,6,7 included (B = 8); golden produced by the reference's fastdvdnet_denoiser_full_tensor_v2."""
    from adaptivepnp_sci_amd import fastdvdnet_denoiser_full_tensor_v2
    from oracle.nets import cpu_data_parallel, synth_fastdvdnet_weights
    g = load_gold('fastdvd_forward')
    net = cpu_data_parallel(synth_fastdvdnet_weights(0))          # the reference wraps it like this
    out = fastdvdnet_denoiser_full_tensor_v2(dev(g['v']), float(g['sigma']), None, None, net, True, 1e-6)
    # 32 fp32-MFMA conv layers + folded BatchNorm vs PyTorch-CPU: summation order / fold rounding only
    assert rel_l2(out.cpu().numpy(), g['out']) < 5e-6


def test_wgrad_bgrad_backward_data_vs_autograd(ops):
    """conv weight / bias gradients (MFMA GEMM over pixels) and the backward-data conv (transposed, flipped
    weights + ReLU mask) against PyTorch autograd in float64."""
    import ctypes as C
    from adaptivepnp_sci_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    n, cin, cout, h, w = 2, 96, 96, 10, 37
    x = torch.relu(torch.randn(n, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    bias = torch.randn(cout, generator=g)
    dz = torch.randn(n, cout, h, w, generator=g)
    xd, wd, bd = x.double().requires_grad_(), wt.double().requires_grad_(), bias.double().requires_grad_()
    out = torch.nn.functional.conv2d(xd, wd, bd, padding=1)
    out.backward(dz.double())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    x8, dz8 = ops.to_c8(x.cuda()), ops.to_c8(dz.cuda())
    nslab = 16
    ws = torch.empty(lib.scipnp_conv3x3_wgrad_workspace_floats(cin, cout, nslab), device='cuda')
    dW = torch.empty(cout, cin, 3, 3, device='cuda')
    _lib.check(lib.scipnp_conv3x3_wgrad(p(x8), p(dz8), p(dW), p(ws), nslab, n, cin, cout, cin, cout, h, w, s), 'wgrad')
    assert rel_l2(dW.cpu().numpy(), wd.grad.numpy()) < 1e-6
    db = torch.empty(cout, device='cuda')
    bws = torch.empty((cout // 8) * 64 * 8, device='cuda')
    _lib.check(lib.scipnp_conv_bias_grad(p(dz8), p(db), p(bws), n, cout, cout, h, w, s), 'bgrad')
    assert rel_l2(db.cpu().numpy(), bd.grad.numpy()) < 1e-6
    # head-like shapes: 13 real input channels in 16, and 12 real output channels in 16
    for ci_r, co_r, ci, co in ((13, 96, 16, 96), (96, 12, 96, 16)):
        x = torch.randn(n, ci_r, h, w, generator=g)
        wt2 = (torch.randn(co_r, ci_r, 3, 3, generator=g) * 0.05).double().requires_grad_()
        dz = torch.randn(n, co_r, h, w, generator=g)
        torch.nn.functional.conv2d(x.double(), wt2, None, padding=1).backward(dz.double())
        ws = torch.empty(lib.scipnp_conv3x3_wgrad_workspace_floats(ci, co, nslab), device='cuda')
        dW = torch.empty(co_r, ci_r, 3, 3, device='cuda')
        x8b, dz8b = ops.to_c8(x.cuda()), ops.to_c8(dz.cuda())      # keep alive: p() only takes the address
        _lib.check(lib.scipnp_conv3x3_wgrad(p(x8b), p(dz8b), p(dW), p(ws), nslab, n, ci_r, co_r, ci, co, h, w, s), 'wgrad')
        assert rel_l2(dW.cpu().numpy(), wt2.grad.numpy()) < 1e-6
    # backward-data with device-side transposed/flipped packing and the ReLU mask
    dz = torch.randn(n, cout, h, w, generator=g)
    packed = torch.empty(lib.scipnp_conv3x3_packed_floats(cout, cin), device='cuda')
    wdev = wt.cuda()
    _lib.check(lib.scipnp_pack_conv3x3_device(p(wdev), None, p(packed), cin, cout, cin, cout, 1, s), 'pack bwd')
    xin = torch.randn(n, cin, h, w, generator=g)
    act = torch.relu(xin)
    xin_d = xin.double().requires_grad_()
    torch.nn.functional.conv2d(torch.relu(xin_d), wt.double(), None, padding=1).backward(dz.double())
    got = torch.empty(n, cin // 8, h, w, 8, device='cuda')
    dz8c, act8 = ops.to_c8(dz.cuda()), ops.to_c8(act.cuda())
    _lib.check(lib.scipnp_conv3x3_c8(p(dz8c), p(packed), p(got), p(act8), n, cout, cin, h, w, 16, s), 'bwd-data')
    assert rel_l2(ops.from_c8(got).cpu().numpy(), xin_d.grad.numpy()) < 1e-6
    # forward device packing == host packing
    pk_host = ops.pack_conv3x3(wt, bias, Cin=cin, Cout=cout, device='cuda')
    pk_dev = torch.empty_like(pk_host)
    _lib.check(lib.scipnp_pack_conv3x3_device(p(wdev), p(bias.cuda()), p(pk_dev), cin, cout, cin, cout, 0, s), 'pack fwd')
    assert torch.equal(pk_host, pk_dev)


def test_adam_step_matches_torch(ops):
    import ctypes as C
    from adaptivepnp_sci_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(5000, generator=g)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=2e-6)
    pd, m, v = p0.clone().cuda(), torch.zeros(5000, device='cuda'), torch.zeros(5000, device='cuda')
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for step in (1, 2, 3):
        grad = torch.randn(5000, generator=g) * 10 ** float(torch.randint(-4, 1, (1,), generator=g))
        ref.grad = grad.clone()
        opt.step()
        gd = grad.cuda()
        _lib.check(lib.scipnp_adam_step(C.c_void_p(pd.data_ptr()), C.c_void_p(gd.data_ptr()), C.c_void_p(m.data_ptr()),
                                        C.c_void_p(v.data_ptr()), 5000, 2e-6, 0.9, 0.999, 1e-8, step, s), 'adam')
        upd_ref, upd = (ref.detach() - p0).numpy(), (pd.cpu() - p0).numpy()
        assert rel_l2(upd, upd_ref) < 1e-3, step          # the update itself (~lr), float32 round-off of p dominates
        assert rel_l2(pd.cpu().numpy(), ref.detach().numpy()) < 1e-7


def test_fastdvd_backward_glue_vs_autograd(ops):
    """stride-2 conv gradients through zero-insertion upsampling, PixelShuffle backward, folded-BN parameter
    gradients: against float64 autograd."""
    import ctypes as C
    from adaptivepnp_sci_amd import _lib
    lib = _lib.load()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
    g = torch.Generator().manual_seed(21)
    n, cin, cout, h, w = 2, 32, 64, 12, 40
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    mean, var = torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5
    dy = torch.randn(n, cout, h // 2, w // 2, generator=g)
    xd, wd, gd, bd = (t.double().requires_grad_() for t in (x, wt, gamma, beta))
    u = torch.nn.functional.conv2d(xd, wd, None, stride=2, padding=1)
    y = torch.nn.functional.batch_norm(u, mean.double(), var.double(), gd, bd, training=False, eps=1e-5)
    y.backward(dy.double())
    # device side
    dy8 = ops.to_c8(dy.cuda())
    up = torch.empty(n, cout // 8, h, w, 8, device='cuda')
    _lib.check(lib.scipnp_upsample_zero_c8(p(dy8), p(up), n, cout, h // 2, w // 2, h, w, s), 'up')
    x8 = ops.to_c8(x.cuda())
    nslab = 8
    ws = torch.empty(lib.scipnp_conv3x3_wgrad_workspace_floats(cin, cout, nslab), device='cuda')
    G = torch.empty(cout, cin, 3, 3, device='cuda')
    _lib.check(lib.scipnp_conv3x3_wgrad(p(x8), p(up), p(G), p(ws), nslab, n, cin, cout, cin, cout, h, w, s), 'wgrad')
    sdy = torch.empty(cout, device='cuda')
    bws = torch.empty((cout // 8) * 64 * 8, device='cuda')
    _lib.check(lib.scipnp_conv_bias_grad(p(dy8), p(sdy), p(bws), n, cout, cout, h // 2, w // 2, s), 'bgrad')
    dW, dga, dbe = torch.empty_like(G), torch.empty(cout, device='cuda'), torch.empty(cout, device='cuda')
    W_d, ga_d, mu_d, var_d = wt.cuda(), gamma.cuda(), mean.cuda(), var.cuda()
    _lib.check(lib.scipnp_bn_fold_grads(p(W_d), p(G), p(sdy), p(ga_d), p(mu_d), p(var_d), 1e-5, p(dW), p(dga), p(dbe), cout,
                                        cin * 9, s), 'bn grads')
    assert rel_l2(dW.cpu().numpy(), wd.grad.numpy()) < 2e-6
    assert rel_l2(dga.cpu().numpy(), gd.grad.numpy()) < 2e-5
    assert rel_l2(dbe.cpu().numpy(), bd.grad.numpy()) < 2e-6
    # backward-data of the stride-2 + BN layer = stride-1 transposed conv of the upsampled gradient, scale folded
    sc, sh = torch.empty(cout, device='cuda'), torch.empty(cout, device='cuda')
    be_d = beta.cuda()
    _lib.check(lib.scipnp_bn_fold(p(ga_d), p(be_d), p(mu_d), p(var_d), 1e-5, p(sc), p(sh), cout, s), 'fold')
    pk = torch.empty(lib.scipnp_conv3x3_packed_floats(cout, cin), device='cuda')
    _lib.check(lib.scipnp_pack_conv3x3_device_scaled(p(W_d), None, p(sc), p(pk), cin, cout, cin, cout, 1, s), 'pack')
    dx = torch.empty(n, cin // 8, h, w, 8, device='cuda')
    _lib.check(lib.scipnp_conv3x3_c8_ex(p(up), p(pk), p(dx), None, None, n, cout, cin, h, w, 0, s), 'bwd')
    assert rel_l2(ops.from_c8(dx).cpu().numpy(), xd.grad.numpy()) < 2e-6
    # forward with the folded pack == conv + BN
    pkf = torch.empty(lib.scipnp_conv3x3_packed_floats(cin, cout), device='cuda')
    _lib.check(lib.scipnp_pack_conv3x3_device_scaled(p(W_d), p(sh), p(sc), p(pkf), cin, cout, cin, cout, 0, s), 'pack')
    yf = ops.from_c8(ops.conv3x3_c8(x8, pkf, cout, stride2=True)).cpu()
    assert rel_l2(yf.numpy(), y.detach().numpy()) < 2e-6
    # PixelShuffle backward
    ds = torch.randn(n, 16, 2 * h, 2 * w, generator=g)
    cd = torch.randn(n, 64, h, w, generator=g).double().requires_grad_()
    torch.nn.functional.pixel_shuffle(cd, 2).backward(ds.double())
    ds8 = ops.to_c8(ds.cuda())
    dc = torch.empty(n, 8, h, w, 8, device='cuda')
    _lib.check(lib.scipnp_pixel_shuffle_bwd_c8(p(ds8), p(dc), n, 16, h, w, s), 'unshuffle')
    assert torch.equal(ops.from_c8(dc).cpu(), cd.grad.float())


@pytest.mark.parametrize('cin,cout,h,w,n', [(16, 96, 16, 32, 2), (96, 96, 24, 40, 1), (96, 16, 9, 33, 2), (32, 64, 8, 32, 1)])
def test_conv3x3_split_fp16_vs_fp64(ops, cin, cout, h, w, n):
    """error-compensated split-fp16 MFMA conv: 22-bit operands, exact products, fp32 accumulation."""
    g = torch.Generator().manual_seed(cin * 77 + cout)
    x = torch.randn(n, cin, h, w, generator=g) * 3.0
    x[0, 0, 0, :8] = torch.tensor([1e-6, -3e-5, 2e-4, 0.0, 1e-3, -0.04, 37.5, -120.0])   # tiny and large magnitudes
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    wt[0, 0, 1, 1] = 4.4886
    bias = torch.randn(cout, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    packed = ops.pack_conv3x3_split(wt, bias, Cin=cin, Cout=cout, device='cuda')
    xs = ops.c8_to_c8s(ops.to_c8(x.cuda()))
    assert rel_l2(ops.c8s_to_float(xs).cpu().numpy(), ops.to_c8(x).numpy()) < 3e-7          # 2^-22 operand format
    got = ops.from_c8(ops.conv3x3_c8s(xs, packed, cout, f32_out=True)).cpu()
    assert rel_l2(got.numpy(), ref.numpy()) < 2e-6, rel_l2(got.numpy(), ref.numpy())
    got_s = ops.conv3x3_c8s(xs, packed, cout, relu=True)                                     # split output + ReLU
    got2 = ops.from_c8(ops.c8s_to_float(got_s)).cpu()
    assert rel_l2(got2.numpy(), torch.relu(ref).numpy()) < 2e-6


def test_split_pack_rejects_huge_weights(ops):
    with pytest.raises(ValueError):
        ops.pack_conv3x3_split(torch.full((8, 8, 3, 3), 40.0), None, Cin=8, Cout=8)


def test_ffdnet_forward_split_precision(ffdnet_state_dict, monkeypatch):
    from adaptivepnp_sci_amd.nets import FFDNet
    monkeypatch.setenv('SCIPNP_FFDNET_PRECISION', 'f16x3')
    g = load_gold('ffdnet_forward')
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    x = dev(g['in_128x128'])
    out = net(x, torch.full((1, 1, 1, 1), 25 / 255.)).cpu().numpy()
    assert rel_l2(out, g['out_128x128_s25']) < 5e-6


def test_split_overflow_guard(ops):
    x = torch.zeros(1, 8, 8, 32)
    wt = torch.zeros(8, 8, 3, 3)
    wt[:, :, 1, 1] = 30.0
    packed = ops.pack_conv3x3_split(wt, None, Cin=8, Cout=8, device='cuda')
    ops.split_overflow()                                           # clear
    ops.conv3x3_c8s(ops.c8_to_c8s(ops.to_c8(x.cuda() + 1.0)), packed, 8)
    assert not ops.split_overflow()
    ops.conv3x3_c8s(ops.c8_to_c8s(ops.to_c8(x.cuda() + 400.0)), packed, 8)    # 8*30*400 = 96000 > fp16 max
    assert ops.split_overflow() and not ops.split_overflow()       # reported once, then reset


def test_split_overflow_words_are_per_solve(ops):
    """the range guard raises the word the CALLING THREAD bound (scipnp_bind_overflow_word), not one process-wide flag:
    two solves that overlap in time -- here two host threads on two streams -- never see each other's report"""
    import threading
    wt = torch.zeros(8, 8, 3, 3)
    wt[:, :, 1, 1] = 30.0
    packed = ops.pack_conv3x3_split(wt, None, Cin=8, Cout=8, device='cuda')
    ops.split_overflow()                                           # clear the process-wide word
    words = [torch.zeros(1, dtype=torch.int32, device='cuda') for _ in range(2)]
    torch.cuda.synchronize()
    seen, errs = [None, None], []
    barrier = threading.Barrier(2)

    def solve(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st), ops.overflow_scope(words[i]):
                barrier.wait()                                     # both scopes are open at the same time
                x = torch.zeros(1, 8, 8, 32, device='cuda') + (400.0 if i == 1 else 1.0)    # thread 1 overflows
                for _ in range(4):
                    ops.conv3x3_c8s(ops.c8_to_c8s(ops.to_c8(x)), packed, 8)
                barrier.wait()
                seen[i] = ops.split_overflow(word=words[i])
        except Exception as e:                                     # noqa: BLE001
            errs.append(e)
            barrier.abort()

    ts = [threading.Thread(target=solve, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert seen == [False, True]
    assert not ops.split_overflow()                                # the process-wide word was never touched
    # scopes nest and restore: after the inner scope the outer word is bound again
    with ops.overflow_scope(words[0]):
        with ops.overflow_scope(words[1]):
            pass
        ops.conv3x3_c8s(ops.c8_to_c8s(ops.to_c8(torch.zeros(1, 8, 8, 32, device='cuda') + 400.0)), packed, 8)
    assert ops.split_overflow(word=words[0]) and not ops.split_overflow(word=words[1]) and not ops.split_overflow()


def test_conv3x3_split_stride2_shuffle_bn(ops):
    g = torch.Generator().manual_seed(17)
    x = torch.randn(2, 32, 12, 36, generator=g)
    wt = torch.randn(64, 32, 3, 3, generator=g) * 0.06
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), None, stride=2, padding=1) * sc.double()[None, :, None, None] \
        + sh.double()[None, :, None, None]
    pk = ops.pack_conv3x3_split(wt, None, Cin=32, Cout=64, device='cuda', bn_scale=sc, bn_shift=sh)
    xs = ops.c8_to_c8s(ops.to_c8(x.cuda()))
    got = ops.from_c8(ops.c8s_to_float(ops.conv3x3_c8s(xs, pk, 64, stride2=True))).cpu()
    assert got.shape == ref.shape and rel_l2(got.numpy(), ref.numpy()) < 2e-6
    # 128 output channels (COB = 4), PixelShuffle store in fp32 + skip connection added by c8_add_to_c8s
    x = torch.randn(1, 64, 10, 33, generator=g)
    wt = torch.randn(128, 64, 3, 3, generator=g) * 0.04
    res = torch.randn(1, 32, 20, 66, generator=g)
    ref = torch.nn.functional.pixel_shuffle(torch.nn.functional.conv2d(x.double(), wt.double(), None, padding=1), 2) + res.double()
    pk = ops.pack_conv3x3_split(wt, None, Cin=64, Cout=128, device='cuda')
    sh32 = ops.conv3x3_c8s(ops.c8_to_c8s(ops.to_c8(x.cuda())), pk, 128, shuffle=True)
    got = ops.from_c8(ops.c8s_to_float(ops.c8_add_to_c8s(sh32, ops.c8_to_c8s(ops.to_c8(res.cuda()))))).cpu()
    assert got.shape == ref.shape and rel_l2(got.numpy(), ref.numpy()) < 2e-6
    # the same in one launch: PixelShuffle + skip add stored straight into c8s (epilogue flag bit6)
    res_s = ops.c8_to_c8s(ops.to_c8(res.cuda()))
    fused = ops.conv3x3_c8s(ops.c8_to_c8s(ops.to_c8(x.cuda())), pk, 128, shuffle='c8s', residual=res_s)
    got = ops.from_c8(ops.c8s_to_float(fused)).cpu()
    assert got.shape == ref.shape and rel_l2(got.numpy(), ref.numpy()) < 2e-6
    plain = ops.conv3x3_c8s(ops.c8_to_c8s(ops.to_c8(x.cuda())), pk, 128, shuffle='c8s')
    assert rel_l2(ops.from_c8(ops.c8s_to_float(plain)).cpu().numpy(), (ref - res.double()).numpy()) < 2e-6


# ------------------------------------------------------------------ DDnet (deep demosaicking) glue and forward
def test_ddnet_glue_kernels_vs_torch(ops):
    g = torch.Generator().manual_seed(5)
    F_, C_, h, w, E = 5, 4, 10, 14, 7
    src = torch.randn(F_, C_, h, w, generator=g).cuda()
    idx = torch.randint(0, F_, (E, 3), generator=g).to(torch.int32).cuda()
    scale = (1 + 0.1 * torch.randn(E, 3, C_, generator=g)).cuda()
    ref = torch.stack([torch.cat([src[idx[e, i]] * scale[e, i][:, None, None] for i in range(3)]) for e in range(E)])
    out = torch.empty(E, 2, h, w, 8, device='cuda')
    ops.ddnet_gather(src, idx, scale, out, C_, h, w)
    got = ops.from_c8(out)
    assert torch.equal(got[:, :12], ref) and not got[:, 12:].any()
    outs = torch.empty(E, 2, 2, h, w, 8, dtype=torch.float16, device='cuda')
    ops.ddnet_gather(src, idx, scale, outs, C_, h, w)
    assert rel_l2(ops.from_c8(ops.c8s_to_float(outs))[:, :12].cpu().numpy(), ref.cpu().numpy()) < 1e-6
    # finish: in1 + x, with the one-channel centre frame broadcast over three outputs
    src1 = torch.randn(F_, 1, h, w, generator=g).cuda()
    sc1 = (1 + 0.1 * torch.randn(E, 3, 1, generator=g)).cuda()
    x8 = torch.randn(E, 1, h, w, 8, generator=g).cuda()
    fin = ops.ddnet_finish(src1, idx, sc1, x8, torch.empty(E, 3, h, w, device='cuda'), 1, 3, h, w)
    ref = torch.stack([src1[idx[e, 1]] * sc1[e, 1, 0] + ops.from_c8(x8)[e, :3] for e in range(E)])
    assert torch.equal(fin, ref)
    # bilinear x2, align_corners=True
    p4 = torch.randn(E, 4, h, w, generator=g).cuda()
    up = ops.from_c8(ops.bilinear_up2_c8(p4, torch.empty(E, 1, 2 * h, 2 * w, 8, device='cuda')))
    ref = torch.nn.functional.interpolate(p4.cpu(), scale_factor=2, mode='bilinear', align_corners=True)
    assert rel_l2(up[:, :4].cpu().numpy(), ref.numpy()) < 1e-6 and not up[:, 4:].any()
    # mix
    br = torch.randn(6, 3, h, w, generator=g).cuda()
    gates = torch.randn(2, 3, generator=g).cuda()
    mix = ops.ddnet_mix(br, gates, torch.empty(3, 3, h, w, device='cuda'))
    assert torch.equal(mix, gates[0][None, :, None, None] * br[:3] + gates[1][None, :, None, None] * br[3:])


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_ddnet_forward_vs_reference_golden(precision, monkeypatch):
    """all frames of a 32x48x8 cube through the three-branch demosaicker; golden produced by the reference's
    test_ddnet on the same synthetic weights (non-trivial gate scalars)."""
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', precision)
    from adaptivepnp_sci_amd import test_ddnet as ddnet_plugin
    from oracle.nets import cpu_data_parallel, synth_ddnet_weights
    from oracle.sci_ops import one_to_three_channel
    g = load_gold('ddnet_forward')
    net = cpu_data_parallel(synth_ddnet_weights(0))
    out = ddnet_plugin(dev(one_to_three_channel(torch.from_numpy(g['mosaic']))), None, None, net)
    err = rel_l2(out.cpu().numpy(), g['out'])
    assert err <= 2e-6, err


def test_frame_metrics_vs_skimage_restatement(ops):
    """device PSNR / SSIM per frame vs the host restatement of scikit-image (itself pinned to the real skimage 0.18.3
    golden in tests/test_oracle_golden.py)"""
    from adaptivepnp_sci_amd.metrics import frame_metrics
    from oracle.metrics import psnr_frames, ssim_frames
    rng = np.random.default_rng(2)
    a = rng.random((40, 52, 5)).astype(np.float32)
    b = np.clip(a + 0.05 * rng.standard_normal(a.shape), 0, 1).astype(np.float32)
    p, s = frame_metrics(ops.mosaic_to_state(dev(a)), ops.mosaic_to_state(dev(b)))
    assert np.abs(np.array(p) - np.array(psnr_frames(a, b))).max() < 1e-9
    assert np.abs(np.array(s) - np.array(ssim_frames(a, b))).max() < 1e-12
    with pytest.raises(Exception):
        frame_metrics(ops.mosaic_to_state(dev(a[:4, :4])), ops.mosaic_to_state(dev(b[:4, :4])))


def test_winograd_wgrad_vs_autograd(ops):
    """the fp32 weight gradient in the Winograd domain (csrc/wgrad_wino.hip: dg = G^T [sum_tiles (A dY A^T) .* (B^T d B)] G)
    against PyTorch autograd in float64 and against the direct-form kernel: full, head-like, wide and ragged shapes (odd
    heights / widths, fewer tiles than one chunk, several images, more output channels than one 96-channel chunk)"""
    import ctypes as C
    from adaptivepnp_sci_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(13)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    cases = ((2, 96, 96, 96, 96, 10, 37, 16), (2, 13, 96, 16, 96, 9, 5, 3), (3, 96, 12, 96, 16, 7, 33, 5),
             (1, 64, 128, 64, 128, 16, 16, 7), (2, 32, 64, 32, 64, 2, 2, 1), (1, 128, 256, 128, 256, 6, 20, 4),
             (4, 24, 40, 24, 40, 12, 18, 255))
    for n, ci_r, co_r, ci, co, h, w, nslab in cases:
        x = torch.relu(torch.randn(n, ci_r, h, w, generator=g))
        wt = (torch.randn(co_r, ci_r, 3, 3, generator=g) * 0.05).double().requires_grad_()
        dz = torch.randn(n, co_r, h, w, generator=g)
        torch.nn.functional.conv2d(x.double(), wt, None, padding=1).backward(dz.double())
        x8, dz8 = ops.to_c8(x.cuda()), ops.to_c8(dz.cuda())
        ws = torch.empty(lib.scipnp_conv3x3_wgrad_wino_workspace_floats(ci, co, nslab), device='cuda')
        dW = torch.full((co_r, ci_r, 3, 3), float('nan'), device='cuda')
        _lib.check(lib.scipnp_conv3x3_wgrad_wino(p(x8), p(dz8), p(dW), p(ws), nslab, n, ci_r, co_r, ci, co, h, w, s), 'wgrad wino')
        err = rel_l2(dW.cpu().numpy(), wt.grad.numpy())
        assert err < 1e-6, (n, ci_r, co_r, h, w, nslab, err)
        ws2 = torch.empty(lib.scipnp_conv3x3_wgrad_workspace_floats(ci, co, 16), device='cuda')
        dW2 = torch.empty(co_r, ci_r, 3, 3, device='cuda')
        _lib.check(lib.scipnp_conv3x3_wgrad(p(x8), p(dz8), p(dW2), p(ws2), 16, n, ci_r, co_r, ci, co, h, w, s), 'wgrad')
        assert rel_l2(dW.cpu().numpy(), dW2.cpu().numpy()) < 2e-6
        # deterministic: fixed-order slab reduction, no atomics
        dW3 = torch.empty_like(dW)
        _lib.check(lib.scipnp_conv3x3_wgrad_wino(p(x8), p(dz8), p(dW3), p(ws), nslab, n, ci_r, co_r, ci, co, h, w, s), 'wgrad wino')
        assert torch.equal(dW, dW3)


def test_winograd_f4_wgrad_vs_autograd(ops):
    """the fp32 weight gradient in the Winograd F(4x4) domain (csrc/wgrad_wino4.hip: 36 positions, 32 x 32 channel blocks per
    workgroup, 8 waves with producer / consumer roles) against PyTorch autograd in float64 and the F(2x2) form: full, head-like,
    tail-like, wide and ragged shapes (heights / widths that are no multiples of 4, fewer tiles than one chunk, several images,
    slab counts with and without the XCD placement, more slabs than chunks)"""
    import ctypes as C
    from adaptivepnp_sci_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(14)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    cases = ((2, 96, 96, 96, 96, 10, 37, 16), (2, 13, 96, 16, 96, 9, 5, 3), (3, 96, 12, 96, 16, 7, 33, 5),
             (1, 64, 128, 64, 128, 16, 16, 8), (2, 32, 64, 32, 64, 2, 2, 1), (1, 128, 256, 128, 256, 6, 20, 4),
             (4, 24, 40, 24, 40, 12, 18, 64), (1, 96, 96, 96, 96, 64, 96, 24), (2, 32, 32, 32, 32, 33, 70, 7))
    for n, ci_r, co_r, ci, co, h, w, nslab in cases:
        x = torch.relu(torch.randn(n, ci_r, h, w, generator=g))
        wt = (torch.randn(co_r, ci_r, 3, 3, generator=g) * 0.05).double().requires_grad_()
        dz = torch.randn(n, co_r, h, w, generator=g)
        torch.nn.functional.conv2d(x.double(), wt, None, padding=1).backward(dz.double())
        x8, dz8 = ops.to_c8(x.cuda()), ops.to_c8(dz.cuda())
        ws = torch.full((lib.scipnp_conv3x3_wgrad_wino4_workspace_floats(ci, co, nslab),), float('nan'), device='cuda')
        dW = torch.full((co_r, ci_r, 3, 3), float('nan'), device='cuda')
        _lib.check(lib.scipnp_conv3x3_wgrad_wino4(p(x8), p(dz8), p(dW), p(ws), nslab, n, ci_r, co_r, ci, co, h, w, s), 'wgrad wino4')
        err = rel_l2(dW.cpu().numpy(), wt.grad.numpy())
        assert err < 1e-5, (n, ci_r, co_r, h, w, nslab, err)
        ws2 = torch.empty(lib.scipnp_conv3x3_wgrad_wino_workspace_floats(ci, co, 16), device='cuda')
        dW2 = torch.empty(co_r, ci_r, 3, 3, device='cuda')
        _lib.check(lib.scipnp_conv3x3_wgrad_wino(p(x8), p(dz8), p(dW2), p(ws2), 16, n, ci_r, co_r, ci, co, h, w, s), 'wgrad wino')
        assert rel_l2(dW.cpu().numpy(), dW2.cpu().numpy()) < 1e-5
        # deterministic: fixed-order slab reduction, no atomics
        dW3 = torch.empty_like(dW)
        _lib.check(lib.scipnp_conv3x3_wgrad_wino4(p(x8), p(dz8), p(dW3), p(ws), nslab, n, ci_r, co_r, ci, co, h, w, s), 'wgrad wino4')
        assert torch.equal(dW, dW3)
    assert lib.scipnp_conv3x3_wgrad_wino4(p(x8), p(dz8), p(dW), p(ws), 0, n, ci_r, co_r, ci, co, h, w, s) != 0


def test_multi_layer_entries_equal_their_single_layer_calls(ops):
    """round 5: the trainer's many tiny per-layer launches as ONE launch per kind -- scipnp_pack_conv3x3_device_multi,
    scipnp_pack_conv3x3_wino4_multi, scipnp_conv3x3_wgrad_wino4 with dW = NULL + scipnp_conv3x3_wgrad_wino4_finish_multi,
    scipnp_conv_bias_grad with db = NULL + scipnp_conv_bias_grad_reduce_multi: bit-identical to the per-layer calls, for layers of
    different shapes in one table (narrow head / tail, padded channels, transposed packs), more jobs than one table holds, n = 0"""
    import ctypes as C
    from adaptivepnp_sci_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(515)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    layers = [(13, 96, 16, 96), (96, 96, 96, 96), (96, 12, 96, 16), (24, 40, 24, 40)] * 10          # 40 layers > 32 per table
    ws_ = [torch.randn(co_r, ci_r, 3, 3, generator=g).cuda() for ci_r, co_r, _, _ in layers]
    bs_ = [torch.randn(co_r, generator=g).cuda() for _, co_r, _, _ in layers]
    jobs, single = [], []
    for (ci_r, co_r, ci, co), w, b in zip(layers, ws_, bs_):
        for tr in (0, 1):
            n = lib.scipnp_conv3x3_packed_floats(co, ci) if tr else lib.scipnp_conv3x3_packed_floats(ci, co)
            a, b_ = torch.full((n,), float('nan'), device='cuda'), torch.full((n,), float('nan'), device='cuda')
            jobs.append((w, None if tr else b, a, ci_r, co_r, ci, co, tr))
            _lib.check(lib.scipnp_pack_conv3x3_device(p(w), None if tr else p(b), p(b_), ci_r, co_r, ci, co, tr, s), 'pack')
            single.append(b_)
    ops.pack_conv3x3_device_multi(jobs)
    assert all(torch.equal(j[2], r) for j, r in zip(jobs, single))
    ops.pack_conv3x3_device_multi([])                                            # nothing to do
    # F(4x4) packs from the forward packs
    fw = [(j[2], j[5], j[6]) for j in jobs if not j[7]]
    outs = [torch.full((lib.scipnp_conv3x3_wino4_packed_floats(ci, co),), float('nan'), device='cuda') for _, ci, co in fw]
    ops.pack_conv3x3_wino4_multi([(pk, o, ci, co) for (pk, ci, co), o in zip(fw, outs)])
    for (pk, ci, co), o in zip(fw, outs):
        assert torch.equal(o, ops.pack_conv3x3_wino4(pk, ci, co))
    # weight / bias gradients: slabs and partials first, one finish for all layers
    n, h, w = 2, 13, 37
    four = layers[:4]
    xs = [ops.to_c8(torch.relu(torch.randn(n, ci_r, h, w, generator=g)).cuda()) for ci_r, _, _, _ in four]
    dzs = [ops.to_c8(torch.randn(n, co_r, h, w, generator=g).cuda()) for _, co_r, _, _ in four]
    nsl = [5, 3, 7, 64]
    wss = [torch.full((lib.scipnp_conv3x3_wgrad_wino4_workspace_floats(ci, co, k),), float('nan'), device='cuda')
           for (_, _, ci, co), k in zip(four, nsl)]
    bws = [torch.full(((co // 8) * 64 * 8,), float('nan'), device='cuda') for _, _, _, co in four]
    dW = [torch.full((co_r, ci_r, 3, 3), float('nan'), device='cuda') for ci_r, co_r, _, _ in four]
    db = [torch.full((co_r,), float('nan'), device='cuda') for _, co_r, _, _ in four]
    for (ci_r, co_r, ci, co), x8, dz8, ws, bw, k in zip(four, xs, dzs, wss, bws, nsl):
        _lib.check(lib.scipnp_conv3x3_wgrad_wino4(p(x8), p(dz8), None, p(ws), k, n, ci_r, co_r, ci, co, h, w, s), 'slabs')
        _lib.check(lib.scipnp_conv_bias_grad(p(dz8), None, p(bw), n, co_r, co, h, w, s), 'partials')
    P, I = C.c_void_p * 4, C.c_int * 4
    _lib.check(lib.scipnp_conv3x3_wgrad_wino4_finish_multi(4, P(*[t.data_ptr() for t in wss]), P(*[t.data_ptr() for t in dW]), I(*nsl),
                                                           I(*[l[0] for l in four]), I(*[l[1] for l in four]), I(*[l[2] for l in four]),
                                                           I(*[l[3] for l in four]), s), 'finish')
    _lib.check(lib.scipnp_conv_bias_grad_reduce_multi(4, P(*[t.data_ptr() for t in bws]), P(*[t.data_ptr() for t in db]),
                                                      I(*[l[1] for l in four]), s), 'reduce')
    for (ci_r, co_r, ci, co), x8, dz8, k, dw_m, db_m in zip(four, xs, dzs, nsl, dW, db):
        ws1 = torch.empty(lib.scipnp_conv3x3_wgrad_wino4_workspace_floats(ci, co, k), device='cuda')
        dw1, db1, bw1 = torch.empty_like(dw_m), torch.empty_like(db_m), torch.empty((co // 8) * 64 * 8, device='cuda')
        _lib.check(lib.scipnp_conv3x3_wgrad_wino4(p(x8), p(dz8), p(dw1), p(ws1), k, n, ci_r, co_r, ci, co, h, w, s), 'wgrad')
        _lib.check(lib.scipnp_conv_bias_grad(p(dz8), p(db1), p(bw1), n, co_r, co, h, w, s), 'bgrad')
        assert torch.equal(dw_m, dw1) and torch.equal(db_m, db1), (ci_r, co_r)
    assert lib.scipnp_conv3x3_wgrad_wino4_finish_multi(1, P(None, None, None, None), P(*[t.data_ptr() for t in dW]), I(*nsl), I(1, 1, 1, 1),
                                                       I(1, 1, 1, 1), I(8, 8, 8, 8), I(8, 8, 8, 8), s) != 0      # null workspace refused


def test_split_wgrad_bgrad_backward_data_vs_autograd(ops):
    """the finetune's split-fp16 kernels: weight / bias gradients from c8s operands (transposing LDS reads, pre-scaled
    dZ) and the backward-data conv with the ReLU-mask epilogue, against PyTorch autograd in float64"""
    import ctypes as C
    from adaptivepnp_sci_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(12)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    n, h, w, nslab, S = 2, 11, 37, 16, 1024.0
    for ci_r, co_r, ci, co in ((96, 96, 96, 96), (13, 96, 16, 96), (96, 12, 96, 16), (64, 128, 64, 128)):
        x = torch.relu(torch.randn(n, ci_r, h, w, generator=g))
        wt = (torch.randn(co_r, ci_r, 3, 3, generator=g) * 0.05).double().requires_grad_()
        bias = torch.zeros(co_r, dtype=torch.float64, requires_grad=True)
        dz = torch.randn(n, co_r, h, w, generator=g) * 1e-3
        xd = x.double().requires_grad_()
        torch.nn.functional.conv2d(xd, wt, bias, padding=1).backward(dz.double())
        xs = ops.c8_to_c8s(ops.to_c8(x.cuda()))
        dzs = ops.c8_scale_to_c8s(ops.to_c8(dz.cuda()), scale=S)
        ws = torch.empty(lib.scipnp_conv3x3_wgrad_workspace_floats(ci, co, nslab), device='cuda')
        dW = torch.empty(co_r, ci_r, 3, 3, device='cuda')
        _lib.check(lib.scipnp_conv3x3_wgrad_split(p(xs), p(dzs), p(dW), p(ws), nslab, n, ci_r, co_r, ci, co, h, w, 1 / S, s),
                   'wgrad split')
        assert rel_l2(dW.cpu().numpy(), wt.grad.numpy()) < 2e-6, (ci_r, co_r, rel_l2(dW.cpu().numpy(), wt.grad.numpy()))
        db = torch.empty(co_r, device='cuda')
        bws = torch.empty((co // 8) * 64 * 8, device='cuda')
        _lib.check(lib.scipnp_conv_bias_grad_split(p(dzs), p(db), p(bws), n, co_r, co, h, w, 1 / S, s), 'bgrad split')
        assert rel_l2(db.cpu().numpy(), bias.grad.numpy()) < 2e-6
        # backward-data: conv of dZ with the transposed / flipped weights (packed on the device), masked by x > 0
        pk = torch.empty(lib.scipnp_conv3x3_split_packed_bytes(co, ci), dtype=torch.uint8, device='cuda')
        ops.pack_conv3x3_split_device(wt.detach().float().cuda().contiguous(), None, pk, ci, co, transpose=True)
        dx = ops.conv3x3_c8s(dzs, pk, ci, mask=xs)
        got = ops.from_c8(ops.c8s_to_c8(dx, scale=1 / S))[:, :ci_r].cpu()
        ref = xd.grad * (x > 0)
        assert rel_l2(got.numpy(), ref.numpy()) < 2e-6, (ci_r, co_r)


def test_denoiser_plugins_with_online_update_vs_oracle(ffdnet_state_dict):
    """the reference's plug-in entry points called directly with updata_=True (finetune on the measurement loss, then
    denoise): ffdnet_rgb_denoise_full_tensor (test_ffdnet_ipol.py:240-359) and fastdvdnet_denoiser_full_tensor_v2
    (test_fastdvdnet.py:325-500) on reference-layout yall (M,N,4) / Phiall (M,N,B,4)"""
    from adaptivepnp_sci_amd import fastdvdnet_denoiser_full_tensor_v2, ffdnet_rgb_denoise_full_tensor, synth
    from adaptivepnp_sci_amd.nets import FFDNet
    from oracle import denoisers as OD
    from oracle import nets as ON
    from oracle import sci_ops as OO
    y, Phi, orig = synth.make_problem(48, 64, 8, seed=17)
    yall, Phiall, _ps, _x0 = OO.setup_planes(torch.from_numpy(y), torch.from_numpy(Phi))
    rng = np.random.default_rng(3)
    x = torch.from_numpy(np.clip(np.repeat(orig[:, :, None, :], 3, 2) + 0.05 * rng.standard_normal((48, 64, 3, 8)), 0, 1)
                         .astype(np.float32))
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    out, net2 = ffdnet_rgb_denoise_full_tensor(dev(x), dev(yall), dev(Phiall), 25 / 255, net, True, 2e-6, True, 2)
    onet = ON.OracleFFDNet()
    onet.load_state_dict(ffdnet_state_dict)
    ref, onet2 = OD.ffdnet_pass(x.clone(), yall, Phiall, 25 / 255, onet, 2e-6, True, 2)
    assert net2 is net and rel_l2(out.cpu().numpy(), ref.detach().numpy()) <= 1e-5
    w_got, w_ref = net.state_dict()['model.2.weight'], onet2.state_dict()['model.2.weight']
    assert not torch.equal(w_got, ffdnet_state_dict['model.2.weight'])
    assert rel_l2((w_got - ffdnet_state_dict['model.2.weight']).numpy(), (w_ref - ffdnet_state_dict['model.2.weight']).numpy()) < 2e-2
    # FastDVDnet: same noise for both sides (the reference draws it from the global NumPy RNG)
    fnet = torch.nn.DataParallel(synth.synth_fastdvdnet(0))
    onet = ON.cpu_data_parallel(ON.synth_fastdvdnet_weights(0))
    np.random.seed(5)
    out, _ = fastdvdnet_denoiser_full_tensor_v2(dev(x), 8 / 255, dev(yall), dev(Phiall), fnet, True, 2e-6, True, 1)
    np.random.seed(5)
    ref, _ = OD.fastdvdnet_pass(x.clone(), 8 / 255, yall, Phiall, onet, 2e-6, True, 1)
    assert rel_l2(out.cpu().numpy(), ref.detach().numpy()) <= 1e-5


def test_ffdnet_single_call_c_entries_equal_the_layerwise_path(ffdnet_state_dict, monkeypatch):
    """scipnp_ffdnet_forward (fp32 direct), scipnp_ffdnet_forward_c8w (fp32 Winograd) and scipnp_ffdnet_forward_c8s
    (split-fp16): the whole 12-layer pass as ONE C-ABI call each, bit-identical to the layer-by-layer launches the solver
    issues"""
    from adaptivepnp_sci_amd.nets import FFDNet, FFDNetEngine
    net = FFDNet()
    net.load_state_dict(ffdnet_state_dict)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(3, 16, 20, 36, generator=g)
    x[:, 12] = 25 / 255
    x[:, 13:] = 0
    for prec in ('f32', 'f32-direct', 'f16x3'):
        monkeypatch.setenv('SCIPNP_F32_CONV', 'direct' if prec == 'f32-direct' else 'winograd')
        eng = FFDNetEngine(net, 3, 20, 36, torch.device('cuda'), precision=prec.split('-')[0])
        assert (eng.packed_wino is not None) == (prec == 'f32')
        eng.in_c8.copy_(__import__('adaptivepnp_sci_amd').ops.to_c8(x.cuda()))
        if prec == 'f16x3':
            from adaptivepnp_sci_amd import ops as O
            eng.in_c8s.copy_(O.c8_to_c8s(eng.in_c8))
            a = eng.forward().clone()
            b = eng.forward_c_entry_split().clone()
        else:
            a = eng.forward().clone()
            b = eng.forward_c_entry().clone()
        assert torch.equal(a, b), prec
        if prec == 'f16x3':
            # scipnp_ffdnet_forward_c8s_2s: half of the frames on the CALLER's side stream, forked from and joined to the
            # calling stream through the caller's two events inside the call (the library creates neither) -- bit-identical,
            # and legal under stream capture as include/scipnp.h promises: capture the call in a hipGraph and replay it
            side = (torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event())
            assert torch.equal(eng.forward_c_entry_split(side=side), a), 'two-stream C entry'
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=st):
                    out = eng.forward_c_entry_split(side=side)
                out.zero_()
                graph.replay()
            torch.cuda.current_stream().wait_stream(st)
            torch.cuda.synchronize()
            assert torch.equal(out, a), 'hipGraph replay of the two-stream C entry'


def test_pm_project_with_more_than_16_frames(ops):
    """B in 17..32 takes the two-pixels-per-thread path; bit-exact against the oracle's projection"""
    from oracle import sci_ops as OO
    rng = np.random.default_rng(8)
    M, N, B = 12, 20, 23
    theta, b = rng.random((M, N, B, 4), np.float32), (0.2 * rng.standard_normal((M, N, B, 4))).astype(np.float32)
    Phi = (rng.random((M, N, B, 4)) < 0.5).astype(np.float32)
    y = (rng.random((M, N, 4)) * B / 2).astype(np.float32)
    Phisum = Phi.sum(2)
    Phisum[Phisum == 0] = 1
    T = torch.from_numpy
    ref = OO.project_two_stage(T(theta), T(b), T(Phi), T(y), T(Phisum), 0.55, 1.0)
    pm = lambda a: dev(np.ascontiguousarray(a.transpose(2, 3, 0, 1)))  # noqa: E731  (M,N,B,4) -> [B][4][M][N]
    th_d = pm(theta)
    out = ops.pm_project(th_d, pm(b), pm(Phi), dev(np.ascontiguousarray(y.transpose(2, 0, 1))),
                         dev(np.ascontiguousarray(Phisum.transpose(2, 0, 1))), 0, np.float32(1 / 0.55), np.float32(0.55),
                         out=torch.empty_like(th_d))
    assert np.array_equal(out.cpu().numpy(), ref.numpy().transpose(2, 3, 0, 1))


@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_ffdnet_gray_forward_vs_reference_golden(precision, monkeypatch):
    """the grayscale FFDNet of the reference's model zoo (ffdnet_gray.pth: 5 -> 64 x 13 -> 4 channels, 15 layers) on the
    HIP kernels; golden = the reference network class on the reference weights (odd image size included)"""
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', precision)
    from adaptivepnp_sci_amd.nets import FFDNet
    gw, g = load_gold('ffdnet_gray_weights'), load_gold('ffdnet_gray_forward')
    net = FFDNet(in_nc=1, out_nc=1, nc=64, nb=15)
    net.load_state_dict({k: torch.from_numpy(gw[k]) for k in gw.files})
    for tag in ('2x64x96', '1x37x50'):
        for s in (10, 40):
            out = net(dev(g[f'in_{tag}']), s / 255.)
            err = rel_l2(out.cpu().numpy(), g[f'out_{tag}_s{s}'])
            assert err <= 2e-6, (tag, s, err)


@pytest.mark.parametrize('cin,cout,h,w,n', [(16, 96, 37, 45, 2), (96, 96, 64, 64, 2), (96, 16, 33, 31, 1), (32, 32, 16, 32, 3),
                                            (128, 128, 20, 70, 1), (8, 40, 5, 3, 1)])
def test_conv3x3_winograd_fp32_vs_fp64(ops, cin, cout, h, w, n):
    """fp32 Winograd F(2x2,3x3) on the fp32 MFMA (csrc/conv_wino.hip) against an fp64 convolution and against the
    direct fp32-MFMA kernel: ragged sizes (odd heights / widths cut 2x2 tiles at the border), every epilogue"""
    g = torch.Generator().manual_seed(cin * 991 + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    fwd = torch.randn(n, cout, h, w, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    packed = ops.pack_conv3x3(wt, bias, Cin=cin, Cout=cout, device='cuda')
    pw = ops.pack_conv3x3_wino(packed, cin, cout)
    xc = ops.to_c8(x.cuda())
    got = ops.from_c8(ops.conv3x3_c8w(xc, pw, cout)).cpu()
    direct = ops.from_c8(ops.conv3x3_c8(xc, packed, cout)).cpu()
    # F(2x2,3x3) in fp32: the transforms add a few roundings per product (|G g G^T| <= |g|, B^T d B sums 4 inputs);
    # measured 2e-7 .. 6e-7 against fp64, the direct kernel 1.5e-7 .. 4.7e-7
    assert rel_l2(got.numpy(), ref.numpy()) < 1.5e-6, rel_l2(got.numpy(), ref.numpy())
    assert rel_l2(got.numpy(), direct.numpy()) < 1.5e-6
    got = ops.from_c8(ops.conv3x3_c8w(xc, pw, cout, relu=True, residual=ops.to_c8(res.cuda()))).cpu()
    assert rel_l2(got.numpy(), torch.relu(ref + res.double()).numpy()) < 1.5e-6
    fw8 = ops.to_c8(fwd.cuda())
    got = ops.from_c8(ops.conv3x3_c8w(xc, pw, cout, mask_src=fw8, residual=ops.to_c8(res.cuda()))).cpu()
    want = torch.where(fwd > 0, ref + res.double(), torch.zeros_like(ref))
    assert rel_l2(got.numpy(), want.numpy()) < 1.5e-6
    # the 16-row / 8-wave workgroup form runs the same arithmetic per tile
    got16 = ops.from_c8(ops.conv3x3_c8w(xc, pw, cout, mask_src=fw8, residual=ops.to_c8(res.cuda()), rows16=True)).cpu()
    assert torch.equal(got16, got)


def test_conv3x3_winograd_identity_asymmetric(ops):
    """centre-tap channel permutation through the Winograd kernel: catches row / column swaps of the 16x16x4 MFMA
    operand and accumulator maps (exact: U = G g G^T of a centre tap is 0.25 / 0.5 / 1 patterns, sums are exact)"""
    cin = cout = 32
    wt = torch.zeros(cout, cin, 3, 3)
    for c in range(cout):
        wt[c, (c * 7 + 3) % cin, 1, 1] = 1.0
    x = (torch.arange(1 * cin * 8 * 32, dtype=torch.float32).reshape(1, cin, 8, 32) % 251) * 0.25
    packed = ops.pack_conv3x3(wt, None, Cin=cin, Cout=cout, device='cuda')
    got = ops.from_c8(ops.conv3x3_c8w(ops.to_c8(x.cuda()), ops.pack_conv3x3_wino(packed, cin, cout), cout)).cpu()
    perm = [(c * 7 + 3) % cin for c in range(cout)]
    assert torch.equal(got, x[:, perm])


def test_conv3x3_winograd_pixel_shuffle_store(ops):
    """PixelShuffle(2) folded into the Winograd kernel's store (flags bit3: the UpBlocks of FastDVDnet / DDnet in fp32),
    with and without the skip tensor and ReLU, against float64 conv + pixel_shuffle and against the direct kernel's shuffle
    epilogue; ragged heights / widths, several images, 96 / 128 / 160 / 256 conv channels"""
    g = torch.Generator().manual_seed(21)
    for n, cin, cout, h, w in ((2, 64, 128, 16, 32), (1, 128, 256, 9, 21), (3, 40, 160, 5, 7), (2, 24, 96, 13, 70)):
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        b = torch.randn(cout, generator=g) * 0.1
        res = torch.randn(n, cout // 4, 2 * h, 2 * w, generator=g)
        cin_p = (cin + 7) // 8 * 8
        pk = ops.pack_conv3x3(wt, b, Cin=cin_p, Cout=cout, device='cuda')
        pw = ops.pack_conv3x3_wino(pk, cin_p, cout)
        xc, rc = ops.to_c8(x.cuda()), ops.to_c8(res.cuda())
        ref = torch.nn.functional.pixel_shuffle(torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), padding=1), 2)
        for residual, relu in ((None, False), (rc, False), (rc, True)):
            want = ref + (res.double() if residual is not None else 0)
            want = torch.relu(want) if relu else want
            got = ops.conv3x3_c8w(xc, pw, cout, relu=relu, residual=residual, shuffle=True)
            assert got.shape == (n, cout // 32, 2 * h, 2 * w, 8)
            err = rel_l2(ops.from_c8(got)[:, :cout // 4].cpu().numpy(), want.numpy())
            assert err < 2e-6, (n, cin, cout, h, w, relu, err)
            direct = ops.conv3x3_c8(xc, pk, cout, relu=relu, residual=residual, shuffle=True)
            assert rel_l2(got.cpu().numpy(), direct.cpu().numpy()) < 2e-6


def test_conv3x3_winograd_random_shapes(ops):
    """seeded sweep of small / ragged / thin shapes through the fp32 Winograd kernel (both workgroup forms) against the
    direct fp32-MFMA kernel and fp64: tiles cut by every border, single channel groups, co-blocks with padding"""
    rng = np.random.default_rng(2024)
    g = torch.Generator().manual_seed(2024)
    for _ in range(40):
        n = int(rng.integers(1, 4))
        cin, cout = 8 * int(rng.integers(1, 7)), 8 * int(rng.integers(1, 14))
        h, w = int(rng.integers(1, 41)), int(rng.integers(1, 75))
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        bias = torch.randn(cout, generator=g)
        ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1)
        packed = ops.pack_conv3x3(wt, bias, Cin=cin, Cout=cout, device='cuda')
        pw = ops.pack_conv3x3_wino(packed, cin, cout)
        xc = ops.to_c8(x.cuda())
        got = ops.conv3x3_c8w(xc, pw, cout)
        assert torch.equal(got, ops.conv3x3_c8w(xc, pw, cout, rows16=True)), (n, cin, cout, h, w)
        err = rel_l2(ops.from_c8(got).cpu().numpy(), ref.numpy())
        assert err < 2e-6, (n, cin, cout, h, w, err)
        direct = ops.from_c8(ops.conv3x3_c8(xc, packed, cout)).cpu()
        assert rel_l2(ops.from_c8(got).cpu().numpy(), direct.numpy()) < 2e-6


def test_conv3x3_winograd_f4_random_shapes(ops):
    """csrc/conv_wino4.hip -- fp32 Winograd F(4x4,3x3): seeded sweep of small / ragged / thin shapes (tiles cut by every border,
    one and many channel groups, co-blocks with padding channels, several frames) and every epilogue against float64 and the
    direct fp32-MFMA kernel.  Measured 0.7e-6 .. 1.7e-6 relative L2 per layer on zero-mean random data (the transforms' 4, 5, 8
    multipliers; F(2x2): 1e-7 .. 2.6e-7), 7.7e-7 on the final iterate of the 28-iteration headline run against the oracle."""
    rng = np.random.default_rng(44)
    g = torch.Generator().manual_seed(44)
    shapes = [(1, 8, 32, 8, 64), (2, 96, 96, 20, 70), (1, 16, 96, 37, 129), (3, 64, 24, 5, 7), (1, 40, 72, 64, 64), (1, 8, 8, 1, 1),
              (2, 32, 32, 9, 65), (1, 96, 96, 66, 130)]
    for _ in range(24):
        shapes.append((int(rng.integers(1, 4)), 8 * int(rng.integers(1, 13)), 8 * int(rng.integers(1, 14)),
                       int(rng.integers(1, 41)), int(rng.integers(1, 150))))
    for n, cin, cout, h, w in shapes:
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        bias = torch.randn(cout, generator=g)
        res, fwd = torch.randn(n, cout, h, w, generator=g), torch.randn(n, cout, h, w, generator=g)
        ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1)
        packed = ops.pack_conv3x3(wt, bias, Cin=cin, Cout=cout, device='cuda')
        p4 = ops.pack_conv3x3_wino4(packed, cin, cout)
        xc, rc, fc = ops.to_c8(x.cuda()), ops.to_c8(res.cuda()), ops.to_c8(fwd.cuda())
        got = ops.from_c8(ops.conv3x3_c8w4(xc, p4, cout)).cpu()
        err = rel_l2(got.numpy(), ref.numpy())
        assert err < 4e-6, (n, cin, cout, h, w, err)
        direct = ops.from_c8(ops.conv3x3_c8(xc, packed, cout)).cpu()
        assert rel_l2(got.numpy(), direct.numpy()) < 4e-6
        got = ops.from_c8(ops.conv3x3_c8w4(xc, p4, cout, relu=True, residual=rc, head=True)).cpu()
        assert rel_l2(got.numpy(), torch.relu(ref + res.double()).numpy()) < 4e-6, (n, cin, cout, h, w)
        got = ops.from_c8(ops.conv3x3_c8w4(xc, p4, cout, mask_src=fc, residual=rc)).cpu()
        want = torch.where(fwd > 0, ref + res.double(), torch.zeros_like(ref))
        assert rel_l2(got.numpy(), want.numpy()) < 4e-6, (n, cin, cout, h, w)


def test_conv3x3_winograd_f4_pixel_shuffle_store(ops):
    """PixelShuffle(2) folded into the F(4x4,3x3) kernel's store (flags bit3: the UpBlocks of FastDVDnet / DDnet in fp32), with and
    without the skip tensor and ReLU, against float64 conv + pixel_shuffle and against the F(2x2,3x3) kernel's shuffle epilogue;
    ragged heights / widths, several images, 96 / 128 / 160 / 256 conv channels"""
    g = torch.Generator().manual_seed(22)
    for n, cin, cout, h, w in ((2, 64, 128, 16, 32), (1, 128, 256, 9, 21), (3, 40, 160, 5, 7), (2, 32, 96, 13, 70), (1, 64, 128, 64, 64)):
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        b = torch.randn(cout, generator=g) * 0.1
        res = torch.randn(n, cout // 4, 2 * h, 2 * w, generator=g)
        pk = ops.pack_conv3x3(wt, b, Cin=cin, Cout=cout, device='cuda')
        p4, p2 = ops.pack_conv3x3_wino4(pk, cin, cout), ops.pack_conv3x3_wino(pk, cin, cout)
        xc, rc = ops.to_c8(x.cuda()), ops.to_c8(res.cuda())
        ref = torch.nn.functional.pixel_shuffle(torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), padding=1), 2)
        for residual, relu in ((None, False), (rc, False), (rc, True)):
            want = ref + (res.double() if residual is not None else 0)
            want = torch.relu(want) if relu else want
            got = ops.conv3x3_c8w4(xc, p4, cout, relu=relu, residual=residual, shuffle=True)
            assert got.shape == (n, cout // 32, 2 * h, 2 * w, 8)
            err = rel_l2(ops.from_c8(got)[:, :cout // 4].cpu().numpy(), want.numpy())
            assert err < 4e-6, (n, cin, cout, h, w, relu, err)
            f2 = ops.conv3x3_c8w(xc, p2, cout, relu=relu, residual=residual, shuffle=True)
            assert rel_l2(got.cpu().numpy(), f2.cpu().numpy()) < 4e-6


def test_conv3x3_winograd_f4_is_what_the_engines_run(ops, monkeypatch):
    """conv3x3_c8w takes the F(4x4,3x3) kernel for a layer packed by pack_conv3x3_wino_both when the shape has one (at least
    24 output channels); SCIPNP_WINO_F4=0 at pack or launch time keeps the F(2x2,3x3) kernel; the FFDNet
    engine's C entry (scipnp_ffdnet_forward_c8w4) runs the same launches as its Python layer loop"""
    assert ops.wino_f4_shape(16, 96) and ops.wino_f4_shape(96, 96) and ops.wino_f4_shape(96, 24) and not ops.wino_f4_shape(96, 16)
    g = torch.Generator().manual_seed(45)
    x = ops.to_c8(torch.randn(2, 96, 24, 40, generator=g).cuda())
    packed = ops.pack_conv3x3(torch.randn(96, 96, 3, 3, generator=g) * 0.05, torch.randn(96, generator=g), Cin=96, Cout=96, device='cuda')
    monkeypatch.setenv('SCIPNP_WINO_F4', '1')
    both = ops.pack_conv3x3_wino_both(packed, 96, 96)
    assert both.f4 is not None
    ops.LAUNCH_LOG = log = []
    try:
        a = ops.conv3x3_c8w(x, both, 96, relu=True)
        monkeypatch.setenv('SCIPNP_WINO_F4', '0')
        b = ops.conv3x3_c8w(x, both, 96, relu=True)
    finally:
        ops.LAUNCH_LOG = None
    assert [e[0] for e in log] == ['conv3x3_c8w4_kernel', 'conv3x3_c8w_kernel']
    assert torch.equal(a, ops.conv3x3_c8w4(x, both.f4, 96, relu=True)) and torch.equal(b, ops.conv3x3_c8w(x, both.w, 96, relu=True))
    assert 0 < rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 4e-6
    assert ops.pack_conv3x3_wino_both(packed, 96, 96).f4 is None                 # (not packed while switched off)
    monkeypatch.setenv('SCIPNP_WINO_F4', '1')
    for cin, cout, f4 in ((8, 96, True), (96, 16, False), (24, 24, True), (24, 8, False), (96, 24, True)):
        # fewer than 24 output channels keep F(2x2,3x3); the input width does not matter (profiles/r05zj_narrow_layers.txt)
        pk = ops.pack_conv3x3(torch.zeros(cout, cin, 3, 3), None, Cin=cin, Cout=cout, device='cuda')
        assert (ops.pack_conv3x3_wino_both(pk, cin, cout).f4 is not None) == f4
    assert ops.conv3x3_c8w(x, both, 96, shuffle=True).shape == (2, 3, 48, 80, 8)   # (the PixelShuffle store is an epilogue of both kernels)


def test_tv_banded_kernel_random_shapes(ops):
    """seeded sweep of plane shapes and channel counts through the banded TV kernel against the tiled one: bit-identical
    `out` and stop iterations whatever the band / strip / wave seams and the iteration count"""
    rng = np.random.default_rng(77)
    for _ in range(40):
        M, N, C_ = int(rng.integers(1, 300)), int(rng.integers(1, 257)), int(rng.integers(1, 40))
        n_iter = int(rng.integers(1, 6))
        x = dev((rng.uniform(0, 1, (C_, M, N)) * rng.choice([1.0, 1.0, 60.0, 900.0], (C_, 1, 1))).astype(np.float32))
        b = dev(rng.normal(0, 0.1, (C_, M, N)).astype(np.float32))
        coef = float(rng.choice([-1.0, 1.0, 1 / 0.55]))
        p1, p3 = ops.TvPlan(M, N, C_, n_iter, x.device), ops.TvPlan(M, N, C_, n_iter, x.device)
        o1, o3 = torch.empty_like(x), torch.full_like(x, -5.0)
        ops.tv_chambolle(x, b, coef, o1, p1, 0.1, kernel=1)
        ops.tv_chambolle(x, b, coef, o3, p3, 0.1, kernel=3)
        assert torch.equal(p1.stop_iter, p3.stop_iter), (M, N, C_, n_iter)
        assert torch.equal(o1, o3), (M, N, C_, n_iter)


@pytest.mark.parametrize('shape', [(3, 8, 8), (8, 37, 70), (2, 128, 256)])
def test_mosaic_hand_off_between_the_pre_and_post_denoiser_kernels_is_bit_identical(ops, shape):
    """round 6: scipnp_pm_pre_denoise_mosaic stores the mosaic x + b/rho (4 bytes per state element) instead of x_rgb (12) and
    scipnp_pm_post_denoise_mosaic demosaicks it again for w += x_rgb - out -- every output of the pair (both denoiser-input layouts,
    theta, b, w, the out_store copy, the squared-error partials) equals the x_rgb pair's bit for bit, borders (reflect-101) included,
    with a planar and with a c8 denoiser output, in the first-iteration alias mode too"""
    B, M, N = shape
    g = torch.Generator().manual_seed(7 * M + N)
    x = torch.rand(B, 4, M, N, generator=g).cuda()
    b = ((torch.rand(B, 4, M, N, generator=g) - 0.5) * 0.2).cuda()
    w0 = ((torch.rand(B, 3, 2 * M, 2 * N, generator=g) - 0.5) * 0.1).cuda()
    orig = torch.rand(B, 4, M, N, generator=g).cuda()
    out_rgb = torch.rand(B, 3, 2 * M, 2 * N, generator=g).cuda() * 1.2 - 0.1
    out_c8 = torch.rand(B, 2, M, N, 8, generator=g).cuda() * 1.2 - 0.1
    for alias in (False, True):
        for planar in (True, False):
            res = []
            for mode in ('x_rgb', 'mosaic'):
                xs, bs, ws, th = x.clone(), b.clone(), w0.clone(), torch.empty_like(x)
                x_rgb, mos = torch.empty_like(w0), torch.full_like(x, -3.0)
                rgb_w, c8 = torch.empty_like(w0), torch.empty(B, 2, M, N, 8, device='cuda')
                part = torch.empty(ops.post_nblocks(M, N, B), dtype=torch.float64, device='cuda')
                store = torch.empty_like(w0) if not planar else None
                if mode == 'x_rgb':
                    ops.pm_pre_denoise(xs, bs, ws, x_rgb, rgb_w, c8, 1 / 0.55, 0.01, 25 / 255)
                    ops.pm_post_denoise(out_rgb if planar else None, None if planar else out_c8, store, xs, x_rgb, th, bs, ws, alias, orig, part)
                else:
                    ops.pm_pre_denoise(xs, bs, ws, None, rgb_w, c8, 1 / 0.55, 0.01, 25 / 255, mosaic=mos)
                    assert torch.equal(mos, x + np.float32(1 / 0.55) * b)
                    ops.pm_post_denoise(out_rgb if planar else None, None if planar else out_c8, store, xs, None, th, bs, ws, alias, orig, part,
                                        mosaic=mos)
                res.append((rgb_w, c8, xs, bs, ws, th, part, store))
            for a_, b_ in zip(*res):
                assert (a_ is None and b_ is None) or torch.equal(a_, b_), (shape, alias, planar)
    with pytest.raises(ValueError):
        ops.pm_pre_denoise(x, b, w0, torch.empty_like(w0), None, None, 1.0, 0.01, 0.1, mosaic=torch.empty_like(x))
    c8 = torch.empty(B, 2, M, N, 8, device='cuda')
    with pytest.raises(ValueError, match='alias'):            # the mosaic may not be x or b: their neighbours are still being read
        ops.pm_pre_denoise(x, b, w0, None, None, c8, 1.0, 0.01, 0.1, mosaic=x)
    with pytest.raises(ValueError, match='layout'):           # ... and some denoiser-input layout has to be asked for
        ops.pm_pre_denoise(x, b, w0, None, None, None, 1.0, 0.01, 0.1, mosaic=torch.empty_like(x))
    th = torch.empty_like(x)
    with pytest.raises(ValueError, match='alias'):
        ops.pm_post_denoise(out_rgb, None, None, x.clone(), None, th, b.clone(), w0.clone(), False, mosaic=th)


def test_projection_on_a_state_beyond_the_infinity_cache_equals_the_small_state_kernel(ops):
    """round 6: above 384 MB per launch scipnp_pm_project reads theta, b, Phi with non-temporal loads (pm_project_kernel<4, 8, MODE, true>:
    6.0 instead of 4.8 TB/s on a 2048 x 2048 x 8 state, profiles/r06k_proj_stream.txt) -- the same per-pixel expressions: a unit batch of
    16 cubes of 512 x 512 x 8 (570 MB per launch, the large-state kernel) gives bit for bit what 16 single-cube launches (one pixel per
    thread, plain loads) give, in both conventions"""
    g = torch.Generator().manual_seed(61)
    U, B, M, N = 16, 8, 256, 256
    th = torch.rand(B * U, 4, M, N, generator=g).cuda()
    b = (torch.rand(B * U, 4, M, N, generator=g) - 0.5).cuda()
    Phi = (torch.rand(B * U, 4, M, N, generator=g) > 0.5).float().cuda()
    y = (torch.rand(U * 4, M, N, generator=g) * B / 2).cuda()
    Ps, _ = ops.pm_setup_units(Phi, y, U, want_x0=False)
    for mode, c0, c1 in ((0, 1 / 0.55, 0.55), (1, 1.0, 0.01)):
        big = ops.pm_project(th, b, Phi, y, Ps, mode, c0, c1, torch.empty_like(th), units=U)
        fr = lambda t, u: t.view(B, U, 4, M, N)[:, u].contiguous()       # noqa: E731  (frame f = t * U + u)
        for u in (0, 7, 15):
            one = ops.pm_project(fr(th, u), fr(b, u), fr(Phi, u), y.view(U, 4, M, N)[u].contiguous(), Ps.view(U, 4, M, N)[u].contiguous(),
                                 mode, c0, c1, torch.empty(B, 4, M, N, device='cuda'))
            assert torch.equal(fr(big, u), one), (mode, u)


def test_tv_banded_kernel_tiny_operands_take_the_general_path(ops):
    """round 5: the banded kernel writes out sqrt and the two divisions of the dual update itself (csrc/tv.hip tv_p_update_fast:
    v_sqrt + 1-ulp fix-up, one shared reciprocal refinement, no operand scaling) and a wave falls back to the compiler's
    correctly rounded sqrtf and '/' when a gradient energy or a numerator lies in (0, 2^-60) -- amplitudes from 1 down to the
    denormals, alone and mixed inside one wave, flat (all-zero-gradient) regions included: bit-identical to the tiled kernel,
    which only uses sqrtf and '/'"""
    rng = np.random.default_rng(2025)
    M, N = 70, 128
    # (... and on the upper side, round 6: gradient energies above 2^100 -- amplitudes of 1e16 and more --
    # take the general path as well: the hand-expanded division would make NaNs where the IEEE one rounds to zero)
    scales = [1.0, 1e-9, 1e-15, 1e-20, 1e-25, 1e-32, 1e-38, 3e-42, 0.0, 1e12, 1e16, 1e18]
    base = rng.uniform(0, 1, (len(scales) + 3, M, N)).astype(np.float32)
    for c, sc in enumerate(scales):
        base[c] *= np.float32(sc)
    base[-3, :, 40:] *= np.float32(1e-24)             # one wave, both regimes
    base[-2, 20:50, :] = 0.25                         # a flat region inside a normal image
    base[-1, ::2, :] *= np.float32(1e-30)             # alternating rows
    x = dev(base)
    for b, coef in ((None, 0.0), (dev((rng.normal(0, 1, base.shape) * 1e-28).astype(np.float32)), 1.0)):
        for n_iter in (2, 5):
            p1, p3 = ops.TvPlan(M, N, x.shape[0], n_iter, x.device), ops.TvPlan(M, N, x.shape[0], n_iter, x.device)
            o1, o3, o4 = torch.empty_like(x), torch.full_like(x, -5.0), torch.full_like(x, -6.0)
            ops.tv_chambolle(x, b, coef, o1, p1, 0.1, kernel=1)
            ops.tv_chambolle(x, b, coef, o3, p3, 0.1, kernel=3)
            assert torch.equal(p1.stop_iter, p3.stop_iter), (n_iter, p1.stop_iter, p3.stop_iter)
            assert torch.equal(o1, o3), n_iter
            ops.tv_chambolle(x, b, coef, o4, p3, 0.1, kernel=4)
            assert torch.equal(o1, o4) and torch.equal(p1.stop_iter, p3.stop_iter), n_iter

"""-m gpu: DDnet's own online finetune -- the `args.dm_update` branch of the reference's `test_ddnet`
(packages/DDnet/DDnet_test.py:218-296), the last branch of a SURVEY 8(b) plug-in signature -- on the HIP kernels
(adaptivepnp_sci_amd/ddnet_train.py) against values captured FROM THE REFERENCE (tests/golden/ddnet_finetune_32x48x8.npz,
tools/make_golden.py ddnettune; the oracle reproduces them with rel-L2 0.0): the output cube after the two Adam steps, the
losses, the reference's `.grad` after its first backward, the weights' change."""
import types

import numpy as np
import pytest
import torch

from conftest import load_gold, rel_l2

pytestmark = pytest.mark.gpu

FULL = ('weight_tensor_in', 'weight_tensor_in2', 'weight_tensor_out', 'temp1.inc_1.convblock.0.weight', 'temp1.outc.convblock.2.weight',
        'temp2.inc_1.convblock.2.weight', 'temp2.downc0.convblock.0.weight', 'temp2.upc1.convblock.1.weight',
        'temp11.inc_1.convblock.0.weight', 'temp11.downc1.convblock.2.convblock.0.weight', 'temp11.fusion.convblock.0.weight',
        'temp11.fusion.convblock.2.weight', 'temp11.outc.convblock.2.weight')


GOLDS = ('ddnet_finetune_32x48x8', 'ddnet_finetune_64x64x8')


def _run(precision, monkeypatch, hook=None, gold='ddnet_finetune_32x48x8', steps=None):
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', precision)
    from adaptivepnp_sci_amd import ddnet_train
    from adaptivepnp_sci_amd import test_ddnet as ddnet_plugin
    from oracle.nets import cpu_data_parallel, synth_ddnet_weights
    from oracle.sci_ops import one_to_three_channel
    g = load_gold(gold)
    net = cpu_data_parallel(synth_ddnet_weights(0))
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    args = types.SimpleNamespace(dm_update=True, dm_lr=float(g['lr']), dm_update_per_iter=int(g['steps']) if steps is None else steps)
    ddnet_train.GRAD_HOOK = hook
    try:
        out, model = ddnet_plugin(one_to_three_channel(torch.from_numpy(g['mosaic'])).cuda(), None, None, net, True, args)
    finally:
        ddnet_train.GRAD_HOOK = None
    assert model is net
    return g, out, net, sd0


@pytest.mark.parametrize('gold', GOLDS)
@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_ddnet_online_finetune_matches_reference(precision, gold, monkeypatch, capsys):
    grads = {}
    g, out, net, sd0 = _run(precision, monkeypatch, hook=grads.update, gold=gold)
    # the demosaicked cube after the update (the final pass runs in the engine's precision)
    assert rel_l2(out.cpu().numpy(), g['out']) <= 1e-5
    # losses, as the reference prints them
    printed = [float(l.split(':')[1]) for l in capsys.readouterr().out.splitlines() if l.startswith('ddn loss:')]
    assert len(printed) == 2 and np.allclose(printed, g['losses'], rtol=1e-5)
    # the reference's .grad after its first backward(): every tensor's norm, and the tensors kept in full element by element
    worst = 0.0
    for k, gv in grads.items():
        key = k.replace('.', '_')
        want = float(g['gradnorm_' + key])
        got = float(torch.linalg.vector_norm(gv.double()))
        assert abs(got / want - 1) <= 1e-4, (k, got, want)
        if k in FULL:
            # element by element.  The ReLU behind a DenBlock's FIRST convolution masks dZ_0, so an input there within round-off
            # of zero -- decided by the last bit, as for FastDVDnet (tests/test_gpu_solver.py, ..._under_its_own_relu_masks) --
            # reaches that convolution's weight gradient and the input gates and nothing else: those tensors carry the 1e-3 /
            # 1e-4 a flip is worth (measured 2.8e-5 .. 1.6e-4 and 3.9e-6 on the 64 x 64 x 8 problem, 2e-7 on 32 x 48 x 8, in BOTH
            # precisions: the trainer's backward is fp32 either way); every other tensor agrees to arithmetic accuracy, 1e-5
            # (measured <= 4.1e-7)
            err = rel_l2(gv.cpu().numpy(), g['grad_' + key])
            worst = max(worst, err)
            gate = 1e-3 if k.endswith('inc_1.convblock.0.weight') else 1e-4 if k.startswith('weight_tensor_in') else 1e-5
            assert err <= gate, (k, err, gate)
    assert len(grads) == 53 and worst > 0                      # 16 + 16 + 18 conv weights, three gate tensors
    # the unused `inc` blocks were not touched (no gradient -> Adam skips them); everything else moved by what the reference's
    # fresh-Adam steps moved it (|update| = lr per step wherever |g| >> 1e-8)
    sd = net.state_dict()
    for k, v in sd.items():
        kk = k.replace('module.', '', 1)
        if '.inc.' in kk:
            assert torch.equal(v, sd0[k]), k
            continue
        d = (v.float() - sd0[k].float())
        want = float(g['dnorm_' + kk.replace('.', '_')])
        assert abs(float(torch.linalg.vector_norm(d.double())) / want - 1) <= 2e-2, (k, want)
        if kk in FULL:
            ref = g['delta_' + kk.replace('.', '_')]
            # element by element where the reference's first gradient is not within rounding of zero.  A fresh Adam moves an element by
            # lr * g / (|g| + 1e-8): where the SECOND step's gradient of an element is tiny its update depends on the last bits, so the
            # bulk is gated tightly (99 % within 5 % of the total step) and the tail by a quarter of it
            gref = np.abs(g['grad_' + kk.replace('.', '_')])
            sel = gref > 1e-3 * gref.max()
            dev_ = np.abs(d.numpy() - ref)[sel]
            total = float(g['lr']) * int(g['steps'])
            assert np.quantile(dev_, 0.99) <= 0.05 * total + 1e-12 and dev_.max() <= 0.25 * total, (k, float(dev_.max()))


@pytest.mark.parametrize('gold', GOLDS)
@pytest.mark.parametrize('precision', ['f32', 'f16x3'])
def test_ddnet_first_adam_step_is_the_references_element_by_element(precision, gold, monkeypatch):
    """The FIRST step of the reference's fresh Adam moves every element by -lr g / (|g| + 1e-8) = -lr sign(g) (1 - 1e-8 / |g|):
    wherever the reference's gradient is not on Adam's eps floor (|g| >= 1e-6) the update is fixed to the rounding of the weight
    itself whatever the last bits of g are -- an exact, element-by-element check of the HIP Adam against `delta1_*` captured
    from the reference (tools/make_golden.py ddnettune), EVERY such element, no quantiles."""
    g, _out, net, sd0 = _run(precision, monkeypatch, gold=gold, steps=1)
    lr = float(g['lr'])
    n_cmp = n_floor = 0
    for k, v in net.state_dict().items():
        kk = k.replace('module.', '', 1)
        if kk not in FULL:
            continue
        key = kk.replace('.', '_')
        d = (v.float() - sd0[k].float()).numpy()
        ref, gref = g['delta1_' + key], g['grad_' + key]
        live = np.abs(gref) >= 1e-6
        n_cmp, n_floor = n_cmp + int(live.sum()), n_floor + int((~live).sum())
        ulp = np.spacing(np.abs(sd0[k].float().numpy()).astype(np.float32))
        assert np.all(np.sign(d[live]) == -np.sign(gref[live])), k
        assert np.all(np.abs(np.abs(d[live]) - lr) <= 0.011 * lr + ulp[live]), (k, float(np.abs(np.abs(d[live]) - lr).max()))
        assert np.all(np.abs(d - ref)[live] <= 1e-3 * lr + 2 * ulp[live]), (k, float(np.abs(d - ref)[live].max()))
        assert np.abs(d).max() <= 1.001 * lr + ulp.max()                 # nobody moves further than one Adam step can
    assert n_cmp > 2000 and n_floor < n_cmp, (n_cmp, n_floor)


def test_ddnet_finetune_with_zero_steps_is_the_plain_pass(monkeypatch):
    """dm_update with dm_update_per_iter = 0: no step, the weights stay, the output is the plain pass"""
    from adaptivepnp_sci_amd import test_ddnet as ddnet_plugin
    from oracle.nets import cpu_data_parallel, synth_ddnet_weights
    from oracle.sci_ops import one_to_three_channel
    g = load_gold('ddnet_finetune_32x48x8')
    net = cpu_data_parallel(synth_ddnet_weights(0))
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x = one_to_three_channel(torch.from_numpy(g['mosaic'])).cuda()
    plain = ddnet_plugin(x, None, None, net)
    out, _m = ddnet_plugin(x, None, None, net, True, types.SimpleNamespace(dm_update=True, dm_lr=1e-5, dm_update_per_iter=0))
    assert torch.equal(out, plain)
    assert all(torch.equal(v, sd0[k]) for k, v in net.state_dict().items())


def test_ddnet_glue_adjoints_vs_autograd():
    """the adjoint kernels of csrc/ddnet.hip against PyTorch autograd of the forward glue in float64: bilinear x2
    (align_corners), the gather with its gate gradients and centre path, the mix"""
    import ctypes as C
    from adaptivepnp_sci_amd import _lib, ops
    lib = _lib.load()
    st = _lib.stream_ptr
    P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())          # noqa: E731
    gen = torch.Generator().manual_seed(5)
    # ---- bilinear
    E, h, w = 3, 6, 10
    x = torch.randn(E, 4, h, w, generator=gen, dtype=torch.float64, requires_grad=True)
    up = torch.nn.functional.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    gup = torch.randn(up.shape, generator=gen, dtype=torch.float64)
    up.backward(gup)
    d_up = torch.zeros(E, 1, 2 * h, 2 * w, 8)
    d_up[:, 0, :, :, :4] = gup.permute(0, 2, 3, 1).float()
    d_in = torch.empty(E, 4, h, w, device='cuda')
    d_up_d = d_up.cuda()                                  # (device copies are kept in variables: a temporary would be freed -- and its
    _lib.check(lib.scipnp_bilinear_up2_bwd_c8(P(d_up_d), P(d_in), E, h, w, st()), 'bilinear bwd')        # memory reused -- before the launch)
    assert rel_l2(d_in.cpu().numpy(), x.grad.numpy()) <= 1e-6
    # ---- gather (+ centre path) for C = 1 (centre gradient summed over 3 output channels) and C = 4
    for Cc, Cd in ((1, 3), (4, 4), (3, 3)):
        Bn, hh, ww = 4, 8, 12
        E = 3 * Bn
        HW = hh * ww
        src = torch.randn(Bn, Cc, hh, ww, generator=gen)
        n_ = torch.arange(Bn)
        idx = torch.stack([torch.stack([(n_ - 2 + j + i) % Bn for i in range(3)], 1) for j in range(3)]).reshape(E, 3).int()
        gate = (1 + 0.1 * torch.randn(3, 3, Cc, generator=gen)).double().requires_grad_(True)
        scale = gate[:, None].expand(3, Bn, 3, Cc).reshape(E, 3, Cc)
        t_in = torch.stack([torch.cat([src[idx[e, i]].double() * scale[e, i][:, None, None] for i in range(3)]) for e in range(E)])
        centre = torch.stack([src[idx[e, 1]].double() * scale[e, 1][:, None, None] for e in range(E)])       # [E][C][h][w]
        g_t = torch.randn(t_in.shape, generator=gen, dtype=torch.float64)
        g_c = torch.randn(E, Cd, hh, ww, generator=gen, dtype=torch.float64)
        centre_out = centre.expand(E, Cd, hh, ww) if Cc == 1 else centre
        ((t_in * g_t).sum() + (centre_out * g_c).sum()).backward()
        K, G = 3 * Cc, (3 * Cc + 7) // 8
        d_tin = torch.zeros(E, G * 8, hh, ww)
        d_tin[:, :K] = g_t.float()
        d_tin_c8 = d_tin.view(E, G, 8, hh, ww).permute(0, 1, 3, 4, 2).contiguous().cuda()
        ncols = C.c_int(0)
        _lib.check(lib.scipnp_ddnet_gather_bwd(None, None, 0, None, None, None, None, None, E, Bn, Cc, hh, ww, C.byref(ncols), None), 'size')
        part = torch.empty(9 * Cc, ncols.value, dtype=torch.float64, device='cuda')
        g_c_d, src_d, idx_d = g_c.float().cuda(), src.cuda(), idx.cuda()
        _lib.check(lib.scipnp_ddnet_gather_bwd(P(d_tin_c8), P(g_c_d), Cd, P(src_d), P(idx_d), None, None, P(part), E,
                                               Bn, Cc, hh, ww, C.byref(ncols), st()), 'gather bwd')
        got = ops.sum_rows_f64(part).cpu().numpy().reshape(3, 3, Cc)
        assert rel_l2(got, gate.grad.numpy()) <= 1e-6, (Cc, rel_l2(got, gate.grad.numpy()))
    # ---- mix
    B, H, W = 3, 8, 10
    s2 = torch.randn(2 * B, 3, H, W, generator=gen)
    a3 = (0.5 + 0.1 * torch.randn(2, 3, generator=gen)).double().requires_grad_(True)
    s2d = s2.double().requires_grad_(True)
    outm = a3[0][None, :, None, None] * s2d[:B] + a3[1][None, :, None, None] * s2d[B:]
    gout = torch.randn(outm.shape, generator=gen, dtype=torch.float64)
    outm.backward(gout)
    ncols = C.c_int(0)
    _lib.check(lib.scipnp_ddnet_mix_bwd(None, None, None, None, None, B, H, W, C.byref(ncols), None), 'size')
    part = torch.empty(6, ncols.value, dtype=torch.float64, device='cuda')
    d_s2 = torch.empty(2 * B, 3, H, W, device='cuda')
    gout_d, s2_d, a3_d = gout.float().cuda(), s2.cuda(), a3.detach().float().cuda()
    _lib.check(lib.scipnp_ddnet_mix_bwd(P(gout_d), P(s2_d), P(a3_d), P(d_s2), P(part), B, H, W,
                                        C.byref(ncols), st()), 'mix bwd')
    assert rel_l2(d_s2.cpu().numpy(), s2d.grad.numpy()) <= 1e-6
    assert rel_l2(ops.sum_rows_f64(part).cpu().numpy().reshape(2, 3), a3.grad.numpy()) <= 1e-6

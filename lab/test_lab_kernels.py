"""-m gpu, NOT part of the product's suite (`pytest tests/` does not collect this directory): the kernels of lab/csrc -- built,
measured on MI355X and not adopted -- against the product kernels they re-arrange, bit for bit.
    make -C lab && python -m pytest lab/test_lab_kernels.py -q -m gpu"""
import os
import sys

import numpy as np
import pytest
import torch

LAB = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(LAB)
sys.path.insert(0, LAB)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import rel_l2  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from adaptivepnp_sci_amd import ops as O
    return O


def test_conv3x3_winograd_persistent_equals_classic_kernel(ops, monkeypatch):
    """lab/csrc/conv_winop.hip (LABORATORY, lab/libscipnp_lab.so) -- the persistent 96-output-channel form (input transform shared
    through LDS, 12-wave resident workgroups, channel-group pipeline running across unit boundaries) -- computes the same
    products in the same order as the product's csrc/conv_wino.hip: BIT-IDENTICAL outputs for every epilogue, ragged sizes
    (units cut by the right / bottom border), one and many channel groups, fewer units than CUs and many units per
    workgroup, several frames.  (Measured 10 % slower in round 3 and not adopted; kept as the measured alternative.)"""
    import lablib as diaglib
    monkeypatch.setenv('SCIPNP_WINO_F4', '0')
    g = torch.Generator().manual_seed(96)
    shapes = [(1, 96, 4, 32), (2, 96, 20, 36), (3, 16, 13, 70), (1, 8, 1, 1), (2, 48, 37, 97), (8, 96, 128, 128),
              (5, 96, 66, 250), (1, 24, 300, 33)]
    for n, cin, h, w in shapes:
        cout = 96
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        bias = torch.randn(cout, generator=g)
        packed = ops.pack_conv3x3(wt, bias, Cin=cin, Cout=cout, device='cuda')
        both = ops.pack_conv3x3_wino_both(packed, cin, cout)
        pwinop = diaglib.pack_winop(packed, cin, cout)
        assert pwinop is not None and both.f4 is None
        xc = ops.to_c8(x.cuda())
        res = ops.to_c8(torch.randn(n, cout, h, w, generator=g).cuda())
        fwd = ops.to_c8(torch.randn(n, cout, h, w, generator=g).cuda())
        for kw in (dict(), dict(relu=True), dict(relu=True, residual=res), dict(mask_src=fwd), dict(mask_src=fwd, residual=res),
                   dict(relu=True, head=True)):
            got = diaglib.conv3x3_c8p(xc, pwinop, cout, **kw)
            want = ops.conv3x3_c8w(xc, both, cout, **kw)
            assert torch.equal(got, want), (n, cin, h, w, sorted(kw), float((got - want).abs().max()))
        if h * w <= 4096:
            ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1)
            err = rel_l2(ops.from_c8(diaglib.conv3x3_c8p(xc, pwinop, cout)).cpu().numpy(), ref.numpy())
            assert err < 2e-6, (n, cin, h, w, err)
    # shapes without a persistent form
    assert diaglib.pack_winop(ops.pack_conv3x3(torch.zeros(64, 64, 3, 3), None, Cin=64, Cout=64, device='cuda'), 64, 64) is None


def test_conv3x3_winograd_f4_three_waves_per_simd_equals_the_product_kernel(ops):
    """Both three-waves-per-SIMD prototypes of round 5 (profiles/r05a_*, r05e_*: measured slower, not adopted).
    lab/csrc/conv_wino4x.hip (LABORATORY, lab/libscipnp_lab.so; round 5) -- the F(4x4,3x3) convolution with a tile's 36 positions split
    over THREE waves (12-wave workgroups of 16 x 64 pixels, 164 VGPRs, three waves per SIMD) on the SAME packed weights:
    BIT-IDENTICAL to the product's lab/csrc/conv_wino4.hip (whose whole-line store epilogue is thereby checked against the classic
    per-lane epilogue this kernel still has) for every plain-store epilogue, ragged sizes, narrow outputs, several frames.
    (Measured 277 us against 249 us on the FFDNet body layer and not adopted: profiles/r05a_*.)"""
    import lablib as diaglib
    g = torch.Generator().manual_seed(5)
    for (n, cin, cout, h, w) in ((1, 16, 32, 8, 64), (2, 24, 40, 13, 70), (1, 96, 96, 37, 131), (3, 32, 16, 20, 64), (1, 8, 96, 9, 9),
                                 (2, 64, 64, 64, 64), (8, 96, 96, 128, 128)):
        x = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
        pk = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) * 0.1, torch.randn(cout, generator=g), Cin=cin, Cout=cout,
                              device='cuda')
        p4 = ops.pack_conv3x3_wino4(pk, cin, cout)
        res = ops.to_c8(torch.randn(n, cout, h, w, generator=g).cuda())
        msk = ops.to_c8(torch.randn(n, cout, h, w, generator=g).cuda())
        for kw in ({}, {'relu': True}, {'relu': True, 'residual': res, 'head': True}, {'mask_src': msk, 'residual': res}):
            want = ops.conv3x3_c8w4(x, p4, cout, **kw)
            assert torch.equal(want, diaglib.conv3x3_c8w6(x, p4, cout, **kw)), ((n, cin, cout, h, w), sorted(kw))
            # and the 16-channel-workgroup form (lab/csrc/conv_wino4n.hip: three independent workgroups per CU, re-laid weights)
            kwn = {k: v for k, v in kw.items() if k != 'head'}
            assert torch.equal(want, diaglib.conv3x3_c8wn(x, diaglib.repack_wino4n(p4, cin, cout), cout, **kwn)), ((n, cin, cout, h, w), sorted(kw))


def test_conv3x3_winograd_f4_producer_consumer_kernel_equals_the_product_kernel(ops):
    """lab/csrc/conv_wino4p.hip (LABORATORY; round 5): 12-wave workgroups of 8 consumer waves (MFMAs only, operands from LDS) and 4
    producer waves (raw-tile requests, the input transform ONCE for 64 output channels) -- BIT-IDENTICAL to the product's
    scipnp_conv3x3_c8w4 for Cout % 64 == 0 on ragged shapes, one to sixteen channel groups, every plain-store epilogue.
    (Measured at parity inside a network pass, profiles/r05f_*: kept as the first step of DESIGN.md section 9.12.)"""
    import lablib as diaglib
    g = torch.Generator().manual_seed(7)
    for (n, cin, cout, h, w) in ((1, 8, 64, 8, 64), (2, 16, 64, 13, 70), (1, 64, 64, 37, 131), (3, 32, 128, 20, 64), (1, 128, 128, 9, 9),
                                 (1, 24, 192, 16, 128), (8, 64, 64, 128, 128)):
        x = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
        pk = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) * 0.1, torch.randn(cout, generator=g), Cin=cin, Cout=cout,
                              device='cuda')
        p4 = ops.pack_conv3x3_wino4(pk, cin, cout)
        res = ops.to_c8(torch.randn(n, cout, h, w, generator=g).cuda())
        msk = ops.to_c8(torch.randn(n, cout, h, w, generator=g).cuda())
        for kw in ({}, {'relu': True}, {'relu': True, 'residual': res}, {'mask_src': msk, 'residual': res}):
            assert torch.equal(ops.conv3x3_c8w4(x, p4, cout, **kw), diaglib.conv3x3_c8wp(x, p4, cout, **kw)), ((n, cin, cout, h, w), sorted(kw))


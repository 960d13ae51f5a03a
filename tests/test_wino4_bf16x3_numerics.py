"""CPU: the numerics of the F(4x4,3x3) Winograd convolution on SPLIT bf16 operands (both operands of the position
GEMMs as three bf16 planes, six partial products, fp32 accumulation -- the arithmetic a bf16-MFMA form of csrc/conv_wino4.hip would run).
What is pinned here is the QUALIFICATION the verdict set: through the 12 layers of the colour FFDNet (reference network_ffdnet.py:46-69,
committed weights) the error against the float64 direct convolution must not exceed the fp32 F(4x4) path's -- and the properties of the
split itself.  The kernel was NOT built: tools/probes/wino4b_loop_probe.hip measured its inner loop on MI355X at 0.91x of the fp32
loop's clocks at best (profiles/r06c_wino4b_loop_probe.txt, DESIGN.md section 5 "Round 6"); this file keeps the arithmetic side of
that decision reproducible."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools', 'probes'))


def _bf16_rn(a):
    """float32 array -> nearest bf16 (ties to even), as float32 -- what v_cvt_pk_bf16_f32 does"""
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def test_three_bf16_planes_hold_every_float32_exactly():
    """x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid: both remainders are exact float32 differences and
    lo is itself a bf16 value (8 + 8 + 8 significant bits, round-to-nearest remainders are signed) -- for every binade whose third
    plane is still a normal number (|x| >= 2^-100 here; below ~2^-110 the low plane falls into bf16's subnormals and loses bits: a
    magnitude no activation or transformed weight of the networks has)"""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp2(rng.integers(-100, 100, 200000))).astype(np.float32)
    x = np.concatenate([x, np.float32([0, -0.0, 1, 1 + 2.0 ** -8, 1 + 2.0 ** -9, 1 - 2.0 ** -9, 65280, 3.0e-38, np.finfo(np.float32).max / 2])])
    hi = _bf16_rn(x)
    r1 = x - hi
    mid = _bf16_rn(r1)
    lo = r1 - mid
    assert np.array_equal(x.astype(np.float64), hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64))
    big = np.abs(x) >= 2.0 ** -100
    assert np.array_equal(_bf16_rn(lo[big]), lo[big])                        # the third plane needs no rounding
    assert np.all(np.abs(r1[big]) <= np.abs(x[big]) * 2.0 ** -8) and np.all(np.abs(lo[big]) <= np.abs(x[big]) * 2.0 ** -16)


def test_six_products_on_round_to_nearest_planes_qualify_against_fp32_f4x4():
    """12 FFDNet layers, 128 x 128 noisy image: relative L2 error against the float64 direct convolution of
      fp32 F(4x4,3x3)                                  (the product kernel's arithmetic, 3.0e-7)
      bf16x3 F(4x4,3x3), products hh hm mh hl lh mm    round-to-nearest split: NOT above the fp32 path's -> qualifies
      the same on planes cut by TRUNCATION             biased remainders: above it even with eight products -> does not
    (round 6 measured 3.04e-7 | 2.74e-7 | 5.35e-7 (six) and 3.72e-7 (eight), profiles/r06_wino_bf16x3_numerics_sim.txt)"""
    import wino_bf16x3_numerics_sim as sim
    from wino_numerics_sim_lib import cook_toom, ffdnet, load_case, wino_conv
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    W, x64, ref, rel = load_case()
    x32 = x64.float()
    mats = cook_toom([0, 1, -1, 2, -2], 4, 3)
    e_f32 = rel(ffdnet(x32, 25 / 255, W, lambda x, w, b: wino_conv(x, w, b, mats, torch.float32)))
    e_rn6 = rel(ffdnet(x32, 25 / 255, W, lambda x, w, b: sim.wino_conv_bf16x3(x, w, b, mats, 6, True)))
    e_tr6 = rel(ffdnet(x32, 25 / 255, W, lambda x, w, b: sim.wino_conv_bf16x3(x, w, b, mats, 6, False)))
    e_tr8 = rel(ffdnet(x32, 25 / 255, W, lambda x, w, b: sim.wino_conv_bf16x3(x, w, b, mats, 8, False)))
    assert 1e-7 < e_f32 < 4e-7, e_f32
    assert e_rn6 <= e_f32, (e_rn6, e_f32)                     # the verdict's bar: not narrower than the reference's fp32
    assert e_tr6 > e_f32 and e_tr8 > e_f32, (e_tr6, e_tr8, e_f32)
    # three products (two planes, 16 bits per operand) are two orders of magnitude away: the third plane is what buys fp32
    e_rn3 = rel(ffdnet(x32, 25 / 255, W, lambda x, w, b: sim.wino_conv_bf16x3(x, w, b, mats, 3, True)))
    assert e_rn3 > 20 * e_f32, e_rn3

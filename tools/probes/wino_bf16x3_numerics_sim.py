#!/usr/bin/env python3
"""CPU (PyTorch, a minute or two): rounding error of Winograd F(4x4,3x3) with BOTH operands of the position GEMMs carried as
three bf16 planes (hi, mid, lo: 8 + 8 + 8 significant bits = every fp32 value EXACTLY, see split3), products accumulated in fp32,
through the 12 layers of the colour FFDNet against the float64 direct convolution -- the qualification of csrc/conv_wino4b.hip
("not narrower than the reference's fp32": the error must not exceed the fp32 F(4x4) path's 3.0e-7).

Product sets (U plane x V plane), by the size of what they drop relative to |U||V|:
  9: all                                            -> exact products, fp32 accumulation only
  8: all but lo*lo (2^-32)
  7: 8 minus Ulo*Vmid  (what the kernel issues: two MFMAs of four K-slots each, one slot zero)
  6: hh hm mh hl lh mm (drops the two 2^-24 cross terms)
  3: hh hm mh  (16 bits: for the scale of the error only)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wino_numerics_sim_lib import cook_toom, ffdnet, load_case   # noqa: E402

torch.set_num_threads(8)


def trunc_bf16(t):
    """fp32 -> the fp32 value of its upper 16 bits (bf16 by truncation)"""
    return (t.contiguous().view(torch.int32) & -65536).view(torch.float32)


def split3(t, rn=False):
    """t fp32 -> (hi, mid, lo) fp32 tensors, each exactly a bf16 value, hi + mid + lo == t exactly (both roundings)"""
    t = t.float()
    if rn:
        hi = t.bfloat16().float(); r = t - hi
        mid = r.bfloat16().float(); lo = r - mid
    else:
        hi = trunc_bf16(t); r = t - hi
        mid = trunc_bf16(r); lo = r - mid
    assert torch.equal(lo.bfloat16().float(), lo), 'lo is not a bf16 value'
    assert torch.equal(hi + mid + lo, t)
    return hi, mid, lo


SETS = {9: ['hh', 'hm', 'mh', 'hl', 'lh', 'mm', 'ml', 'lm', 'll'],
        8: ['hh', 'hm', 'mh', 'hl', 'lh', 'mm', 'ml', 'lm'],
        7: ['hh', 'hm', 'mh', 'hl', 'lh', 'mm', 'ml'],          # U plane first: 'ml' = Umid * Vlo; dropped: Ulo * Vmid
        6: ['hh', 'hm', 'mh', 'hl', 'lh', 'mm'],
        3: ['hh', 'hm', 'mh']}


def wino_conv_bf16x3(x, w, b, mats, nprod, rn):
    AT, G, BT = [torch.from_numpy(a) for a in mats]
    m, n = AT.shape
    N, C, H, W = x.shape
    Th, Tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, Tw * m + 1 - W, 1, Th * m + 1 - H))
    t = xp.unfold(2, n, m).unfold(3, n, m)
    U = torch.einsum('ij,ocjk,lk->iloc', G, w.double(), G).float()        # packed in fp64, rounded to fp32 once (as the fp32 kernel)
    Up = dict(zip('hml', split3(U, rn)))
    BTd, ATd = BT.float(), AT.float()
    Vt = torch.einsum('ij,ncthjk,lk->ilncth', BTd, t.float(), BTd)        # fp32 transform
    Vp = dict(zip('hml', split3(Vt, rn)))
    M = None
    for pr in SETS[nprod]:                                                # fp32 accumulation of the partial GEMMs
        term = torch.einsum('iloc,ilncth->ilnoth', Up[pr[0]], Vp[pr[1]])
        M = term if M is None else M + term
    Y = torch.einsum('ai,ilnoth,bl->nothab', ATd, M, ATd)
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, -1, Th * m, Tw * m)[:, :, :H, :W]
    return Y + b.float().view(1, -1, 1, 1)


def main():
    W, x64, ref, rel = load_case()
    x32 = x64.float()
    mats = cook_toom([0, 1, -1, 2, -2], 4, 3)
    from wino_numerics_sim_lib import wino_conv
    print('fp32 direct                 ', rel(ffdnet(x32, 25 / 255, W, lambda x, w, b: F.conv2d(x, w.float(), b.float(), padding=1))))
    print('fp32 F(4x4,3x3)             ', rel(ffdnet(x32, 25 / 255, W, lambda x, w, b: wino_conv(x, w, b, mats, torch.float32))))
    for rn in (False, True):
        for nprod in (9, 8, 7, 6, 3):
            e = rel(ffdnet(x32, 25 / 255, W, lambda x, w, b: wino_conv_bf16x3(x, w, b, mats, nprod, rn)))
            print(f'bf16x3 F(4x4,3x3) {nprod} products, split by {"round-to-nearest" if rn else "truncation      "}: {e:.3e}')


if __name__ == '__main__':
    main()

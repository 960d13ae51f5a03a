#!/usr/bin/env python3
"""CPU (PyTorch, a minute): rounding error of Winograd convolution forms through the 12 layers of the colour FFDNet (committed
weights, a noisy 128 x 128 test image), against the float64 direct convolution -- the numbers quoted in csrc/conv_wino4.hip and
DESIGN.md section 5 / 9:
  fp32 direct | fp32 F(2x2,3x3) | fp32 F(4x4,3x3) with several point sets | fp32 F(3x3,3x3)
  and the same transforms on SPLIT-fp16 operands (U and V carried as (hi, lo) fp16 pairs, three of the four partial products,
  fp32 accumulation; weights pre-scaled by 2^11 like the library's split kernel) -- the candidate for the library's default path.
The Winograd matrices are generated exactly (Cook-Toom over the given points + infinity, sympy rationals)."""
import os
import sys

import numpy as np
import sympy
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(8)

def cook_toom(points, m, r):
    n = m + r - 1
    pts = [sympy.Rational(p) for p in points]; assert len(pts) == n - 1
    V = sympy.zeros(n, n)
    for j, p in enumerate(pts):
        for k in range(n): V[j, k] = p ** k
    V[n - 1, n - 1] = 1
    D = sympy.eye(n)
    for j, p in enumerate(pts):
        D[j, j] = sympy.prod([p - q for k, q in enumerate(pts) if k != j])
    Vinv = V.inv()
    BT = D * Vinv.T
    Vr = V[:, :r].copy(); Vr[n - 1, r - 1] = 1
    Vm = V[:, :m].copy(); Vm[n - 1, m - 1] = 1
    G = D.inv() * Vr
    AT = Vm.T
    f = lambda M: np.array(M.tolist(), dtype=np.float64)
    return f(AT), f(G), f(BT)

def wino_conv(x, w, b, mats, dt=torch.float32):
    """x (N,C,H,W) dt; w (O,C,3,3) float64; stride 1 pad 1"""
    AT, G, BT = [torch.from_numpy(a) for a in mats]
    m, n = AT.shape
    N, C, H, W = x.shape
    Th, Tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, Tw * m + 1 - W, 1, Th * m + 1 - H))
    t = xp.unfold(2, n, m).unfold(3, n, m)           # N,C,Th,Tw,n,n
    U = torch.einsum('ij,ocjk,lk->iloc', G, w.double(), G).to(dt)     # n,n,O,C  (packed in fp64, rounded)
    BTd = BT.to(dt); ATd = AT.to(dt)
    Vt = torch.einsum('ij,ncthjk,lk->ilncth', BTd, t, BTd)
    M = torch.einsum('iloc,ilncth->ilnoth', U, Vt)
    Y = torch.einsum('ai,ilnoth,bl->nothab', ATd, M, ATd)     # N,O,Th,Tw,m,m
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, -1, Th * m, Tw * m)[:, :, :H, :W]
    return Y + b.to(dt).view(1, -1, 1, 1)

def ffdnet(x, sigma, W, conv):
    n, c, h, w = x.shape
    hh, ww = h // 2, w // 2
    x = x.reshape(n, c, hh, 2, ww, 2).permute(0, 1, 3, 5, 2, 4).reshape(n, c * 4, hh, ww)
    x = torch.cat((x, torch.full((n, 1, hh, ww), sigma, dtype=x.dtype)), 1)
    for i in range(12):
        x = conv(x, W[f'model.{2*i}.weight'], W[f'model.{2*i}.bias'])
        if i < 11: x = torch.relu(x)
    return F.pixel_shuffle(x, 2)

d = np.load(os.path.join(ROOT, 'tests', 'golden', 'ffdnet_color_weights.npz'))
W = {k: torch.from_numpy(d[k]) for k in d.keys()}
rng = np.random.default_rng(0)
H = 128
yy, xx = np.mgrid[0:H, 0:H] / H
img = np.stack([0.5 + 0.3 * np.sin(7 * xx + 3 * yy + c) * np.cos(5 * yy - c) for c in range(3)])[None]
img = np.clip(img + (25 / 255) * rng.standard_normal(img.shape), 0, 1)
x64 = torch.from_numpy(img)
direct = lambda dt: (lambda x, w, b: F.conv2d(x, w.to(dt), b.to(dt), padding=1))
ref = ffdnet(x64, 25 / 255, W, direct(torch.float64))
x32 = x64.float()
def rel(a): return float((a.double() - ref).norm() / ref.norm())
print('fp32 direct         ', rel(ffdnet(x32, 25 / 255, W, direct(torch.float32))))
for name, pts, m in (('F(2,3) 0,1,-1', [0, 1, -1], 2),
                     ('F(4,3) 0,1,-1,2,-2', [0, 1, -1, 2, -2], 4),
                     ('F(4,3) 0,1,-1,1/2,-1/2', [0, 1, -1, sympy.Rational(1, 2), -sympy.Rational(1, 2)], 4),
                     ('F(4,3) 0,1,-1,1/2,-2', [0, 1, -1, sympy.Rational(1, 2), -2], 4),
                     ('F(4,3) 0,1,-1,2,-1/2', [0, 1, -1, 2, -sympy.Rational(1, 2)], 4),
                     ('F(3,3) 0,1,-1,2', [0, 1, -1, 2], 3),
                     ('F(3,3) 0,1,-1,1/2', [0, 1, -1, sympy.Rational(1,2)], 3),
                     ):
    mats = cook_toom(pts, m, 3)
    # sanity in fp64
    e64 = float((ffdnet(x64, 25 / 255, W, lambda x, w, b: wino_conv(x, w, b, mats, torch.float64)) - ref).norm() / ref.norm())
    e32 = rel(ffdnet(x32, 25 / 255, W, lambda x, w, b: wino_conv(x, w, b, mats, torch.float32)))
    print(f'{name:28s} fp64 {e64:.2e}  fp32 {e32:.3e}')

# ---- split-fp16 operands in the Winograd domain: U, V carried as (hi, lo) fp16 pairs, 3 of the 4 partial products, fp32 accumulation
def split16(t, scale=1.0):
    ts = (t * scale).float()
    hi = ts.half()
    lo = (ts - hi.float()).half()
    return hi.float(), lo.float()

def wino_conv_split(x, w, b, mats, wscale=2.0 ** 11):
    AT, G, BT = [torch.from_numpy(a) for a in mats]
    m, n = AT.shape
    N, C, H, W = x.shape
    Th, Tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, Tw * m + 1 - W, 1, Th * m + 1 - H))
    t = xp.unfold(2, n, m).unfold(3, n, m)
    U = torch.einsum('ij,ocjk,lk->iloc', G, w.double(), G)                 # fp64
    Uh, Ul = split16(U, wscale)
    BTd, ATd = BT.float(), AT.float()
    Vt = torch.einsum('ij,ncthjk,lk->ilncth', BTd, t.float(), BTd)        # fp32 transform
    Vh, Vl = split16(Vt)
    M = (torch.einsum('iloc,ilncth->ilnoth', Uh, Vh) + torch.einsum('iloc,ilncth->ilnoth', Uh, Vl) +
         torch.einsum('iloc,ilncth->ilnoth', Ul, Vh)) / wscale
    Y = torch.einsum('ai,ilnoth,bl->nothab', ATd, M, ATd)
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, -1, Th * m, Tw * m)[:, :, :H, :W]
    return Y + b.float().view(1, -1, 1, 1)

for name, pts, m in (('F(2,3) split', [0, 1, -1], 2), ('F(4,3) split', [0, 1, -1, 2, -2], 4)):
    mats = cook_toom(pts, m, 3)
    for ws in (1.0, 2.0 ** 11, 2.0 ** 14):
        e = rel(ffdnet(x32, 25 / 255, W, lambda x, w, b: wino_conv_split(x, w, b, mats, ws)))
        print(f'{name:16s} weight scale 2^{int(np.log2(ws)):2d}: {e:.3e}')
# direct split-fp16 (the library default's arithmetic) for reference
def direct_split(x, w, b):
    xh, xl = split16(x); wh, wl = split16(w, 2.0 ** 11)
    y = (F.conv2d(xh, wh, None, padding=1) + F.conv2d(xh, wl, None, padding=1) + F.conv2d(xl, wh, None, padding=1)) / 2.0 ** 11
    return y + b.float().view(1, -1, 1, 1)
print('direct split-fp16           ', rel(ffdnet(x32, 25 / 255, W, direct_split)))

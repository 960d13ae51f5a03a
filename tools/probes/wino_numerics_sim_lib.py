"""Shared pieces of the Winograd numerics simulations (CPU): exact Cook-Toom matrices, the FFDNet graph on a conv callback, the
test case (committed FFDNet weights, a noisy 128 x 128 image, the float64 direct-convolution reference)."""
import os

import numpy as np
import sympy
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cook_toom(points, m, r):
    n = m + r - 1
    pts = [sympy.Rational(p) for p in points]
    assert len(pts) == n - 1
    V = sympy.zeros(n, n)
    for j, p in enumerate(pts):
        for k in range(n):
            V[j, k] = p ** k
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
    t = xp.unfold(2, n, m).unfold(3, n, m)
    U = torch.einsum('ij,ocjk,lk->iloc', G, w.double(), G).to(dt)
    BTd = BT.to(dt); ATd = AT.to(dt)
    Vt = torch.einsum('ij,ncthjk,lk->ilncth', BTd, t, BTd)
    M = torch.einsum('iloc,ilncth->ilnoth', U, Vt)
    Y = torch.einsum('ai,ilnoth,bl->nothab', ATd, M, ATd)
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, -1, Th * m, Tw * m)[:, :, :H, :W]
    return Y + b.to(dt).view(1, -1, 1, 1)


def ffdnet(x, sigma, W, conv):
    n, c, h, w = x.shape
    hh, ww = h // 2, w // 2
    x = x.reshape(n, c, hh, 2, ww, 2).permute(0, 1, 3, 5, 2, 4).reshape(n, c * 4, hh, ww)
    x = torch.cat((x, torch.full((n, 1, hh, ww), sigma, dtype=x.dtype)), 1)
    for i in range(12):
        x = conv(x, W[f'model.{2*i}.weight'], W[f'model.{2*i}.bias'])
        if i < 11:
            x = torch.relu(x)
    return F.pixel_shuffle(x, 2)


def load_case(H=128):
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'ffdnet_color_weights.npz'))
    W = {k: torch.from_numpy(d[k]) for k in d.keys()}
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:H, 0:H] / H
    img = np.stack([0.5 + 0.3 * np.sin(7 * xx + 3 * yy + c) * np.cos(5 * yy - c) for c in range(3)])[None]
    img = np.clip(img + (25 / 255) * rng.standard_normal(img.shape), 0, 1)
    x64 = torch.from_numpy(img)
    ref = ffdnet(x64, 25 / 255, W, lambda x, w, b: F.conv2d(x, w.double(), b.double(), padding=1))
    rel = lambda a: float((a.double() - ref).norm() / ref.norm())
    return W, x64, ref, rel

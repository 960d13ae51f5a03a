#!/usr/bin/env python3
"""CPU: rounding error of the weight gradient of a 3x3 convolution accumulated in the Winograd F(4x4,3x3) domain in fp32
(dg = G^T [sum_tiles (A dY A^T) .* (B^T d B)] G, transforms and tile sums in fp32, G^T . G in double) against the F(2x2)
form the library uses today (csrc/wgrad_wino.hip) and the direct form, all against float64 autograd.  Inputs as in the FFDNet
trainer: post-ReLU activations (positive mean), zero-mean output gradients."""
import sys
import numpy as np
import torch

BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
               [0, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def wgrad_wino(x, dz, m, BT, G, AT, chunk=8):
    """x [n][ci][h][w], dz [n][co][h][w] (h, w multiples of m) -> dW [co][ci][3][3]; fp32 transforms, fp32 sums over tiles
    in `chunk`-tile MFMA-like partial sums"""
    n, ci, h, w = x.shape
    co = dz.shape[1]
    a = m + 2
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1))).astype(np.float32)
    ty, tx = h // m, w // m
    # patches [n][ci][ty][tx][a][a]
    pat = np.lib.stride_tricks.sliding_window_view(xp, (a, a), axis=(2, 3))[:, :, ::m, ::m]
    til = dz.astype(np.float32).reshape(n, co, ty, m, tx, m).transpose(0, 1, 2, 4, 3, 5)
    bt, at = BT.astype(np.float32), AT.astype(np.float32)
    V = np.einsum('ij,ncyxjk->ncyxik', bt, pat).astype(np.float32)
    V = np.einsum('ncyxik,lk->ncyxil', V, bt).astype(np.float32)
    M = np.einsum('ji,ncyxjk->ncyxik', at, til).astype(np.float32)
    M = np.einsum('ncyxik,kl->ncyxil', M, at).astype(np.float32)
    V = V.transpose(4, 5, 1, 0, 2, 3).reshape(a * a, ci, -1)
    M = M.transpose(4, 5, 1, 0, 2, 3).reshape(a * a, co, -1)
    S = np.zeros((a * a, co, ci), dtype=np.float32)
    T = V.shape[2]
    for t0 in range(0, T, 4096):                              # fp32 matmul (its own summation order), fp32 running sum
        S += np.matmul(M[:, :, t0:t0 + 4096], V[:, :, t0:t0 + 4096].transpose(0, 2, 1))
    S = S.astype(np.float64).reshape(a, a, co, ci)
    return np.einsum('ik,ijoc,jl->ockl', G, S, G)


def main():
    torch.manual_seed(0)
    n, c, h, w = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (4, 32, 128, 128)))
    x = torch.relu(torch.randn(n, c, h, w) * 0.5 + 0.2)
    dz = torch.randn(n, c, h, w) * 1e-3
    wt = torch.zeros(c, c, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.double(), wt, padding=1).backward(dz.double())
    ref = wt.grad.numpy()
    wt32 = torch.zeros(c, c, 3, 3, requires_grad=True)
    torch.nn.functional.conv2d(x, wt32, padding=1).backward(dz)
    rel = lambda a: float(np.linalg.norm(a - ref) / np.linalg.norm(ref))  # noqa: E731
    print(f'{n} x {c} ch x {h} x {w}: post-ReLU activations, dz ~ N(0, 1e-3)')
    print(f'  direct fp32 (PyTorch CPU)      rel-L2 {rel(wt32.grad.numpy().astype(np.float64)):.2e}')
    print(f'  Winograd F(2x2) domain, fp32   rel-L2 {rel(wgrad_wino(x.numpy(), dz.numpy(), 2, BT2, G2, AT2)):.2e}')
    print(f'  Winograd F(4x4) domain, fp32   rel-L2 {rel(wgrad_wino(x.numpy(), dz.numpy(), 4, BT4, G4, AT4)):.2e}')


if __name__ == '__main__':
    main()

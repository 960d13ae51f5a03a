"""Oracle (test infrastructure): SCI forward / transpose operators, Bayer layout helpers and
the two Euclidean-projection forms, restated on PyTorch-CPU tensors.

Layouts follow the reference: quarter-resolution Bayer planes stacked in the LAST dim,
`(M, N, B, 4)` with plane order R, G1, G2, B = offsets (0,0), (0,1), (1,0), (1,1).
"""
import torch

# RGGB offsets of the four Bayer planes (reference: dvp_linear_inv_2_stage_ADMM_tensor_online.py:51)
BAYER_OFFSETS = ((0, 0), (0, 1), (1, 0), (1, 1))


def forward_A(x, Phi):
    """y = sum_t x[..., t] * Phi[..., t]  (reference utilspy.py:28-33)."""
    return torch.sum(x * Phi, dim=2)


def transpose_At(y, Phi):
    """x[..., t] = y * Phi[..., t]  (reference utilspy.py:35-44)."""
    return torch.multiply(torch.repeat_interleave(torch.unsqueeze(y, dim=2), Phi.shape[2], dim=2), Phi)


def bayer_split(mosaic):
    """(H, W[, B]) mosaic -> (H/2, W/2[, B], 4) planes (reference utils/utils_image.py:145-151,
    and the setup loops dvp...:66-69)."""
    planes = [mosaic[dy::2, dx::2] for dy, dx in BAYER_OFFSETS]
    return torch.stack(planes, dim=-1).contiguous()


def bayer_merge(planes):
    """(M, N[, B], 4) planes -> (2M, 2N[, B]) mosaic (reference utils/utils_image.py:130-143,
    and the scatter loops dvp...:170-172)."""
    shape = list(planes.shape[:-1])
    shape[0] *= 2
    shape[1] *= 2
    out = torch.zeros(shape, dtype=planes.dtype)
    for ib, (dy, dx) in enumerate(BAYER_OFFSETS):
        out[dy::2, dx::2] = planes[..., ib]
    return out


def four_to_three_channel(planes):
    """(M,N,B,4) -> sparse (2M,2N,3,B) RGB (reference utils/utils_image.py:162-171)."""
    M, N, B = planes.shape[:3]
    rgb = torch.zeros(2 * M, 2 * N, 3, B)
    rgb[0::2, 0::2, 0, :] = planes[..., 0]
    rgb[0::2, 1::2, 1, :] = planes[..., 1]
    rgb[1::2, 0::2, 1, :] = planes[..., 2]
    rgb[1::2, 1::2, 2, :] = planes[..., 3]
    return rgb


def one_to_three_channel(mosaic):
    """(H,W,B) -> sparse (H,W,3,B) RGB (reference utils/utils_image.py:153-161)."""
    H, W, B = mosaic.shape
    rgb = torch.zeros(H, W, 3, B)
    rgb[0::2, 0::2, 0, :] = mosaic[0::2, 0::2, :]
    rgb[0::2, 1::2, 1, :] = mosaic[0::2, 1::2, :]
    rgb[1::2, 0::2, 1, :] = mosaic[1::2, 0::2, :]
    rgb[1::2, 1::2, 2, :] = mosaic[1::2, 1::2, :]
    return rgb


def cfa_masks(shape):
    """Boolean R, G, B site masks of an RGGB mosaic (reference utils/utils_image.py:106-112)."""
    R = torch.zeros(shape, dtype=torch.bool)
    G = torch.zeros(shape, dtype=torch.bool)
    B = torch.zeros(shape, dtype=torch.bool)
    R[0::2, 0::2] = True
    G[0::2, 1::2] = True
    G[1::2, 0::2] = True
    B[1::2, 1::2] = True
    return R, G, B


def rgb_to_bayer_planes(rgb):
    """Sample a dense (H,W,3,B) RGB cube at its CFA sites -> (M,N,B,4) planes
    (reference dvp...:206-209, test_ffdnet_ipol.py:275-278)."""
    return torch.stack([rgb[0::2, 0::2, 0, :], rgb[0::2, 1::2, 1, :],
                        rgb[1::2, 0::2, 1, :], rgb[1::2, 1::2, 2, :]], dim=-1)


def setup_planes(y_bayer, Phi_bayer, x0_bayer=None):
    """Solver prologue shared by both entry points (reference dvp...:59-83 / :347-370):
    split y, Phi into Bayer planes, Phi_sum with zeros replaced by ones, and the start point."""
    yall = bayer_split(y_bayer)
    Phiall = bayer_split(Phi_bayer)
    Phi_sum = torch.zeros_like(yall)
    # the reference sums each strided plane view separately (:72); keep that for bit parity
    for ib in range(4):
        Phi_sum[..., ib] = torch.sum(Phiall[..., ib], dim=2)
    Phi_sum[Phi_sum == 0] = 1
    if x0_bayer is None:
        x0 = torch.zeros_like(Phiall)
        for ib in range(4):
            x0[..., ib] = transpose_At(yall[..., ib], Phiall[..., ib])
    else:
        x0 = bayer_split(x0_bayer)
    return yall, Phiall, Phi_sum, x0


def project_two_stage(theta, b, Phiall, yall, Phi_sum, rho, alpha, out=None):
    """Two-stage ADMM Euclidean projection (reference dvp...:128-140):
        p = theta - (1/rho) b ;  x = p + Phi * ((y - A p) / (alpha rho + Phi_sum)).
    `out` may alias theta (the reference writes into xall which aliases theta_all at k=0)."""
    if out is None:
        out = torch.empty_like(theta)
    nmask = Phiall.shape[2]
    for ib in range(4):
        p = theta[..., ib] - (1 / rho) * b[..., ib]
        yb = forward_A(p, Phiall[..., ib])
        r = (yall[..., ib] - yb) / (alpha * rho + Phi_sum[..., ib])
        r = Phiall[..., ib] * torch.repeat_interleave(r.unsqueeze(2), nmask, dim=2)
        out[..., ib] = p + r
    return out


def project_one_stage(theta, b, Phiall, yall, Phi_sum, lam, gamma, out=None):
    """One-stage ("GAP form") projection (reference dvp...:389-391):
        v = theta + b ;  x = v + lambda * At((y - A v) / (Phi_sum + gamma))."""
    if out is None:
        out = torch.empty_like(theta)
    for ib in range(4):
        yb = forward_A(theta[..., ib] + b[..., ib], Phiall[..., ib])
        out[..., ib] = theta[..., ib] + b[..., ib] + lam * transpose_At(
            (yall[..., ib] - yb) / (Phi_sum[..., ib] + gamma), Phiall[..., ib])
    return out

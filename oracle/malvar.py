"""Oracle (test infrastructure): Malvar-He-Cutler (2004) gradient-corrected linear demosaicing of
an RGGB mosaic, as the reference's *torch* port evaluates it
(packages/colour_demosaicing/bayer/demosaicing/malvar2004.py:169-246): reflect-101 padding by 2
(`F.pad(..., 'reflect')`, :210 -- NOT the mirror boundary of the numpy variant at :37-160), four
5x5 correlations whose taps are built in float64, divided by 8 and cast to float32 (:174-208), and
a per-site selection (:213-240):

    R site : R = cfa,            G = GR_GB,  B = Rb_BB_Br_RR
    G site on an R row (G1) : R = Rg_RB_Bg_BR,      G = cfa,  B = Rg_RB_Bg_BR^T
    G site on a  B row (G2) : R = Rg_RB_Bg_BR^T,    G = cfa,  B = Rg_RB_Bg_BR
    B site : R = Rb_BB_Br_RR,    G = GR_GB,  B = cfa
"""
import torch
import torch.nn.functional as F


def malvar_taps():
    """The three base 5x5 tap tables (float32, already /8) in the order GR_GB, Rg_RB_Bg_BR,
    Rb_BB_Br_RR; the fourth filter is the transpose of the second."""
    k_g = torch.tensor([[0, 0, -1, 0, 0],
                        [0, 0, 2, 0, 0],
                        [-1, 2, 4, 2, -1],
                        [0, 0, 2, 0, 0],
                        [0, 0, -1, 0, 0]], dtype=torch.float64) / 8
    k_rb_row = torch.tensor([[0, 0, 0.5, 0, 0],
                             [0, -1, 0, -1, 0],
                             [-1, 4, 5, 4, -1],
                             [0, -1, 0, -1, 0],
                             [0, 0, 0.5, 0, 0]], dtype=torch.float64) / 8
    k_diag = torch.tensor([[0, 0, -1.5, 0, 0],
                           [0, 2, 0, 2, 0],
                           [-1.5, 0, 6, 0, -1.5],
                           [0, 2, 0, 2, 0],
                           [0, 0, -1.5, 0, 0]], dtype=torch.float64) / 8
    return k_g.float(), k_rb_row.float(), k_diag.float()


def malvar_demosaic(cfa):
    """cfa: (H, W) float32 RGGB mosaic -> (H, W, 3) RGB."""
    H, W = cfa.shape
    k_g, k_row, k_diag = malvar_taps()
    k_col = k_row.t().contiguous()
    padded = F.pad(cfa[None, None], (2, 2, 2, 2), mode='reflect')

    def corr(k):
        return F.conv2d(padded, k[None, None], stride=1)[0, 0]

    g_at_rb = corr(k_g)
    rb_row = corr(k_row)      # "Rg_RB_Bg_BR": R at G-in-R-row, B at G-in-B-row
    rb_col = corr(k_col)      # its transpose:  R at G-in-B-row, B at G-in-R-row
    rb_diag = corr(k_diag)    # R at B sites, B at R sites

    row_is_r = (torch.arange(H) % 2 == 0)[:, None].expand(H, W)
    col_is_r = (torch.arange(W) % 2 == 0)[None, :].expand(H, W)
    site_r = row_is_r & col_is_r
    site_g1 = row_is_r & ~col_is_r
    site_g2 = ~row_is_r & col_is_r
    site_b = ~row_is_r & ~col_is_r

    zero = torch.zeros_like(cfa)
    R = torch.where(site_r, cfa, zero)
    R = torch.where(site_g1, rb_row, R)
    R = torch.where(site_g2, rb_col, R)
    R = torch.where(site_b, rb_diag, R)
    G = torch.where(site_r | site_b, g_at_rb, cfa)
    B = torch.where(site_b, cfa, zero)
    B = torch.where(site_g2, rb_row, B)
    B = torch.where(site_g1, rb_col, B)
    B = torch.where(site_r, rb_diag, B)
    return torch.stack([R, G, B], dim=-1)


def malvar_demosaic_cube(mosaic):
    """(H, W, B) mosaic -> (H, W, 3, B), frame by frame like the solver loop (dvp...:186-191)."""
    H, W, nB = mosaic.shape
    out = torch.zeros(H, W, 3, nB)
    for t in range(nB):
        out[:, :, :, t] = malvar_demosaic(mosaic[:, :, t])
    return out

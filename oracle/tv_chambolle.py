"""Oracle (test infrastructure): Chambolle total-variation denoising, the TV prior of the
reference (call sites dvp_linear_inv_2_stage_ADMM_tensor_online.py:158 and :405:
`skimage.restoration.denoise_tv_chambolle(v, 0.1, n_iter_max=5, multichannel=True)`).

The routine lives in a third-party dependency that is NOT vendored in the reference:
scikit-image, pinned ==0.18.1 by the reference's readme.md:15.  This file restates the published
algorithm (Chambolle 2004, dual projection, tau = 1/4 in 2-D) exactly as scikit-image 0.18.x
evaluates it, operation by operation in the input dtype (float32 on the hot path):

    p = 0 (2,R,C);  i = 0
    while i < n_iter_max:
        if i > 0:  d = -(p0 + p1);  d[1:,:] += p0[:-1,:];  d[:,1:] += p1[:,:-1];  out = v + d
        else:      d = 0;  out = v
        E  = sum(d*d)                      (float32 array sum, then held in float64: see below)
        g0[:-1,:] = out[1:,:]-out[:-1,:] (last row 0);  g1[:,:-1] = out[:,1:]-out[:,:-1] (last col 0)
        nrm = sqrt(g0^2+g1^2);  E += weight*sum(nrm)
        p = (p - tau*g) / (1 + nrm*tau/weight);  E /= size
        if i == 0: E0 = Eprev = E
        elif |Eprev - E| < eps*E0: break
        else: Eprev = E
        i += 1
    return out

Pinned (tests/test_oracle_golden.py) against vectors produced in the build container by the
scikit-image 0.18.3 source itself (tools/make_golden.py, group G4), including channels that stop
early.  `stop_iter` reports, per channel, the iteration index whose `out` was returned.
"""
import numpy as np


def tv_chambolle_2d(image, weight=0.1, eps=2.e-4, n_iter_max=200, return_info=False):
    image = np.asarray(image)
    dt = image.dtype
    p = np.zeros((2,) + image.shape, dtype=dt)
    g = np.zeros_like(p)
    d = np.zeros_like(image)
    tau = 1. / 4.
    i = 0
    out = image
    energies = []
    E_init = E_prev = None
    while i < n_iter_max:
        if i > 0:
            d = -p.sum(0)
            d[1:, :] += p[0, :-1, :]
            d[:, 1:] += p[1, :, :-1]
            out = image + d
        else:
            out = image
        # NumPy 1.x promotion (the reference's era: scikit-image 0.18 / numpy<2): the two float32
        # array sums become float64 the moment they meet the Python float `weight`, so the energy,
        # its normalisation and the stop test are evaluated in double on float32 partial sums.
        E = np.float64((d ** 2).sum())
        g[0, :-1, :] = np.diff(out, axis=0)
        g[1, :, :-1] = np.diff(out, axis=1)
        norm = np.sqrt((g ** 2).sum(axis=0))[np.newaxis, ...]
        E += weight * np.float64(norm.sum())
        norm *= tau / weight
        norm += 1.
        p -= tau * g
        p /= norm
        E /= float(image.size)
        energies.append(float(E))
        if i == 0:
            E_init = E
            E_prev = E
        else:
            if np.abs(E_prev - E) < eps * E_init:
                break
            E_prev = E
        i += 1
    if return_info:
        return out, min(i, n_iter_max - 1), energies
    return out


def tv_chambolle_multichannel(image, weight=0.1, eps=2.e-4, n_iter_max=200, return_info=False):
    """Channel-by-channel 2-D Chambolle over the last axis (skimage `multichannel=True`)."""
    image = np.asarray(image)
    out = np.zeros_like(image)
    stops, energies = [], []
    for c in range(image.shape[-1]):
        r = tv_chambolle_2d(image[..., c], weight, eps, n_iter_max, return_info=True)
        out[..., c] = r[0]
        stops.append(r[1])
        energies.append(r[2])
    if return_info:
        return out, np.asarray(stops, np.int32), energies
    return out

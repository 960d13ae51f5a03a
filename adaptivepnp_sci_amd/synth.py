"""Seeded synthetic SCI problems (the CACTI dataset of the reference, readme.md:22, is not
shipped): a smooth moving texture as ground truth, an iid Bernoulli(0.5) binary coding mask and the
noise-free snapshot measurement y = sum_t Phi_t x_t.  NumPy only; used by bench.py, the tests and
the golden-vector generator so that every leg sees the same inputs."""
import numpy as np


def make_cube(H, W, B, seed=0):
    """Ground-truth mosaic cube (H,W,B) float32 in [0,1]: six random 2-D sinusoids translated
    1.5 px/frame plus 0.05*U(0,1), min-max scaled."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    cube = np.zeros((H, W, B), np.float64)
    for _ in range(6):
        fx, fy = rng.uniform(0.01, 0.12, 2)
        ph = rng.uniform(0, 2 * np.pi)
        amp = rng.uniform(0.3, 1.0)
        dx, dy = rng.uniform(-1, 1, 2)
        nrm = np.hypot(dx, dy) + 1e-9
        dx, dy = 1.5 * dx / nrm, 1.5 * dy / nrm
        for t in range(B):
            cube[:, :, t] += amp * np.sin(2 * np.pi * (fx * (xx - dx * t) + fy * (yy - dy * t)) + ph)
    cube += 0.05 * rng.uniform(0, 1, cube.shape)
    cube -= cube.min()
    cube /= cube.max()
    return cube.astype(np.float32)


def make_mask(H, W, B, seed=0):
    rng = np.random.default_rng(seed + 1000003)
    return (rng.uniform(0, 1, (H, W, B)) < 0.5).astype(np.float32)


def make_problem(H, W, B, seed=0):
    """-> (y (H,W), Phi (H,W,B), orig (H,W,B)), all float32; y in [0,B]."""
    orig = make_cube(H, W, B, seed)
    Phi = make_mask(H, W, B, seed)
    y = np.sum(orig * Phi, axis=2, dtype=np.float32).astype(np.float32)
    return y, Phi, orig

"""Seeded synthetic SCI problems (the CACTI dataset of the reference, readme.md:22, is not
shipped): a smooth moving texture as ground truth, an iid Bernoulli(0.5) binary coding mask and the
noise-free snapshot measurement y = sum_t Phi_t x_t -- used by bench.py, the tests and the golden-vector generator so
that every leg sees the same inputs -- and seeded synthetic weights for the networks whose checkpoints are not in the
reference snapshot (FastDVDnet model.pth, DDnet ddnet1.pth: .MISSING_LARGE_BLOBS)."""
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


def synth_fastdvdnet(seed=0):
    """A `FastDVDnet` container with seeded synthetic weights: Kaiming-normal convs (activations stay O(1) through the
    U-Net), the last conv of each DenBlock scaled by 0.05 so that the predicted residual is small and the PnP loop stays
    bounded, BatchNorm affine / running statistics randomised so that the BN fold is exercised.  Same recipe and
    generator stream as oracle.nets.synth_fastdvdnet_weights (asserted by tests/test_oracle_golden.py)."""
    import torch
    from .fastdvd import FastDVDnet
    g = torch.Generator().manual_seed(seed)
    net = FastDVDnet()
    sd = net.state_dict()
    for k, v in sd.items():
        if k.endswith('num_batches_tracked'):
            continue
        if v.dim() == 4:
            last = 0.05 if k.endswith('outc.convblock.3.weight') else 1.0
            sd[k] = torch.randn(v.shape, generator=g) * (last * (2.0 / (v.shape[1] * 9)) ** 0.5)
        elif k.endswith('running_var'):
            sd[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith('running_mean'):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith('weight'):
            sd[k] = 0.75 + 0.5 * torch.rand(v.shape, generator=g)
        elif k.endswith('bias'):
            sd[k] = 0.05 * torch.randn(v.shape, generator=g)
    net.load_state_dict(sd)
    return net


def synth_ddnet(seed=0):
    """A `DDnet` container with seeded synthetic weights (recipe of oracle.nets.synth_ddnet_weights): Kaiming-normal
    convs, small last convs, non-trivial gate scalars, and a fusion block that behaves like a crude demosaicker."""
    import torch
    from .ddnet import DDnet
    g = torch.Generator().manual_seed(seed)
    net = DDnet()
    sd = net.state_dict()
    for k, v in sd.items():
        if v.dim() == 4:
            last = 0.05 if k.endswith('outc.convblock.2.weight') else 1.0
            sd[k] = torch.randn(v.shape, generator=g) * (last * (2.0 / (v.shape[1] * 9)) ** 0.5)
        elif k == 'weight_tensor_out':
            sd[k] = 0.5 + 0.05 * torch.randn(v.shape, generator=g)
        else:
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
    f0, f2 = sd['temp11.fusion.convblock.0.weight'] * 0.1, sd['temp11.fusion.convblock.2.weight'] * 0.1
    for c in range(4):
        f0[c, c, 1, 1] += 1.0
    f2[0, 0, 1, 1] += 1.0
    f2[1, 1, 1, 1] += 0.5
    f2[1, 2, 1, 1] += 0.5
    f2[2, 3, 1, 1] += 1.0
    sd['temp11.fusion.convblock.0.weight'], sd['temp11.fusion.convblock.2.weight'] = f0, f2
    net.load_state_dict(sd)
    return net

#!/usr/bin/env python3
"""GPU box: where does the HIP FastDVDnet finetune gradient leave the float64 gradient, and why?

One forward / backward of the trainer on a 64x64x8 problem with a tap on the gradient at every layer's (BatchNorm) output of
the stage-2 DenBlock, against float64 autograd of the oracle network on the same inputs -- twice:
  pass A: the float64 network with its OWN ReLU masks;
  pass B: the float64 network differentiating with the HIP run's masks (out = x * mask_hip): identical values except at
          pre-activations within round-off of zero, where the derivative of ReLU is decided by the last bit.
If the HIP arithmetic is sound, pass B agrees to ~1e-6 and the mask disagreements of pass A sit at |z| / max|z| ~ 1e-7."""
import os, sys
import numpy as np, torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('SCIPNP_CONV_PRECISION', 'f32')
from adaptivepnp_sci_amd import finetune, ops, synth
from adaptivepnp_sci_amd.fastdvd import FastDVDEngine, _LAYERS
from oracle import nets as ON, denoisers as OD, sci_ops as OO

torch.set_num_threads(8)
B, H, W = 8, 64, 64
SEED = int(os.environ.get('PROBE_SEED', 1))
y, Phi, orig = synth.make_problem(H, W, B, seed=5)
rng = np.random.default_rng(SEED)
v = np.clip(np.repeat(orig[:, :, None, :], 3, 2) + 0.05 * rng.standard_normal((H, W, 3, B)), 0, 1).astype(np.float32)
noise = rng.normal(0, 5 / 255, (B, 3, H, W))
frames = torch.from_numpy(np.ascontiguousarray(v.transpose(3, 2, 0, 1)))          # (B,3,H,W)
sigma = 8 / 255
RELU_LAYERS = [i for i, l in enumerate(_LAYERS) if l[4]]
STASH = {0: 't96', 1: 'x0', 2: 'a0', 3: 'a1', 4: 'x1', 5: 'd0', 6: 'd1', 7: 'x2', 8: 'u0', 9: 'u1', 11: 'c0', 12: 'c1', 14: 'o32'}


class MaskReLU(nn.Module):
    """ReLU whose derivative is a GIVEN 0/1 mask (call k uses masks[k]): out = x * mask"""

    def __init__(self):
        super().__init__()
        self.masks, self.k, self.mismatch, self.zrel = None, 0, 0, 0.0

    def forward(self, x):
        if self.masks is None:
            return torch.relu(x)
        m = self.masks[self.k:self.k + 1].to(x.dtype)
        self.k += 1
        bad = (m > 0) != (x > 0)
        self.mismatch += int(bad.sum())
        if bad.any():
            self.zrel = max(self.zrel, float((x.abs() * bad).max() / x.abs().max()))
        return x * m


def mod_of(root, path):
    m = root
    for part in path.split('.'):
        m = m[int(part)] if part.isdigit() else getattr(m, part)
    return m


def oracle64(masks_by_layer):
    """float64 forward / backward; returns ({layer: gradient at its (BN) output, frames stacked}, relu modules, loss)"""
    net = ON.cpu_data_parallel(ON.synth_fastdvdnet_weights(0)).double()
    relus = []

    def swap(mod):
        for name, ch in mod.named_children():
            if isinstance(ch, nn.ReLU):
                r = MaskReLU()
                setattr(mod, name, r)
                relus.append(r)
            else:
                swap(ch)
    swap(net.module.temp2)
    assert len(relus) == len(RELU_LAYERS)
    if masks_by_layer is not None:
        for r, i in zip(relus, RELU_LAYERS):
            r.masks = masks_by_layer[i]
    net.train()
    for m in net.module.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
    taps = {i: [] for i in range(len(_LAYERS))}
    for i, (key, bn, *_r) in enumerate(_LAYERS):
        m = mod_of(net.module.temp2, bn if bn is not None else key)

        def fh(mod, inp, out, i=i):
            out.register_hook(lambda g, i=i: taps[i].append(g.detach().clone()))
        m.register_forward_hook(fh)
    vv = frames.double()
    v_plus = vv + torch.from_numpy(vv.numpy().astype(np.float64) + noise).float().double()
    yp, Pp = OO.bayer_split(torch.from_numpy(y)), OO.bayer_split(torch.from_numpy(Phi))
    Phi_m, y_m = OO.bayer_merge(Pp).double(), OO.bayer_merge(yp).double()
    nm = torch.tensor([sigma], dtype=torch.float64).expand((1, 1, H, W))
    den = torch.empty((B, 3, H, W), dtype=torch.float64)
    for n in range(B):
        idx = (torch.arange(n, n + 5) - 2) % B
        den[n] = net(v_plus[idx].reshape((1, -1, H, W)), nm)
    loss = nn.MSELoss()(torch.sum(OD._rgb_cube_to_mosaic(den.permute(2, 3, 1, 0)) * Phi_m, dim=2), y_m)
    loss.backward()
    # the hooks fire frame by frame in reverse order of the forward loop
    return {i: torch.cat(list(reversed(t)), 0) for i, t in taps.items()}, relus, float(loss.detach())


# ---- HIP trainer, one step, taps on temp2
hnet = ON.cpu_data_parallel(ON.synth_fastdvdnet_weights(0))
eng = FastDVDEngine(hnet, B, H, W, torch.device('cuda'))
hip = {}
finetune.TAP = lambda name, i, dy: hip.__setitem__((name, i), dy.clone())
y_pm, Phi_pm = ops.y_to_meas(torch.from_numpy(y).cuda()), ops.mosaic_to_state(torch.from_numpy(Phi).cuda())
vp = ops.fastdvd_noisy_input(frames.cuda().contiguous(), torch.from_numpy(noise).cuda())
tr = finetune._FastDVDTrainer(hnet, eng)
tr.pack()
tr.forward(vp, sigma)
l = tr.loss_and_grad(y_pm, Phi_pm)
tr.backward_block('temp2', tr.dout, tr.ds1)
torch.cuda.synchronize()
split = eng.precision == 'f16x3'
inv = 1.0 / tr.gscale if split else 1.0
masks = {}
for i in RELU_LAYERS:
    st = tr.stash['temp2'][STASH[i]]
    if split:
        st = ops.c8s_to_c8(st)
    masks[i] = (ops.from_c8(st, _LAYERS[i][3] if _LAYERS[i][3] != 96 else 90).cpu() > 0)
    if i == 0:
        masks[i] = masks[i][:, :90]


def hip_dz(i):
    key, bn, cin, cout, relu, s2, shuf = _LAYERS[i]
    dz = hip[('temp2', i)]
    if split:
        dz = ops.c8s_to_c8(dz, scale=inv)
    d = ops.from_c8(dz, cout).cpu().double()
    return d[:, :, ::2, ::2] if s2 else d


for label, mk in (('A: float64 network with its own masks', None), ('B: float64 network with the HIP masks', masks)):
    g64, relus, loss64 = oracle64(mk)
    print(f'--- pass {label}; float64 loss {loss64:.12f}, HIP loss {float(l.item()):.12f} ({eng.precision}), seed {SEED}')
    print(' layer                                       |dz_hip - dz_64| / |dz_64|   mask disagreements  max |z|/max|z| there')
    rel = dict(zip(RELU_LAYERS, relus))
    for i, (key, bn, cin, cout, relu, s2, shuf) in enumerate(_LAYERS):
        d = hip_dz(i)[:, :g64[i].shape[1]]
        err = float((d - g64[i]).norm() / g64[i].norm())
        extra = ''
        if mk is not None and i in rel:
            extra = f'{rel[i].mismatch:10d} of {masks[i].numel():9d}      {rel[i].zrel:.2e}'
        print(f' {i:2d} {key:38s} {err:10.3e}               {extra}')

#!/usr/bin/env python3
"""GPU box: s_memtime stamps of the F(4x4,3x3) Winograd kernel (scipnp_conv3x3_c8w4_stamped) on the FFDNet body layer:
where a workgroup's life goes -- prologue, the two k-steps of a channel group (MFMA issue phase | wait + barrier), epilogue.
All figures are in units of 100 shader-clock cycles (s_memtime); W4_STAMP_OFF = parts switched off (timing only), SCIPNP_W4_ONE_PER_CU=1 = a
single resident workgroup per CU (a lone wave's pace), W4_STAMP_DETAIL=1 = every k-step of four consecutive workgroups of one CU."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import _lib, ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
import diaglib  # noqa: E402  (libscipnp_diag.so: the laboratory entries)
lib = diaglib.load()
n, c, h, w = 8, 96, 256, 256
if os.environ.get('W4_SHAPE'):                      # 'n,cin,cout,h,w' -- e.g. the narrow layers: 8,16,96,512,512 (FastDVDnet inc conv 1)
    n, c, co_, h, w = (int(v) for v in os.environ['W4_SHAPE'].split(','))
else:
    co_ = c
g = torch.Generator().manual_seed(0)
x8 = ops.to_c8(torch.randn(n, c, h, w, generator=g).cuda())
pk = ops.pack_conv3x3(torch.randn(co_, c, 3, 3, generator=g) * 0.05, torch.randn(co_, generator=g), Cin=c, Cout=co_, device='cuda')
p4 = ops.pack_conv3x3_wino4(pk, c, co_)
out = torch.empty(n, (co_ + 7) // 8, h, w, 8, device='cuda')
ref = torch.empty_like(out)
nwg = (w // 64) * (h // (16 if os.environ.get('W4_KERNEL', '4') == '6' else 8)) * n * ((co_ + 31) // 32)
st = torch.zeros(nwg * 128, dtype=torch.int64, device='cuda')
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KER = os.environ.get('W4_KERNEL', '4')            # '6': the three-waves-per-SIMD kernel (scipnp_conv3x3_c8w6_stamped; masks 0, 1, 6, 7)
if KER in ('6', 'n'):                                        # the not-adopted kernels of lab/ (needs `make -C lab`)
    sys.path.insert(0, os.path.join(ROOT, 'lab'))
    import lablib  # noqa: E402
    stamped = lablib.load().scipnp_conv3x3_c8w6_stamped if KER == '6' else lablib.load().scipnp_conv3x3_c8wn_stamped
else:
    stamped = lib.scipnp_conv3x3_c8w4_stamped
if KER == 'n':                                          # 16-channel workgroups: re-laid weights, twice the workgroups
    p4w = lablib.repack_wino4n(p4, c, co_)
    st = torch.zeros(2 * nwg * 128, dtype=torch.int64, device='cuda')
    nwg *= 2
else:
    p4w = p4
print('kernel: scipnp_conv3x3_c8w' + KER)
OFF = int(os.environ.get('W4_STAMP_OFF', '0'))       # diag bits 0..2 (no transform | no raw staging | no U DMA): timing only
FL = 1 | (OFF << 12)
print('parts switched off (diag mask):', OFF)
for _ in range(3):
    _lib.check(stamped(P(x8), P(p4w), P(out), n, c, co_, h, w, FL, P(st), _lib.stream_ptr()), 'stamped')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.check(stamped(P(x8), P(p4w), P(out), n, c, co_, h, w, FL, P(st), _lib.stream_ptr()), 'stamped')
e1.record()
torch.cuda.synchronize()
ops.conv3x3_c8w4(x8, p4, co_, relu=True, out=ref)
print('stamped launch', round(e0.elapsed_time(e1) * 1e3, 1), 'us; equals the product kernel:', bool(torch.equal(out, ref)))
s = st.cpu().numpy().reshape(nwg, 128).astype(np.float64)
CG = c // 8
t0 = s[:, 0][s[:, 0] > 0].min()
life = s[:, 6] - s[:, 0]
us = lambda d: d / 100.0     # noqa: E731  -- s_memtime counts shader clocks here: print HUNDREDS OF CYCLES
print(f'workgroups {nwg}; workgroup life mean {us(life.mean()):.1f} (min {us(life.min()):.1f}, max {us(life.max()):.1f})   [x 100 cycles]')
print(f'  entry -> tiles/slab landed {us((s[:, 1] - s[:, 0]).mean()):.2f} | first column pass {us((s[:, 2] - s[:, 1]).mean()):.2f} | '
      f'loop {us((s[:, 3] - s[:, 2]).mean()):.2f} | drain + exchange {us((s[:, 4] - s[:, 3]).mean()):.2f} | output transform + stores issued '
      f'{us((s[:, 5] - s[:, 4]).mean()):.2f} | stores acknowledged {us((s[:, 6] - s[:, 5]).mean()):.2f}')
G = s[:, 8:8 + 4 * CG].reshape(nwg, CG, 4)
prev = np.concatenate([s[:, 2:3], G[:, :-1, 3]], axis=1)          # start of each group
a_issue, a_wait = G[:, :, 0] - prev, G[:, :, 1] - G[:, :, 0]
b_issue, b_wait = G[:, :, 2] - G[:, :, 1], G[:, :, 3] - G[:, :, 2]
print('per channel group (mean over workgroups and groups), x 100 cycles:')
print(f'  k-step 0: row pass + 36 MFMAs issued {us(a_issue.mean()):.3f} | wait (U, raw) + barrier {us(a_wait.mean()):.3f}')
print(f'  k-step 1: 36 MFMAs + column pass     {us(b_issue.mean()):.3f} | wait (U) + barrier      {us(b_wait.mean()):.3f}')
print(f'  group total {us((a_issue + a_wait + b_issue + b_wait).mean()):.3f}   (72 MFMAs alone on a SIMD: 23.04; two waves sharing it: 46.08)')
for name, arr in (('k-step 0 issue', a_issue), ('k-step 0 wait', a_wait), ('k-step 1 issue', b_issue), ('k-step 1 wait', b_wait)):
    q = np.percentile(us(arr), [10, 50, 90, 99])
    print(f'  {name:15s} p10 {q[0]:.3f}  p50 {q[1]:.3f}  p90 {q[2]:.3f}  p99 {q[3]:.3f}')
print('  by group index (mean, x 100 cycles): ' + ' '.join(f'{us((a_issue + a_wait + b_issue + b_wait)[:, k].mean()):.2f}' for k in range(CG)))
# ---- one CU's workgroups in start order (cycles from the launch's first entry)
hw = st.cpu().numpy().reshape(nwg, 128)[:, 7]
xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
key = (xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 7) | ((hwid >> 8) & 15)
keys = np.unique(key)
print(f'distinct (xcc, se, sh, cu): {len(keys)}; workgroups per CU: min {min((key == k).sum() for k in keys)} max {max((key == k).sum() for k in keys)}')
k0 = keys[len(keys) // 2]
idx = np.where(key == k0)[0]
idx = idx[np.argsort(s[idx, 0])]
print('one CU (kilo-cycles from launch start): blockIdx | entry | tiles landed | loop start | loop end | stores issued | acked | simd/wave slot')
for i in idx:
    r = s[i]
    print(f'  {i:5d} ' + ' '.join(f'{(r[j] - t0) / 1e3:8.1f}' for j in (0, 1, 2, 3, 5, 6)) + f'   simd {(int(hwid[i]) >> 4) & 3} wave {int(hwid[i]) & 15}')
print('by group index (mean cycles / 100):')
for name, arr in (('k-step 0 issue', a_issue), ('k-step 0 wait ', a_wait), ('k-step 1 issue', b_issue), ('k-step 1 wait ', b_wait)):
    print(f'  {name}: ' + ' '.join(f'{arr[:, k].mean() / 100:6.2f}' for k in range(CG)))
if os.environ.get('W4_STAMP_DETAIL'):
    # three consecutive workgroups of that CU, every k-step: start of the step's MFMAs' issue end and barrier end, relative cycles
    base = s[idx[4], 0]
    for i in idx[4:8]:
        r = s[i]
        ev = [('entry', r[0]), ('landed', r[1]), ('col0', r[2])]
        for gi in range(CG):
            ev += [(f'g{gi}A', r[8 + 4 * gi]), (f'g{gi}a', r[9 + 4 * gi]), (f'g{gi}B', r[10 + 4 * gi]), (f'g{gi}b', r[11 + 4 * gi])]
        ev += [('loopend', r[3]), ('xchg', r[4]), ('stores', r[5]), ('acked', r[6])]
        print(f'WG {i}: ' + ' '.join(f'{n}:{(t - base) / 1e3:.1f}' for n, t in ev))

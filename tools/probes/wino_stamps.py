#!/usr/bin/env python3
"""GPU box: where a workgroup of the fp32 Winograd body layer (96 -> 96, 8 frames of 256x256) spends its life.
The diagnostic instantiation of the kernel (scipnp_conv3x3_c8w_stamped) leaves six s_memtime stamps, its HW_ID / XCC_ID and
an s_memrealtime per workgroup; this script turns them into per-phase means and a per-CU occupancy timeline."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import _lib, ops  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
import diaglib  # noqa: E402  (libscipnp_diag.so: the laboratory entries)
lib = diaglib.load()
dev = torch.device('cuda:0')
B, Cn, M, N = 8, int(os.environ.get('WS_C', 96)), 256, 256
torch.manual_seed(0)
x = torch.rand(B, Cn // 8, M, N, 8, device=dev)
w = torch.randn(Cn, Cn, 3, 3) * 0.05
pk = ops.pack_conv3x3(w, torch.zeros(Cn), Cin=Cn, Cout=Cn, device=dev)
pkw = ops.pack_conv3x3_wino(pk, Cn, Cn)
out = torch.empty_like(x)
nwg = (N // 32) * (M // 8) * B * (Cn // 32)
st = torch.zeros(nwg, 80, dtype=torch.int64, device=dev)
DETAIL = int(os.environ.get('WS_DETAIL', 0))
s = _lib.stream_ptr()
import time
t_end = time.time() + float(os.environ.get('WS_HEAT_S', 2.0))     # the clock the part holds under this kernel: heat first
n_heat = 0
while time.time() < t_end:
    for _ in range(50):
        ops.conv3x3_c8w(x, pkw, Cn, relu=True, out=out)
    torch.cuda.synchronize()
    n_heat += 50
h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
h0.record()
for _ in range(50):
    ops.conv3x3_c8w(x, pkw, Cn, relu=True, out=out)
h1.record()
torch.cuda.synchronize()
prod_us = h0.elapsed_time(h1) / 50 * 1e3
print(f'product kernel after {n_heat} launches: {prod_us:.1f} us per launch')
for _ in range(20):
    ops.conv3x3_c8w(x, pkw, Cn, relu=True, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.check(lib.scipnp_conv3x3_c8w_stamped(C.c_void_p(x.data_ptr()), C.c_void_p(pkw.data_ptr()), C.c_void_p(out.data_ptr()),
                                          B, Cn, Cn, M, N, 1 | (0x800 if DETAIL else 0), C.c_void_p(st.data_ptr()), s), 'stamped')
e1.record()
torch.cuda.synchronize()
t = st.cpu().numpy().astype(np.int64)
print(f'stamped launch: {e0.elapsed_time(e1) * 1e3:.1f} us, {nwg} workgroups')
T = t[:, :6].copy()
hw = t[:, 6]
xcc = (hw >> 32) & 0xF
# in-kernel clock: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz), workgroup by workgroup
clk = (T[:, 5] - T[:, 0]).sum() / ((t[:, 7] - t[:, 31]).sum() / 100.0)
tick_us = 1.0 / float(clk)
hwid = hw & 0xFFFFFFFF
key = ((xcc * 8 + ((hwid >> 13) & 0x7)) * 2 + ((hwid >> 12) & 1)) * 16 + ((hwid >> 8) & 0xF)      # (xcc, se, sh, cu)
cus = np.unique(key)
rates = []
for xc in np.unique(xcc):
    m = xcc == xc
    rates.append(((T[m, 5] - T[m, 0]).sum() / ((t[m, 7] - t[m, 31]).sum() / 100.0)))
origin = np.zeros(len(t), dtype=np.int64)          # s_memtime origins differ between XCDs (and shader engines): rebase per CU
for k in cus:
    m = key == k
    origin[m] = T[m, 0].min()
T -= origin[:, None]
base, span = 0, float(np.mean([T[key == k, 5].max() for k in cus]))
print(f's_memtime: {1 / tick_us:.1f} ticks/us = in-kernel clock in MHz (per XCD: ' + ' '.join(f'{r:.0f}' for r in rates) + f'); launch span {span * tick_us:.1f} us')
mfma_cycles = nwg * 4 * (Cn // 8) * 64 * 32 / 1024.0          # 64 MFMAs of 32 cycles per wave and group, 1024 SIMDs
print(f'matrix-pipe cycles needed per SIMD: {mfma_cycles:.0f}; product launch at this clock: {prod_us / tick_us:.0f} cycles -> '
      f'pipes busy {mfma_cycles / (prod_us / tick_us):.3f} of the cycles; at 2400 MHz the same cycles would take {prod_us / tick_us / 2400:.1f} us')
us = lambda d: d * tick_us  # noqa: E731
names = ['entry -> first tiles + U in LDS', 'first input transform', 'channel-group loop', 'output transform + store issue',
         'store acknowledge']
for i, nme in enumerate(names):
    d = us(T[:, i + 1] - T[:, i])
    print(f'  {nme:34s} mean {d.mean():7.2f} us   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}')
life = us(T[:, 5] - T[:, 0])
print(f'  workgroup lifetime                 mean {life.mean():7.2f} us   p10 {np.percentile(life, 10):7.2f}   p90 {np.percentile(life, 90):7.2f}')
print(f'distinct (xcc, se, sh, cu): {len(cus)}')
# per CU: time with 0 / 1 / 2 workgroups inside the channel-group loop, and with 0 resident
tot = np.zeros(4)
idle_res = 0.0
for k in cus:
    m = key == k
    ev = []
    for a_, b_ in zip(T[m, 2], T[m, 3]):
        ev += [(a_, 1), (b_, -1)]
    ev.sort()
    cur, last = 0, base
    for tt, dlt in ev:
        tot[min(cur, 3)] += tt - last
        cur += dlt
        last = tt
    ev = []
    for a_, b_ in zip(T[m, 0], T[m, 5]):
        ev += [(a_, 1), (b_, -1)]
    ev.sort()
    cur, last = 0, base
    for tt, dlt in ev:
        if cur == 0:
            idle_res += tt - last
        cur += dlt
        last = tt
tot /= tot.sum()
print(f'per-CU share of the launch with 0 / 1 / 2 / 3+ workgroups inside the channel-group loop: '
      + ' / '.join(f'{v:.3f}' for v in tot))
print(f'per-CU share with no workgroup resident: {idle_res / (len(cus) * span):.3f}')
wpc = np.array([np.sum(key == k) for k in cus])
print(f'workgroups per CU: min {wpc.min()} mean {wpc.mean():.1f} max {wpc.max()}')
# start-to-start gap on a CU slot: how long after one workgroup ends does the next start there
gaps = []
for k in cus:
    m = key == k
    starts, ends = np.sort(T[m, 0]), np.sort(T[m, 5])
    for e_ in ends[:-2]:
        nxt = starts[starts > e_]
        if len(nxt):
            gaps.append(nxt[0] - e_)
print(f'end of a workgroup -> next workgroup entry on that CU: mean {us(np.mean(gaps)):.2f} us, p50 {us(np.median(gaps)):.2f}, p90 {us(np.percentile(gaps, 90)):.2f}')
k = cus[len(cus) // 2]
m = key == k
order = np.argsort(T[m, 0])
print('one CU, its workgroups in start order (us): entry | tiles landed | loop start | loop end | stores issued | acked | simd/wave slot')
for row, h_ in zip(T[m][order], hwid[m][order]):
    print('   ' + ' '.join(f'{us(v):8.2f}' for v in row) + f'   simd {(h_ >> 4) & 3} wave {h_ & 15}')

# per-group durations of a workgroup's first wave, split by whether the CU's other workgroup was inside its loop
G = t[:, 8:8 + Cn // 8] - origin[:, None]
alone, shared = [], []
for k in cus:
    m = np.where(key == k)[0]
    for i in m:
        bounds = np.concatenate([[T[i, 2]], G[i]])
        for g in range(len(G[i])):
            a_, b_ = bounds[g], bounds[g + 1]
            mid = 0.5 * (a_ + b_)
            other = [j for j in m if j != i and T[j, 2] <= mid <= T[j, 3]]
            (shared if other else alone).append(b_ - a_)
print(f'channel group, first wave: partner workgroup in its loop: mean {us(np.mean(shared)):.3f} us (n={len(shared)}); '
      f'partner outside its loop: mean {us(np.mean(alone)):.3f} us (n={len(alone)})   ')

if DETAIL:
    Q = (t[:, 32:64] - origin[:, None]).reshape(len(t), 2, 16)
    seg = {True: [], False: []}
    for k in cus:
        m = np.where(key == k)[0]
        for i in m:
            for gi, g in enumerate((4, 5)):
                start = G[i][g - 1]
                mid = 0.5 * (start + G[i][g])
                other = any(j != i and T[j, 2] <= mid <= T[j, 3] for j in m)
                q = Q[i, gi]
                seg[other].append(np.concatenate([[q[0] - start], np.diff(q), [G[i][g] - q[15]]]))
    for other in (True, False):
        a_ = np.mean(np.array(seg[other]), axis=0)
        a_ = a_ / a_[5:12].mean()
        print(('partner in loop:  ' if other else 'partner outside:  ') + 'time per position 0..15 and the end-of-group wait + barrier, '
              'in units of the mean of positions 5..11:\n    ' + ' '.join(f'{v:.2f}' for v in a_) + f'   sum {a_.sum():.2f}')

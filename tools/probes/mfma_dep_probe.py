#!/usr/bin/env python3
"""GPU box: does the distance between two uses of an accumulator matter for v_mfma_f32_16x16x4_f32?  (scipnp_bench_mfma_dep:
32 MFMAs per round on 16 accumulators, second use DIST MFMAs behind the first; 1, 2, 3 waves per SIMD)"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
import diaglib  # noqa: E402  (libscipnp_diag.so: the laboratory entries)
lib = diaglib.load()
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
iters = 4000
for wps in (1, 2, 3):
    blocks = 256 * wps
    out = torch.empty(blocks * 256, device='cuda')
    cyc = torch.zeros(blocks * 4, dtype=torch.int64, device='cuda')
    for dist in (1, 2, 4, 8, 16):
        fn = lambda: _lib.check(lib.scipnp_bench_mfma_dep(P(out), P(cyc), blocks, iters, dist, _lib.stream_ptr()), 'dep')  # noqa: E731
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3
        c = cyc.cpu().numpy().astype(np.float64)
        tf = blocks * 4 * iters * 32 * 2048 / t / 1e12
        # s_memtime ticks at a constant 100 MHz: ticks * (core clock / 100 MHz) = core cycles; report the wall-clock rate instead
        print(f'{wps} wave(s)/SIMD  dist {dist:2d}: {tf:7.1f} TFLOP/s  ({tf / 157.3:.3f} of 157.3)   memtime ticks/MFMA/wave {np.median(c) / (iters * 32):.3f}')

print('--- operand register banks (scipnp_bench_mfma_bank): 0 A,B different banks; 1 same bank; 2 float4 fragments u[nu] x v[nu] (same bank); 3 rotated')
for wps in (1, 3):
    blocks = 256 * wps
    out = torch.empty(blocks * 256, device='cuda')
    cyc = torch.zeros(blocks * 4, dtype=torch.int64, device='cuda')
    for var in (0, 1, 2, 3):
        fn = lambda: _lib.check(lib.scipnp_bench_mfma_bank(P(out), P(cyc), blocks, iters, var, _lib.stream_ptr()), 'bank')  # noqa: E731
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3
        tf = blocks * 4 * iters * 32 * 2048 / t / 1e12
        print(f'{wps} wave(s)/SIMD  var {var}: {tf:7.1f} TFLOP/s  ({tf / 157.3:.3f} of 157.3)')

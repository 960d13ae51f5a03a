#!/usr/bin/env python3
"""GPU box: cycles per MFMA of one wave's loop with NV v_add_f32 after every MFMA -- fp32 16x16x4 (the Winograd kernel's
instruction) against fp16 32x32x16, one and two waves per SIMD.  32 cycles = the matrix pipe's own time per instruction."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
import diaglib  # noqa: E402  (libscipnp_diag.so: the laboratory entries)
lib = diaglib.load()
iters = 4000
names = {1: 'v_mfma_f32_16x16x4_f32 + v_add_f32   ', 2: 'v_mfma_f32_16x16x4_f32 + v_pk_add_f32', 3: 'v_mfma_f32_16x16x4_f32 + ds_read_b128',
         0: 'v_mfma_f32_32x32x16_f16 + v_add_f32  '}
for f32 in (1, 2, 3, 0):
    for wps in (1, 2):
        blocks = 256 * wps
        out = torch.empty(blocks * 256, device='cuda')
        cyc = torch.zeros(blocks * 4, dtype=torch.int64, device='cuda')
        row = []
        for nv in (0, 1, 2, 4, 6, 8):
            for _ in range(2):
                _lib.check(lib.scipnp_bench_mfma_valu(C.c_void_p(out.data_ptr()), C.c_void_p(cyc.data_ptr()), blocks, iters, nv, f32,
                                                      _lib.stream_ptr()), 'bench')
            torch.cuda.synchronize()
            c = cyc.cpu().numpy().astype(np.float64) / (iters * 16)
            row.append(f'NV={nv}: {np.median(c):6.1f}')
        print(names[f32], f'{wps} wave(s)/SIMD, cycles per MFMA of one wave:', '  '.join(row))

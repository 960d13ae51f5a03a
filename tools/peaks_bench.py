#!/usr/bin/env python3
"""GPU box: MEASURED ceilings of this MI355X for the rooflines -- matrix-pipe rate of a register-resident MFMA loop on
random operands (what the chip sustains under its own power management, no memory traffic at all) and HBM stream rates
(writes profiles-style JSON to stdout)."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import diaglib  # noqa: E402  (libscipnp_diag.so: the laboratory entries)
lib = diaglib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return sorted(ts)[len(ts) // 2]


res = {}
for waves_per_simd in (1, 2):
    blocks = 256 * waves_per_simd
    out = torch.empty(blocks * 256, device='cuda')
    for mode, name, per in ((0, 'f16_32x32x16', 4 * 32768), (1, 'f16_16x16x32', 8 * 16384), (2, 'f32_32x32x2', 4 * 4096)):
        iters = 60000 if mode < 2 else 30000
        fn = lambda: _lib.check(lib.scipnp_bench_mfma(p(out), blocks, iters, mode, st()), 'mfma')  # noqa: E731
        t = timed(fn)
        res[f'mfma_{name}_{waves_per_simd}wave_per_simd_TFLOPs'] = blocks * 4 * iters * per / t / 1e12
n = 1 << 30
a = torch.rand(n // 4, device='cuda').repeat(4) if False else torch.rand(n, device='cuda')
b = torch.empty_like(a)
blocks = 256 * 16
sink = torch.empty(blocks * 256, device='cuda')
t = timed(lambda: _lib.check(lib.scipnp_bench_stream(p(a), None, n, 0, blocks, p(sink), st()), 'stream'))
res['hbm_read_GBs'] = 4 * n / t / 1e9
t = timed(lambda: _lib.check(lib.scipnp_bench_stream(p(a), p(b), n, 1, blocks, None, st()), 'stream'))
res['hbm_copy_GBs_read_plus_write'] = 8 * n / t / 1e9
t = timed(lambda: _lib.check(lib.scipnp_bench_stream(p(a), p(b), n, 2, blocks, None, st()), 'stream'))
res['hbm_write_GBs'] = 4 * n / t / 1e9
t = timed(lambda: _lib.check(lib.scipnp_bench_stream(p(a), p(b), n, 3, blocks, None, st()), 'stream'))
res['hbm_write_nt_GBs'] = 4 * n / t / 1e9
res['device'] = torch.cuda.get_device_name(0)
print(json.dumps(res, indent=1))

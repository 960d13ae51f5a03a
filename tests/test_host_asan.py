"""CPU: the pure-host sources of the library (csrc/core.hip, host_rng.hip, host_pack.hip -- the error string, NumPy's legacy
Gaussian stream, the host-side weight packers) built with the HOST compiler under AddressSanitizer + UndefinedBehaviorSanitizer
(`make -C adaptivepnp_sci_amd/csrc asan` -> libscipnp_host_asan.so; SURVEY section 5: the CPU-side sanitizer build) and driven
through the same ctypes calls the package makes, in a child process with the ASan runtime preloaded.  Any out-of-bounds
access / use of undefined behaviour aborts the child (ASan report, or a trap for UBSan), which fails the test.
GPU AddressSanitizer is not available on the pool; the device code is covered by the parity tests instead."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, 'adaptivepnp_sci_amd', 'csrc')
ASAN_LIB = os.path.join(ROOT, 'adaptivepnp_sci_amd', 'libscipnp_host_asan.so')
HOSTCXX = os.environ.get('HOSTCXX', '/opt/rocm/lib/llvm/bin/clang++')

CHILD = r'''
import ctypes as C, sys
import numpy as np
lib = C.CDLL(sys.argv[1])
lib.scipnp_last_error.restype = C.c_char_p
lib.scipnp_version.restype = C.c_char_p
lib.scipnp_conv3x3_packed_floats.restype = C.c_size_t
lib.scipnp_conv3x3_split_packed_bytes.restype = C.c_size_t
lib.scipnp_host_legacy_normal.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_double,
                                          C.c_double, C.c_void_p, C.c_size_t]
assert b'scipnp' in lib.scipnp_version()

# ---- NumPy's legacy Gaussian stream: values and final state, odd and even draw counts, a refill boundary (624 words)
def legacy_normal(loc, scale, n):
    st = np.random.get_state()
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos, has, cached = C.c_int(int(st[2])), C.c_int(int(st[3])), C.c_double(float(st[4]))
    out = np.empty(n, np.float64)
    rc = lib.scipnp_host_legacy_normal(C.c_void_p(key.ctypes.data), C.byref(pos), C.byref(has), C.byref(cached), loc, scale,
                                       C.c_void_p(out.ctypes.data), n)
    assert rc == 0, lib.scipnp_last_error()
    np.random.set_state(('MT19937', key, pos.value, has.value, cached.value))
    return out
for seed in (0, 42, 987654321):
    for n in (0, 1, 2, 7, 311, 312, 313, 624, 625, 1249, 100003):
        np.random.seed(seed); ref = np.random.normal(0.25, 5 / 255, n); nxt = np.random.random(3)
        np.random.seed(seed); got = legacy_normal(0.25, 5 / 255, n)
        assert np.array_equal(got, ref) and np.array_equal(np.random.random(3), nxt), (seed, n)
# argument errors: null pointers, a position outside the state
assert lib.scipnp_host_legacy_normal(None, None, None, None, 0.0, 1.0, None, 4) != 0 and lib.scipnp_last_error()

# ---- host weight packers: fp32 and split-fp16 layouts, ragged real / padded channel counts, BatchNorm folding, range error
rng = np.random.default_rng(1)
for co_r, ci_r, ci, co in ((12, 13, 16, 16), (3, 32, 32, 8), (96, 96, 96, 96), (1, 1, 8, 8), (30, 4, 8, 32)):
    w = np.ascontiguousarray(rng.normal(size=(co_r, ci_r, 3, 3)).astype(np.float32))
    b = rng.normal(size=co_r).astype(np.float32); sc = rng.uniform(0.5, 1.5, co_r).astype(np.float32); sh = rng.normal(size=co_r).astype(np.float32)
    P = lambda a: C.c_void_p(0 if a is None else a.ctypes.data)
    nfl = lib.scipnp_conv3x3_packed_floats(ci, co)
    coP = (co + 31) // 32 * 32
    assert nfl == (ci // 8) * 9 * coP * 8 + coP
    for bias, s_, t_ in ((b, sc, sh), (None, None, None), (b, None, None)):
        packed = np.full(nfl, np.nan, np.float32)                        # exactly the queried size: an overrun is an ASan report
        assert lib.scipnp_pack_conv3x3_weights(P(w), P(bias), P(s_), P(t_), ci_r, co_r, ci, co, P(packed)) == 0
        body = packed[:-coP].reshape(ci // 8, 9, coP, 8)
        exp = w * (s_[:, None, None, None] if s_ is not None else 1)
        assert np.array_equal(body[:, :, :co_r, :].transpose(2, 0, 3, 1).reshape(co_r, ci, 9)[:, :ci_r], exp.reshape(co_r, ci_r, 9))
        assert np.isfinite(packed).all()
        nby = lib.scipnp_conv3x3_split_packed_bytes(ci, co)
        assert nby == (ci // 8) * 9 * 2 * coP * 16 + coP * 4
        ps = np.full(nby, 0xff, np.uint8)
        assert lib.scipnp_pack_conv3x3_split_bn(P(w), P(bias), P(s_), P(t_), ci_r, co_r, ci, co, P(ps)) == 0
        h = ps[:nby - coP * 4].view(np.float16).reshape(ci // 8, 9, 2, coP, 8)
        rec = h[:, :, 0].astype(np.float32) + h[:, :, 1].astype(np.float32) / 2048.0   # hi + lo' 2^-11 carries >= 21 bits
        got = rec[:, :, :co_r, :].transpose(2, 0, 3, 1).reshape(co_r, ci, 9)[:, :ci_r]
        assert np.abs(got - exp.reshape(co_r, ci_r, 9)).max() <= 2.0 ** -20 * np.abs(exp).max()
    assert lib.scipnp_pack_conv3x3_split(P(w), P(b), ci_r, co_r, ci, co, P(ps)) == 0
    big = w.copy(); big.flat[0] = 100.0                                   # outside the split-fp16 weight range
    assert lib.scipnp_pack_conv3x3_split(P(big), P(b), ci_r, co_r, ci, co, P(ps)) == -1 and b'31.9' in lib.scipnp_last_error()
    assert lib.scipnp_pack_conv3x3_weights(P(w), None, None, None, ci_r + 100, co_r, ci, co, P(packed)) == -1     # Cin_real > Cin
assert lib.scipnp_conv3x3_packed_floats(13, 16) == 0 and lib.scipnp_conv3x3_split_packed_bytes(8, 0) == 0
assert lib.scipnp_pack_conv3x3_weights(None, None, None, None, 8, 8, 8, 8, None) == -1 and b'null' in lib.scipnp_last_error()
print('ASAN_CHILD_OK')
'''


def _asan_runtime():
    out = subprocess.run([HOSTCXX, '-print-file-name=libclang_rt.asan-x86_64.so'], capture_output=True, text=True)
    path = out.stdout.strip()
    if out.returncode == 0 and os.path.isabs(path) and os.path.exists(path):
        return path
    hits = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    return hits[0] if hits else None


@pytest.mark.skipif(not os.path.exists(HOSTCXX), reason='host clang++ of the ROCm toolchain not present')
def test_host_sources_under_address_and_ub_sanitizers():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip('no AddressSanitizer runtime next to the host compiler')
    r = subprocess.run(['make', '-C', CSRC, 'asan', f'HOSTCXX={HOSTCXX}'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1')
    r = subprocess.run([sys.executable, '-c', CHILD, ASAN_LIB], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ASAN_CHILD_OK' in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert 'AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-4000:]


@pytest.mark.skipif(not os.path.exists(HOSTCXX), reason='host clang++ of the ROCm toolchain not present')
def test_the_sanitizer_build_does_catch_an_overrun():
    """the harness itself: a packed buffer ONE float short must produce an AddressSanitizer report (otherwise the test above
    proves nothing)"""
    rt = _asan_runtime()
    if rt is None or not os.path.exists(ASAN_LIB):
        pytest.skip('sanitizer build not available')
    code = ('import ctypes as C, sys, numpy as np\n'
            'lib = C.CDLL(sys.argv[1]); lib.scipnp_conv3x3_packed_floats.restype = C.c_size_t\n'
            'n = lib.scipnp_conv3x3_packed_floats(8, 8)\n'
            'w = np.ones((8, 8, 3, 3), np.float32)\n'
            'buf = (C.c_float * (n - 1))()\n'
            'lib.scipnp_pack_conv3x3_weights(C.c_void_p(w.ctypes.data), None, None, None, 8, 8, 8, 8, buf)\n'
            'print("NOT_CAUGHT")\n')
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1')
    r = subprocess.run([sys.executable, '-c', code, ASAN_LIB], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'AddressSanitizer' in r.stderr and 'NOT_CAUGHT' not in r.stdout

"""CPU: the C-ABI library loads and exports every symbol include/scipnp.h declares; host-only entry points
work; device entry points fail LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope='module')
def lib():
    from adaptivepnp_sci_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def header_symbols(name='scipnp.h'):
    src = open(os.path.join(ROOT, 'include', name)).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(scipnp_\w+)\s*\(', src)))


def exported_symbols(path):
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True, check=True).stdout
    return sorted({l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith('scipnp_')})


def test_every_declared_symbol_is_exported_and_bound(lib):
    from adaptivepnp_sci_amd import _lib
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/scipnp.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in _lib.py'
    assert sorted(_lib.SIGNATURES) == names
    # ... and the product library exports NOTHING else: no micro-benchmark, no stamped / ablated kernel instantiation
    assert exported_symbols(_lib.LIB_PATH) == names


def test_the_diagnostics_are_a_separate_library(lib):
    """include/scipnp_diag.h <-> libscipnp_diag.so: peaks.hip's micro-benchmarks and the DIAG / stamped instantiations of the
    product's Winograd kernels are built, exported and bound THERE; the product header and library hold none of them.  The
    kernels that were measured and not adopted live under lab/ and are not built by the default make at all."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import diaglib
    from adaptivepnp_sci_amd import _lib
    dlib = diaglib.load()
    names = header_symbols('scipnp_diag.h')
    assert len(names) == 9 and sorted(diaglib.SIGNATURES) == names
    assert exported_symbols(diaglib.DIAG_LIB_PATH) == names
    assert not set(names) & set(header_symbols()) and not set(names) & set(_lib.SIGNATURES)
    for n in names:
        assert hasattr(dlib, n) and not hasattr(lib, n), n
    # argument errors of the diagnostic entries surface through the product library's error string
    assert dlib.scipnp_bench_mfma(C.c_void_p(256), 1, 1, 7, None) == -1 and lib.scipnp_last_error()
    mk = open(os.path.join(ROOT, 'adaptivepnp_sci_amd', 'csrc', 'Makefile')).read()
    for src in ('peaks.o', 'conv_wino4_diag.o'):                            # (diagnostic objects never enter the product link)
        hip = src.replace('.o', '.hip')
        assert src in mk.split('DIAG_OBJS =')[1].splitlines()[0] and hip not in mk.split('\nSRCS =')[1].splitlines()[0]
    for src in ('conv_winop', 'conv_wino4x', 'conv_wino4n', 'conv_wino4p'):   # the lab: own directory, own Makefile, own header
        assert src not in mk and os.path.exists(os.path.join(ROOT, 'lab', 'csrc', src + '.hip'))
        assert not os.path.exists(os.path.join(ROOT, 'adaptivepnp_sci_amd', 'csrc', src + '.hip'))


def test_identity(lib):
    assert b'scipnp' in lib.scipnp_version()
    assert lib.scipnp_arch() == b'gfx950'


def test_host_weight_packing_layout(lib):
    from adaptivepnp_sci_amd import ops
    rng = np.random.default_rng(0)
    co_r, ci_r, ci, co = 12, 13, 16, 16
    w = torch.from_numpy(rng.normal(size=(co_r, ci_r, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=co_r).astype(np.float32))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, co_r).astype(np.float32))
    sh = torch.from_numpy(rng.normal(size=co_r).astype(np.float32))
    packed = ops.pack_conv3x3(w, b, sc, sh, Cin=ci, Cout=co).numpy()
    coP = 32
    assert packed.size == lib.scipnp_conv3x3_packed_floats(ci, co) == (ci // 8) * 9 * coP * 8 + coP
    body = packed[:-coP].reshape(ci // 8, 9, coP, 8)
    for o in range(coP):
        for i in range(ci):
            exp = (w[o, i].reshape(9) * sc[o]).numpy() if (o < co_r and i < ci_r) else np.zeros(9, np.float32)
            assert np.array_equal(body[i // 8, :, o, i % 8], exp)
    assert np.array_equal(packed[-coP:][:co_r], (b * sc + sh).numpy())
    assert not packed[-coP:][co_r:].any()
    assert lib.scipnp_conv3x3_packed_floats(13, 16) == 0          # channels must be padded to multiples of 8


def test_workspace_queries(lib):
    assert lib.scipnp_tv_workspace_bytes(128, 128, 32, 5) > 2 * 2 * 32 * 128 * 128 * 4
    assert lib.scipnp_tv_workspace_bytes(0, 128, 32, 5) == 0
    assert lib.scipnp_conv3x3_wgrad_workspace_floats(96, 96, 256) == 256 * 9 * 96 * 96
    nb = C.c_int(0)
    assert lib.scipnp_sse_partials(C.c_void_p(1), C.c_void_p(1), 5000, None, C.byref(nb), None) == 0
    assert nb.value == 3
    # Winograd forms: 16 positions per (8-channel group, 32-output block) + bias; 16 positions per slab
    assert lib.scipnp_conv3x3_wino_packed_floats(96, 96) == 12 * 3 * 4096 + 96
    assert lib.scipnp_conv3x3_wino_packed_floats(96, 12) == 0 and lib.scipnp_conv3x3_wino_packed_floats(96, 16) == 12 * 4096 + 32
    assert lib.scipnp_conv3x3_wgrad_wino_workspace_floats(96, 96, 85) == 85 * 16 * 96 * 96
    assert lib.scipnp_conv3x3_wgrad_wino_workspace_floats(16, 128, 10) == 10 * 16 * 96 * 32
    assert lib.scipnp_conv3x3_wgrad_wino_workspace_floats(0, 96, 85) == 0


def test_f32_conv_form_switch(monkeypatch):
    """SCIPNP_F32_CONV and the size bounds of the Winograd kernels (32-bit buffer offsets)"""
    from adaptivepnp_sci_amd import nets
    monkeypatch.delenv('SCIPNP_F32_CONV', raising=False)
    assert nets.f32_conv_form() == 'winograd' and nets.f32_conv_form(256, 256) == 'winograd'
    assert nets.f32_conv_form(8192, 4096) == 'direct' and nets.f32_conv_form(4096, 4096) == 'winograd'
    monkeypatch.setenv('SCIPNP_F32_CONV', 'direct')
    assert nets.f32_conv_form(256, 256) == 'direct'
    monkeypatch.setenv('SCIPNP_F32_CONV', 'fft')
    with pytest.raises(ValueError):
        nets.f32_conv_form()
    from adaptivepnp_sci_amd import finetune
    assert finetune._wino_wgrad_fits(8, 96, 96, 256, 256) and not finetune._wino_wgrad_fits(32, 96, 96, 512, 512)
    assert finetune._wino_slabs(96) == 85 and finetune._wino_slabs(16) == 255 and finetune._wino_slabs(128) == 63


def test_argument_errors_are_reported(lib):
    rc = lib.scipnp_A(None, None, None, 4, 4, 8, None)
    assert rc == -1 and b'null' in lib.scipnp_last_error()
    rc = lib.scipnp_conv3x3_c8(C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), None, 1, 13, 96, 8, 8, 0, None)
    assert rc == -1 and b'multiples of 8' in lib.scipnp_last_error()
    rc = lib.scipnp_proj_twostage(C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16),
                                  C.c_void_p(20), 4, 4, 8, 1.0, 1.0, None)
    assert rc == -2 and b'aligned' in lib.scipnp_last_error()


def test_argument_errors_of_the_wider_entries(lib):
    """validation happens before any launch, so the error behaviour of every family of entries is checkable without a GPU"""
    p = C.c_void_p(256)
    # split conv: incompatible epilogue flags
    assert lib.scipnp_conv3x3_c8s_ex(p, p, p, None, None, 1, 96, 96, 8, 8, 16, None) == -1      # mask flag without mask
    assert b'mask' in lib.scipnp_last_error()
    assert lib.scipnp_conv3x3_c8s_ex(p, p, p, None, None, 1, 96, 96, 8, 8, 4 | 8, None) == -1   # stride 2 + pixel shuffle
    assert lib.scipnp_conv3x3_c8s_ex(p, p, p, None, None, 1, 96, 80, 8, 8, 8, None) == -1       # shuffle needs Cout % 32 == 0
    assert lib.scipnp_conv3x3_c8s_ex(p, p, p, p, None, 1, 96, 96, 8, 8, 2 | 8, None) == -1      # residual + fp32 shuffle store
    # iteration-level entries: null block / null members
    assert lib.scipnp_twostage_ffdnet_iterate(None, None, None) == -1 and b'null' in lib.scipnp_last_error()
    from adaptivepnp_sci_amd import _lib
    tv = _lib.AdmmTvArgs()
    assert lib.scipnp_admm_tv_iterate(C.byref(tv), None, None) == -1 and b'null' in lib.scipnp_last_error()
    # DDnet glue, metrics, weight gradients
    assert lib.scipnp_ddnet_gather(p, p, None, p, None, 4, 2, 8, 8, None) == -1 and b'1, 3 or 4' in lib.scipnp_last_error()
    nb = C.c_int(0)
    assert lib.scipnp_frame_metrics(p, p, p, 2, 2, 8, 7, 1.0, C.byref(nb), None) == -1          # 4x4 image < 7x7 window
    assert b'win_size' in lib.scipnp_last_error()
    assert lib.scipnp_frame_metrics(None, None, None, 32, 32, 8, 7, 1.0, C.byref(nb), None) == 0 and nb.value == 16   # size query
    assert lib.scipnp_conv3x3_wgrad_split(p, p, p, p, 0, 1, 96, 96, 96, 96, 8, 8, 1.0, None) == -1   # nslab = 0
    assert lib.scipnp_conv3x3_wgrad_split(C.c_void_p(260), p, p, p, 4, 1, 96, 96, 96, 96, 8, 8, 1.0, None) == -2


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_no_cpu_fallback():
    from adaptivepnp_sci_amd import _lib, synth, twoStageAdmm_denoise_bayer
    y, Phi, _ = synth.make_problem(16, 16, 8, seed=0)
    with pytest.raises(_lib.ScipnpError, match='no CPU fallback'):
        twoStageAdmm_denoise_bayer(y, Phi, denoiser='tv', sigma=[0], iter_max=[1])


def test_signatures_mirror_the_reference():
    import inspect
    from adaptivepnp_sci_amd import admm_denoise_bayer_demosaic_pre, twoStageAdmm_denoise_bayer
    two = list(inspect.signature(twoStageAdmm_denoise_bayer).parameters)
    assert two == ['y_bayer', 'Phi_bayer', '_lambda', 'gamma', 'denoiser', 'iter_max', 'noise_estimate', 'sigma',
                   'x0_bayer', 'X_orig', 'model_denoise', 'model_demosaic', 'show_iqa', 'demosaic_method', 'lr_',
                   'inital_iter', 'interval_iter', 'logf', 'useGPU', 'update_', 'update_per_iter',
                   'close_form_demosaic', 'large', 'update_times', 'args']
    one = list(inspect.signature(admm_denoise_bayer_demosaic_pre).parameters)
    assert one == ['y_bayer', 'Phi_bayer', '_lambda', 'gamma', 'denoiser', 'iter_max', 'noise_estimate', 'sigma',
                   'x0_bayer', 'X_orig', 'model', 'show_iqa', 'demosaic_method', 'lr_', 'inital_iter', 'interval_iter',
                   'logf', 'useGPU', 'device', 'update_', 'update_per_iter']
    d = inspect.signature(twoStageAdmm_denoise_bayer).parameters
    assert d['iter_max'].default == 50 and d['lr_'].default == 1e-6 and d['interval_iter'].default == 5


def test_host_thread_cap(monkeypatch):
    """the package lowers (never raises) PyTorch's intra-op pool to the CPUs the container may use -- affinity mask and
    cgroup quota, shared by the ranks of a node -- and SCIPNP_KEEP_TORCH_THREADS opts out"""
    import torch
    from adaptivepnp_sci_amd import _lib
    n = _lib.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    before = torch.get_num_threads()
    try:
        monkeypatch.setenv('SCIPNP_KEEP_TORCH_THREADS', '1')
        torch.set_num_threads(before)
        _lib.cap_host_threads()
        assert torch.get_num_threads() == before
        monkeypatch.delenv('SCIPNP_KEEP_TORCH_THREADS')
        monkeypatch.setenv('LOCAL_WORLD_SIZE', str(4 * n))           # more ranks than CPUs: one thread each
        _lib.cap_host_threads()
        assert torch.get_num_threads() == 1
        torch.set_num_threads(1)
        monkeypatch.setenv('LOCAL_WORLD_SIZE', '1')
        _lib.cap_host_threads()
        assert torch.get_num_threads() == 1                          # never raised
    finally:
        torch.set_num_threads(before)


def test_host_flat_matches_torch_cat():
    import numpy as np
    import torch
    from adaptivepnp_sci_amd import ops
    ts = [torch.randn(3, 4), torch.randn(5).double(), torch.randn(6, 2).t(), torch.nn.Parameter(torch.randn(2, 2, 3, 3))]
    ref = torch.cat([t.detach().reshape(-1).float() for t in ts]).numpy()
    got = ops.host_flat(ts)
    assert got.dtype == np.float32 and np.array_equal(got, ref)
    assert ops.host_flat([]).shape == (0,)


def test_bench_cli_contract():
    """the driver calls `python bench.py --gpus N --steps K --warmup W`: the flags must exist (no GPU needed to parse them)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--help'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for flag in ('--gpus', '--steps', '--warmup'):
        assert flag in r.stdout

"""The C ABI from a plain C host (no Python, no PyTorch on the calling side): build examples/host_c with gcc and run it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_plain_c_host_projects_through_the_abi(tmp_path):
    gcc = shutil.which('gcc') or 'gcc'
    libdir = os.path.join(ROOT, 'adaptivepnp_sci_amd')
    exe = str(tmp_path / 'gap_projection_host')
    subprocess.run([gcc, '-std=c11', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'examples', 'host_c', 'gap_projection_host.c'), '-L', libdir, '-lscipnp',
                    '-L/opt/rocm/lib', '-lamdhip64', '-lm', f'-Wl,-rpath,{libdir}', '-Wl,-rpath,/opt/rocm/lib', '-o', exe],
                   check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'projection rel-L2' in r.stdout and 'misaligned call -> -2' in r.stdout


def _build(src, exe):
    gcc = shutil.which('gcc') or 'gcc'
    libdir = os.path.join(ROOT, 'adaptivepnp_sci_amd')
    subprocess.run([gcc, '-std=c11', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I', os.path.join(ROOT, 'include'), src,
                    '-L', libdir, '-lscipnp', '-L/opt/rocm/lib', '-lamdhip64', '-lm', '-lpthread', f'-Wl,-rpath,{libdir}',
                    '-Wl,-rpath,/opt/rocm/lib', '-o', exe], check=True, timeout=300)


def _write_problem(path, H, W, B, tv_iters, iters, sigma, seed):
    import numpy as np
    from adaptivepnp_sci_amd import synth
    y, Phi, _orig = synth.make_problem(H, W, B, seed=seed)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'ffdnet_color_weights.npz'))
    with open(path, 'wb') as f:
        np.array([H, W, B, 12, tv_iters, iters], np.int32).tofile(f)
        np.array([sigma], np.float32).tofile(f)
        y.astype(np.float32).tofile(f)
        np.ascontiguousarray(Phi, np.float32).tofile(f)
        for l in range(12):
            w, b = g[f'model.{2 * l}.weight'], g[f'model.{2 * l}.bias']
            np.array([w.shape[0], w.shape[1]], np.int32).tofile(f)
            np.ascontiguousarray(w, np.float32).tofile(f)
            np.ascontiguousarray(b, np.float32).tofile(f)
    return y, Phi, g


@pytest.mark.gpu
def test_two_concurrent_c_solves_own_their_streams_and_overflow_words(tmp_path):
    """examples/host_c/two_solves_host.c: two host threads reconstruct at the same time through the iteration-level ABI, each
    with its own stream, side stream + events (scipnp_twostage_ffdnet_args.side_*: the library creates none) and range-guard
    word.  Solve B is driven out of fp16 range: its word is set, solve A's and the process-wide word are not (the program's
    exit code), and solve A's mosaic is bit-identical to the same solve run alone on one stream."""
    import numpy as np
    H, W, B = 96, 128, 8
    blob = str(tmp_path / 'problem.bin')
    _write_problem(blob, H, W, B, 6, 3, np.float32(25 / 255), seed=21)
    one, two = str(tmp_path / 'pnp_host'), str(tmp_path / 'two_host')
    _build(os.path.join(ROOT, 'examples', 'host_c', 'pnp_admm_ffdnet_host.c'), one)
    _build(os.path.join(ROOT, 'examples', 'host_c', 'two_solves_host.c'), two)
    alone, alone2s = str(tmp_path / 'alone.bin'), str(tmp_path / 'alone2s.bin')
    r = subprocess.run([one, blob, alone], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([one, blob, alone2s, '2s'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'two streams' in r.stdout, r.stdout + r.stderr
    ref = np.fromfile(alone, np.float32)
    assert np.array_equal(np.fromfile(alone2s, np.float32), ref)          # the caller's side stream changes no bit
    out_a, out_b = str(tmp_path / 'a.bin'), str(tmp_path / 'b.bin')
    r = subprocess.run([two, blob, out_a, out_b], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'solve A overflow 0, solve B overflow 1, process-wide word 0' in r.stdout
    assert np.array_equal(np.fromfile(out_a, np.float32), ref)


def test_iterate_entries_refuse_a_block_of_another_size():
    """scipnp_*_iterate check struct_size (the blocks changed layout between library versions): a host compiled against
    another header gets SCIPNP_EINVAL and a message instead of misread fields"""
    import ctypes as C
    from adaptivepnp_sci_amd import _lib
    lib = _lib.load()
    tv = _lib.AdmmTvArgs()
    assert tv.struct_size == C.sizeof(_lib.AdmmTvArgs)
    tv.struct_size -= 8
    assert lib.scipnp_admm_tv_iterate(C.byref(tv), None, None) == -1
    assert 'struct_size' in lib.scipnp_last_error().decode()
    a = _lib.TwoStageFfdnetArgs()
    assert a.struct_size == C.sizeof(_lib.TwoStageFfdnetArgs)
    a.struct_size = 0
    assert lib.scipnp_twostage_ffdnet_iterate(C.byref(a), None, None) == -1
    assert 'struct_size' in lib.scipnp_last_error().decode()
    # a block that NAMES an arithmetic its pointers do not provide is refused (conv_form mirrors config.Config)
    a = _lib.TwoStageFfdnetArgs(conv_form=3)
    assert lib.scipnp_twostage_ffdnet_iterate(C.byref(a), None, None) == -1
    assert 'conv_form' in lib.scipnp_last_error().decode()
    a.conv_form = 7
    assert lib.scipnp_twostage_ffdnet_iterate(C.byref(a), None, None) == -1 and 'conv_form' in lib.scipnp_last_error().decode()
    # a side stream without the caller's two events is refused before anything is launched
    p = C.c_void_p(256)
    ptrs = (C.c_void_p * 12)(*[256] * 12)
    assert lib.scipnp_ffdnet_forward_c8s_2s(p, p, ptrs, 12, 96, p, p, 8, 16, 16, None, C.c_void_p(512), None, None) == -1
    assert 'events' in lib.scipnp_last_error().decode()


@pytest.mark.gpu
def test_plain_c_host_reconstructs_like_the_python_solver(tmp_path):
    """examples/host_c/pnp_admm_ffdnet_host.c: ADMM-TV warm start + two-stage ADMM/FFDNet entirely from C through the
    iteration-level ABI entries; the mosaic must equal the Python drop-in solver's bit for bit (same kernels, same order)"""
    import io
    import numpy as np
    from adaptivepnp_sci_amd import admm_denoise_bayer_demosaic_pre, synth, twoStageAdmm_denoise_bayer
    from adaptivepnp_sci_amd.nets import FFDNet
    import torch
    H, W, B, tv_iters, iters, sigma = 96, 128, 8, 6, 3, np.float32(25 / 255)
    y, Phi, _orig = synth.make_problem(H, W, B, seed=13)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'ffdnet_color_weights.npz'))
    blob = str(tmp_path / 'problem.bin')
    with open(blob, 'wb') as f:
        np.array([H, W, B, 12, tv_iters, iters], np.int32).tofile(f)
        np.array([sigma], np.float32).tofile(f)
        y.astype(np.float32).tofile(f)
        np.ascontiguousarray(Phi, np.float32).tofile(f)
        for l in range(12):
            w, b = g[f'model.{2 * l}.weight'], g[f'model.{2 * l}.bias']
            np.array([w.shape[0], w.shape[1]], np.int32).tofile(f)
            np.ascontiguousarray(w, np.float32).tofile(f)
            np.ascontiguousarray(b, np.float32).tofile(f)
    exe, out = str(tmp_path / 'pnp_host'), str(tmp_path / 'out.bin')
    _build(os.path.join(ROOT, 'examples', 'host_c', 'pnp_admm_ffdnet_host.c'), exe)
    r = subprocess.run([exe, blob, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, np.float32).reshape(H, W, B)
    warm = admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [tv_iters], False, [0], logf=io.StringIO())[0]
    net = FFDNet()
    net.load_state_dict({k: torch.from_numpy(g[k]) for k in g.files})
    os.environ['SCIPNP_CONV_PRECISION'] = 'f16x3'          # the C host's first mode is the (opt-in) split-fp16 path
    try:
        ref = twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'ffdnet_color', [iters], False, [float(sigma)], x0_bayer=warm,
                                         model_denoise=net, logf=io.StringIO())[1]
    finally:
        os.environ.pop('SCIPNP_CONV_PRECISION', None)
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())
    # the same from C in fp32 arithmetic (Winograd kernels; scipnp_twostage_ffdnet_args.packed_wino / net_in_c8)
    out32 = str(tmp_path / 'out32.bin')
    r = subprocess.run([exe, blob, out32, 'f32'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'fp32 Winograd' in r.stdout, r.stdout + r.stderr
    got32 = np.fromfile(out32, np.float32).reshape(H, W, B)
    os.environ['SCIPNP_CONV_PRECISION'] = 'f32'
    try:
        ref32 = twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'ffdnet_color', [iters], False, [float(sigma)], x0_bayer=warm,
                                           model_denoise=net, logf=io.StringIO())[1]
    finally:
        os.environ.pop('SCIPNP_CONV_PRECISION', None)
    assert np.array_equal(got32, ref32), float(np.abs(got32 - ref32).max())
    assert not np.array_equal(got32, got) and float(np.abs(got32 - got).max()) < 1e-4


@pytest.mark.gpu
def test_the_ctypes_stub_of_integration_md_runs_as_written():
    """INTEGRATION.md section 2 shows the binding a maintainer of the reference would add to utilspy.py and to the projection
    loop: execute that code block verbatim (only the library path is made absolute) against the reference's own
    expressions (oracle/sci_ops.py) -- bit-exact, including the in-place call where xall aliases theta_all"""
    import re
    import numpy as np
    import torch
    from oracle import sci_ops as OO
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = re.search(r"```python\n(import ctypes, torch\n.*?)```", text, re.S).group(1)
    block = block.replace("'libscipnp.so'", repr(os.path.join(ROOT, 'adaptivepnp_sci_amd', 'libscipnp.so')))
    ns = {}
    exec(compile(block, 'INTEGRATION.md', 'exec'), ns)
    rng = np.random.default_rng(12)
    M, N, B = 12, 20, 8
    T = lambda a: torch.from_numpy(a)               # noqa: E731
    theta, b = rng.random((M, N, B, 4), np.float32), (0.2 * rng.standard_normal((M, N, B, 4))).astype(np.float32)
    Phi = (rng.random((M, N, B, 4)) < 0.5).astype(np.float32)
    y = (rng.random((M, N, 4)) * B / 2).astype(np.float32)
    Ps = Phi.sum(2)
    Ps[Ps == 0] = 1
    dev = lambda a: T(a).cuda()                     # noqa: E731
    A_ref = torch.stack([OO.forward_A(T(theta)[..., ib], T(Phi)[..., ib]) for ib in range(4)], -1)
    assert torch.equal(ns['A_all'](dev(theta), dev(Phi)).cpu(), A_ref)
    ref = OO.project_two_stage(T(theta), T(b), T(Phi), T(y), T(Ps), 0.55, 1.0)
    th = dev(theta)
    ns['project_all'](th, dev(b), dev(Phi), dev(y), dev(Ps), 0.55, 1.0, th)         # in place, as at k = 0
    assert torch.equal(th.cpu(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,U', [((64, 64, 8), 3), ((256, 256, 8), 8)])
def test_plain_c_host_unit_batch_equals_unit_after_unit(tmp_path, shape, U):
    """examples/host_c/tv_units_host.c: the unit axis of the C ABI (scipnp_admm_tv_args.units, scipnp_pm_setup_units) from a plain C
    host -- U ADMM-TV problems stepped by ONE call per iteration on the [B][U][4][M][N] layout against U calls on per-unit
    states: bit-identical (checked by the host itself), and equal to the Python solver's result"""
    import io
    import numpy as np
    from adaptivepnp_sci_amd import admm_denoise_bayer_demosaic_pre, synth
    H, W, B = shape
    iters = 7
    blob, out = str(tmp_path / 'units.bin'), str(tmp_path / 'units_out.bin')
    pr = [synth.make_problem(H, W, B, seed=40 + u) for u in range(U)]
    with open(blob, 'wb') as f:
        np.array([H, W, B, U, iters], np.int32).tofile(f)
        for y, Phi, _o in pr:
            y.astype(np.float32).tofile(f)
            np.ascontiguousarray(Phi, np.float32).tofile(f)
    exe = str(tmp_path / 'tv_units_host')
    _build(os.path.join(ROOT, 'examples', 'host_c', 'tv_units_host.c'), exe)
    r = subprocess.run([exe, blob, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'bit-identical' in r.stdout, r.stdout + r.stderr
    got = np.fromfile(out, np.float32).reshape(U, H, W, B)
    for u, (y, Phi, _o) in enumerate(pr):
        ref = admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [iters], False, [0], logf=io.StringIO())[0]
        assert np.array_equal(got[u], ref), u

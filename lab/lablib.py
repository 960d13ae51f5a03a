"""ctypes binding of lab/libscipnp_lab.so (lab/scipnp_lab.h): the kernels that were built, measured and not adopted -- the
persistent F(2x2) kernel, the three-waves-per-SIMD, 16-channel-workgroup and producer / consumer F(4x4) kernels.  `make -C lab`
builds the library; only lab/test_lab_kernels.py and lab/probes/* use this module."""
import ctypes as C
import os
import sys

LAB = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(LAB)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from adaptivepnp_sci_amd import _lib  # noqa: E402

LAB_LIB_PATH = os.environ.get('SCIPNP_LAB_LIB', os.path.join(LAB, 'libscipnp_lab.so'))
_vp, _int, _sz = C.c_void_p, C.c_int, C.c_size_t

# name -> (restype, argtypes); every symbol lab/scipnp_lab.h declares
SIGNATURES = {
    'scipnp_conv3x3_c8w6': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_wino4n_packed_floats': (_sz, [_int, _int]),
    'scipnp_repack_wino4n': (_int, [_vp, _vp, _int, _int, _vp]),
    'scipnp_conv3x3_c8wn': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_c8wn_stamped': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _vp]),
    'scipnp_conv3x3_c8wn_diag': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_c8wp': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_c8wp_stamped': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _vp]),
    'scipnp_conv3x3_c8w6_stamped': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _vp]),
    'scipnp_conv3x3_c8w6_diag': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_c8p_supported': (_int, [_int, _int]),
    'scipnp_conv3x3_winop_packed_floats': (_sz, [_int, _int]),
    'scipnp_pack_conv3x3_winop': (_int, [_vp, _vp, _int, _int, _vp]),
    'scipnp_conv3x3_c8p': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_c8p_diag': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp]),
}

_lab = None


def load():
    """libscipnp_lab.so with every declared symbol bound (loads the product library first: the two share the error string)"""
    global _lab
    if _lab is not None:
        return _lab
    _lib.load()
    if not os.path.exists(LAB_LIB_PATH):
        raise _lib.ScipnpError(f'{LAB_LIB_PATH} not found: `make -C lab` builds it')
    lib = C.CDLL(LAB_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lab = lib
    return lib


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def pack_winop(packed_f32, Cin, Cout):
    """slab layout of the persistent F(2x2,3x3) kernel from the fp32 direct packing (device buffers); None if unsupported"""
    import torch
    lib = load()
    if not lib.scipnp_conv3x3_c8p_supported(Cin, Cout):
        return None
    p = torch.empty(lib.scipnp_conv3x3_winop_packed_floats(Cin, Cout), dtype=torch.float32, device=packed_f32.device)
    _lib.check(lib.scipnp_pack_conv3x3_winop(_p(packed_f32), _p(p), Cin, Cout, _lib.stream_ptr()), 'scipnp_pack_conv3x3_winop')
    return p


def conv3x3_c8p(x, packed_winop, Cout, relu=False, residual=None, mask_src=None, out=None, head=False):
    """the persistent 96-output-channel F(2x2,3x3) kernel (csrc/conv_winop.hip); arguments as ops.conv3x3_c8w"""
    import torch
    n, cg, h, w, _ = x.shape
    if out is None:
        out = torch.empty(n, Cout // 8, h, w, 8, device=x.device, dtype=torch.float32)
    flags = (1 if relu else 0) | (2 if residual is not None else 0) | (16 if mask_src is not None else 0) | (0x100 if head else 0)
    _lib.check(load().scipnp_conv3x3_c8p(_p(x), _p(packed_winop), _p(out), _p(residual), _p(mask_src), n, cg * 8, Cout, h, w, flags,
                                         _lib.stream_ptr()), 'scipnp_conv3x3_c8p')
    return out


def conv3x3_c8w6(x, packed_wino4, Cout, relu=False, residual=None, mask_src=None, out=None, head=False):
    """scipnp_conv3x3_c8w4's convolution on the three-waves-per-SIMD laboratory kernel (csrc/conv_wino4x.hip): same packing,
    bit-identical results, no PixelShuffle store"""
    import torch
    from adaptivepnp_sci_amd import _lib
    n, cg, h, w, _ = x.shape
    if out is None:
        out = torch.empty(n, Cout // 8, h, w, 8, device=x.device, dtype=torch.float32)
    flags = ((1 if relu else 0) | (2 if residual is not None else 0) | (16 if mask_src is not None else 0) | (0x100 if head else 0))
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    _lib.check(load().scipnp_conv3x3_c8w6(P(x), P(packed_wino4), P(out), P(residual), P(mask_src), n, cg * 8, Cout, h, w, flags,
                                          _lib.stream_ptr()), 'scipnp_conv3x3_c8w6')
    return out


def repack_wino4n(packed_wino4, Cin, Cout):
    """the F(4x4) packing of ops.pack_conv3x3_wino4 re-laid into the 16-channel slabs of scipnp_conv3x3_c8wn"""
    import torch
    from adaptivepnp_sci_amd import _lib
    out = torch.empty(load().scipnp_conv3x3_wino4n_packed_floats(Cin, Cout), dtype=torch.float32, device=packed_wino4.device)
    _lib.check(load().scipnp_repack_wino4n(C.c_void_p(packed_wino4.data_ptr()), C.c_void_p(out.data_ptr()), Cin, Cout, _lib.stream_ptr()),
               'scipnp_repack_wino4n')
    return out


def conv3x3_c8wn(x, packed_wino4n, Cout, relu=False, residual=None, mask_src=None, out=None):
    """scipnp_conv3x3_c8w4's convolution on the 16-channel-workgroup laboratory kernel (csrc/conv_wino4n.hip: three workgroups
    per CU), weights from repack_wino4n: bit-identical results, no PixelShuffle store"""
    import torch
    from adaptivepnp_sci_amd import _lib
    n, cg, h, w, _ = x.shape
    if out is None:
        out = torch.empty(n, Cout // 8, h, w, 8, device=x.device, dtype=torch.float32)
    flags = (1 if relu else 0) | (2 if residual is not None else 0) | (16 if mask_src is not None else 0)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    _lib.check(load().scipnp_conv3x3_c8wn(P(x), P(packed_wino4n), P(out), P(residual), P(mask_src), n, cg * 8, Cout, h, w, flags,
                                          _lib.stream_ptr()), 'scipnp_conv3x3_c8wn')
    return out


def conv3x3_c8wp(x, packed_wino4, Cout, relu=False, residual=None, mask_src=None, out=None):
    """scipnp_conv3x3_c8w4's convolution for Cout % 64 == 0 on the producer / consumer laboratory kernel (csrc/conv_wino4p.hip):
    same packing, bit-identical results; stride 1, plain store"""
    import torch
    from adaptivepnp_sci_amd import _lib
    n, cg, h, w, _ = x.shape
    if out is None:
        out = torch.empty(n, Cout // 8, h, w, 8, device=x.device, dtype=torch.float32)
    flags = (1 if relu else 0) | (2 if residual is not None else 0) | (16 if mask_src is not None else 0)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    _lib.check(load().scipnp_conv3x3_c8wp(P(x), P(packed_wino4), P(out), P(residual), P(mask_src), n, cg * 8, Cout, h, w, flags,
                                          _lib.stream_ptr()), 'scipnp_conv3x3_c8wp')
    return out

"""ctypes binding of libscipnp_diag.so -- the laboratory beside the product library (include/scipnp_diag.h): MFMA / HBM
micro-benchmarks, stamped and ablated instantiations of the product's Winograd kernels.  Used by bench.py (`measured_peaks`),
tools/peaks_bench.py and tools/probes/*; nothing under
adaptivepnp_sci_amd/ imports this module."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from adaptivepnp_sci_amd import _lib  # noqa: E402

DIAG_LIB_PATH = os.environ.get('SCIPNP_DIAG_LIB', os.path.join(ROOT, 'adaptivepnp_sci_amd', 'libscipnp_diag.so'))
_vp, _int, _sz = C.c_void_p, C.c_int, C.c_size_t

# name -> (restype, argtypes); every symbol include/scipnp_diag.h declares
SIGNATURES = {
    'scipnp_bench_mfma': (_int, [_vp, _int, _int, _int, _vp]),
    'scipnp_bench_mfma_valu': (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_bench_mfma_dep': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_bench_mfma_bank': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_bench_stream': (_int, [_vp, _vp, _sz, _int, _int, _vp, _vp]),
    'scipnp_conv3x3_c8w4_stamped': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _vp]),
    'scipnp_conv3x3_c8w4_diag': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_c8w_stamped': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _vp]),
    'scipnp_diag_conv3x3_wgrad_wino4': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
}

_diag = None


def load():
    """libscipnp_diag.so with every declared symbol bound (loads the product library first: the two share the error string)"""
    global _diag
    if _diag is not None:
        return _diag
    _lib.load()
    if not os.path.exists(DIAG_LIB_PATH):
        raise _lib.ScipnpError(f'{DIAG_LIB_PATH} not found: `make -C adaptivepnp_sci_amd/csrc` builds it next to libscipnp.so')
    lib = C.CDLL(DIAG_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise _lib.ScipnpError(f'libscipnp_diag.so lacks symbol {name}; rebuild the library') from e
        fn.restype = res
        fn.argtypes = args
    _diag = lib
    return lib


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())

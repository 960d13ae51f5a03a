"""ctypes binding of libscipnp.so (C ABI declared in include/scipnp.h).

The HIP library IS the product: there is no CPU or PyTorch fallback.  If the shared object is
missing or a symbol is absent this module raises at import of the first op; if no MI355X is
visible every op raises before touching memory.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SCIPNP_LIB', os.path.join(_HERE, 'libscipnp.so'))     # SCIPNP_LIB: kernel-variant experiments

_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
_i32 = C.POINTER(C.c_int32)
_vp = C.c_void_p
_int = C.c_int
_flt = C.c_float
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol include/scipnp.h declares
SIGNATURES = {
    'scipnp_version': (C.c_char_p, []),
    'scipnp_last_error': (C.c_char_p, []),
    'scipnp_arch': (C.c_char_p, []),
    'scipnp_A': (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_At': (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_phisum': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_bayer_split': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_bayer_merge': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_proj_twostage': (_int, [_vp] * 6 + [_int, _int, _int, _flt, _flt, _vp]),
    'scipnp_proj_onestage': (_int, [_vp] * 6 + [_int, _int, _int, _flt, _flt, _vp]),
    'scipnp_mosaic_to_state': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_state_to_mosaic': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_y_to_meas': (_int, [_vp, _vp, _int, _int, _vp]),
    'scipnp_rgb_to_cube': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_cube_to_rgb': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_pm_setup': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_pm_project': (_int, [_vp] * 6 + [_int, _int, _int, _int, _flt, _flt, _vp]),
    'scipnp_pm_setup_units': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_pm_dual_project_blocks': (_int, [_int, _int, _int, _int, _int]),
    'scipnp_pm_project_units': (_int, [_vp] * 6 + [_int, _int, _int, _int, _int, _flt, _flt, _vp]),
    'scipnp_tv_workspace_bytes': (_sz, [_int, _int, _int, _int]),
    'scipnp_tv_chambolle': (_int, [_vp, _vp, _flt, _vp, _int, _int, _int, _flt, _flt, _int, _vp, _sz, _vp, _vp]),
    'scipnp_tv_chambolle_ex': (_int, [_vp, _vp, _flt, _vp, _int, _int, _int, _flt, _flt, _int, _vp, _sz, _vp, _int, _vp]),
    'scipnp_pm_dual_update': (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _flt, _int, _int, _int,
                                     C.POINTER(_int), _vp]),
    'scipnp_pm_pre_denoise': (_int, [_vp] * 6 + [_int, _int, _int, _flt, _flt, _flt, _vp]),
    'scipnp_pm_pre_denoise_ex': (_int, [_vp] * 7 + [_int, _int, _int, _flt, _flt, _flt, _vp]),
    'scipnp_pm_pre_denoise_mosaic': (_int, [_vp] * 7 + [_int, _int, _int, _flt, _flt, _flt, _vp]),
    'scipnp_pm_pre_closed_form': (_int, [_vp] * 8 + [_int, _int, _int, _flt, _flt, _flt, _int, _flt, _vp]),
    'scipnp_pm_post_denoise': (_int, [_vp] * 10 + [_int, _int, _int, _int, C.POINTER(_int), _vp]),
    'scipnp_pm_post_denoise_mosaic': (_int, [_vp] * 10 + [_int, _int, _int, _int, C.POINTER(_int), _vp]),
    'scipnp_sse_partials': (_int, [_vp, _vp, _sz, _vp, C.POINTER(_int), _vp]),
    'scipnp_conv3x3_packed_floats': (_sz, [_int, _int]),
    'scipnp_pack_conv3x3_weights': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_c8': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_ffdnet_loss_grad': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, C.POINTER(_int), _vp]),
    'scipnp_conv3x3_wgrad_workspace_floats': (_sz, [_int, _int, _int]),
    'scipnp_conv3x3_wgrad': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_wgrad_wino_workspace_floats': (_sz, [_int, _int, _int]),
    'scipnp_conv3x3_wgrad_wino': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_wgrad_wino4_workspace_floats': (_sz, [_int, _int, _int]),
    'scipnp_conv3x3_wgrad_wino4': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv_bias_grad': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    'scipnp_adam_step': (_int, [_vp, _vp, _vp, _vp, _sz, C.c_double, C.c_double, C.c_double, C.c_double, _int, _vp]),
    'scipnp_pack_conv3x3_device': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_wgrad_wino4_finish_multi': (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'scipnp_conv_bias_grad_reduce_multi': (_int, [_int, _vp, _vp, _vp, _vp]),
    'scipnp_pack_conv3x3_device_multi': (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'scipnp_pack_conv3x3_wino4_multi': (_int, [_int, _vp, _vp, _vp, _vp, _vp]),
    'scipnp_conv3x3_c8_ex': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_pack_conv3x3_device_scaled': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    'scipnp_bn_fold': (_int, [_vp, _vp, _vp, _vp, _flt, _vp, _vp, _int, _vp]),
    'scipnp_bn_fold_grads': (_int, [_vp] * 6 + [_flt, _vp, _vp, _vp, _int, _int, _vp]),
    'scipnp_upsample_zero_c8': (_int, [_vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_pixel_shuffle_bwd_c8': (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_fastdvd_loss_grad': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, C.POINTER(_int), _vp]),
    'scipnp_fastdvd_finish_bwd': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_fastdvd_unpack_bwd': (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_conv3x3_split_packed_bytes': (_sz, [_int, _int]),
    'scipnp_pack_conv3x3_split': (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_pack_conv3x3_split_bn': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_c8s': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_split_overflow': (_int, [_int, C.POINTER(_int), _vp]),
    'scipnp_bind_overflow_word': (_int, [_vp]),
    'scipnp_read_overflow_word': (_int, [_vp, _int, C.POINTER(_int), _vp]),
    'scipnp_c8_to_c8s': (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_c8_add_to_c8s': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_c8s_ex': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_pack_conv3x3_split_device_scaled': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    'scipnp_upsample_zero_c8s': (_int, [_vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_pixel_shuffle_bwd_c8s': (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_pack_conv3x3_split_device': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_wgrad_split': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _flt, _vp]),
    'scipnp_conv_bias_grad_split': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _flt, _vp]),
    'scipnp_c8s_to_c8': (_int, [_vp, _vp, _flt, _int, _int, _int, _int, _vp]),
    'scipnp_c8_scale_to_c8s': (_int, [_vp, _vp, _flt, _int, _int, _int, _int, _vp]),
    'scipnp_fastdvd_pack_triplets_c8s': (_int, [_vp, _vp, _int, _int, _int, _flt, _vp]),
    'scipnp_fastdvd_pack_triplets': (_int, [_vp, _vp, _int, _int, _int, _flt, _vp]),
    'scipnp_fastdvd_pack_triplets_units': (_int, [_vp, _vp, _int, _int, _int, _int, _flt, _vp]),
    'scipnp_fastdvd_pack_triplets_c8s_units': (_int, [_vp, _vp, _int, _int, _int, _int, _flt, _vp]),
    'scipnp_fastdvd_finish': (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_frame_metrics': (_int, [_vp, _vp, _vp, _int, _int, _int, _int, C.c_double, C.POINTER(_int), _vp]),
    'scipnp_pm_pre_rgb': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _flt, _flt, _vp]),
    'scipnp_pm_ddnet_inputs': (_int, [_vp, _vp, _flt, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_ddnet_gather': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_ddnet_finish': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    'scipnp_bilinear_up2_c8': (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_ddnet_mix': (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_ddnet_loss_grad': (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, C.POINTER(_int), _vp]),
    'scipnp_ddnet_mix_bwd': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, C.POINTER(_int), _vp]),
    'scipnp_ddnet_finish_bwd': (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_ddnet_gather_bwd': (_int, [_vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, C.POINTER(_int), _vp]),
    'scipnp_bilinear_up2_bwd_c8': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_ffdnet_forward': (_int, [_vp, _vp, C.POINTER(_vp), _int, _int, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_host_legacy_normal': (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(C.c_double), C.c_double, C.c_double,
                                        _vp, _sz]),
    'scipnp_twostage_ffdnet_iterate': (_int, [_vp, C.POINTER(_int), _vp]),
    'scipnp_admm_tv_iterate': (_int, [_vp, C.POINTER(_int), _vp]),
    'scipnp_admm_tv_plane_path': (_int, [_vp]),
    'scipnp_admm_tv_flush': (_int, [_vp, C.POINTER(_int), _vp]),
    'scipnp_pm_dual_project_fits': (_int, [_int, _int, _int]),
    'scipnp_pm_dual_project': (_int, [_vp] * 9 + [_int, _int, _int, _int, _int, _flt, _flt, _vp]),
    'scipnp_conv3x3_wino_packed_floats': (_sz, [_int, _int]),
    'scipnp_pack_conv3x3_wino': (_int, [_vp, _vp, _int, _int, _vp]),
    'scipnp_conv3x3_c8w': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_conv3x3_wino4_packed_floats': (_sz, [_int, _int]),
    'scipnp_pack_conv3x3_wino4': (_int, [_vp, _vp, _int, _int, _vp]),
    'scipnp_conv3x3_c8w4': (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    'scipnp_negate': (_int, [_vp, _vp, _sz, _vp]),
    'scipnp_fastdvd_noisy_input': (_int, [_vp, _vp, _vp, _sz, _vp]),
    'scipnp_sum_rows_f64': (_int, [_vp, _vp, _int, _int, _vp]),
    'scipnp_gray_net_input': (_int, [_vp, _vp, _flt, _vp, _int, _int, _int, _vp]),
    'scipnp_gray_net_output': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_cube_to_frames': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_frames_to_cube': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_ffdnet_pack_input': (_int, [_vp, _flt, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_ffdnet_unpack_output': (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    'scipnp_cube_sum3': (_int, [_vp, _vp, _int, _int, _int, _vp]),
    'scipnp_ffdnet_forward_c8w': (_int, [_vp, _vp, C.POINTER(_vp), _int, _int, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_ffdnet_forward_c8w4': (_int, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _int, _int, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_ffdnet_forward_c8s': (_int, [_vp, _vp, C.POINTER(_vp), _int, _int, _vp, _vp, _int, _int, _int, _vp]),
    'scipnp_ffdnet_forward_c8s_2s': (_int, [_vp, _vp, C.POINTER(_vp), _int, _int, _vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp]),
}

class AdmmTvArgs(C.Structure):
    """scipnp_admm_tv_args of include/scipnp.h (struct_size is filled in by the constructor)"""
    _fields_ = [('struct_size', C.c_size_t), ('M', C.c_int), ('N', C.c_int), ('B', C.c_int), ('two_stage', C.c_int),
                ('theta', C.c_void_p), ('b', C.c_void_p), ('x', C.c_void_p), ('theta_raw', C.c_void_p),
                ('Phi', C.c_void_p), ('y', C.c_void_p), ('Phisum', C.c_void_p),
                ('c0', C.c_double), ('c1', C.c_double), ('tv_weight', C.c_float), ('tv_iters', C.c_int),
                ('tv_workspace', C.c_void_p), ('tv_workspace_bytes', C.c_size_t),
                ('orig', C.c_void_p), ('sse_part', C.c_void_p),
                ('defer_state', C.POINTER(C.c_int)), ('sse_part_prev', C.c_void_p), ('units', C.c_int)]

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class TwoStageFfdnetArgs(C.Structure):
    """scipnp_twostage_ffdnet_args of include/scipnp.h (struct_size is filled in by the constructor; set the rest by name)"""
    _fields_ = [('struct_size', C.c_size_t), ('M', C.c_int), ('N', C.c_int), ('B', C.c_int),
                ('theta', C.c_void_p), ('b', C.c_void_p), ('x', C.c_void_p),
                ('Phi', C.c_void_p), ('y', C.c_void_p), ('Phisum', C.c_void_p),
                ('w', C.c_void_p), ('x_rgb', C.c_void_p), ('out_rgb', C.c_void_p),
                ('net_in_c8s', C.c_void_p), ('net_out_c8', C.c_void_p), ('packed_split', C.c_void_p),
                ('nb', C.c_int), ('nc', C.c_int), ('scratch0', C.c_void_p), ('scratch1', C.c_void_p),
                ('orig', C.c_void_p), ('sse_part', C.c_void_p),
                ('rho', C.c_double), ('alpha', C.c_double), ('tau', C.c_double),
                ('sigma', C.c_float), ('first_iter', C.c_int),
                ('packed_wino', C.c_void_p), ('net_in_c8', C.c_void_p),
                ('overflow_word', C.c_void_p), ('side_stream', C.c_void_p),
                ('side_fork_event', C.c_void_p), ('side_join_event', C.c_void_p), ('packed_wino4', C.c_void_p),
                ('units', C.c_int), ('conv_form', C.c_int)]

    def __init__(self, **kw):
        super().__init__(C.sizeof(type(self)), **kw)


_lib = None


class ScipnpError(RuntimeError):
    pass


def load():
    """Load libscipnp.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScipnpError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C adaptivepnp_sci_amd/csrc`).  There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ScipnpError(f'libscipnp.so lacks symbol {name}; rebuild the library') from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().scipnp_last_error().decode()
        if rc in (-1, -2, -4):
            raise ValueError(f'{what}: {msg}')
        raise ScipnpError(f'{what}: {msg} (code {rc})')


_gpu_ok = False


def usable_cpus():
    """CPUs this process may really use: its affinity mask capped by the container's cgroup-v2 CPU quota"""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cap_host_threads():
    """PyTorch sizes its intra-op pool by the VISIBLE cores.  Where the container's CPU quota is smaller (the MI355X boxes:
    256 visible, 16 allowed) every CPU-side tensor op above ATen's grain size -- a deepcopy of the denoiser, a
    load_state_dict -- wakes the whole pool, its idle spinning exhausts the quota and the thread that launches the kernels
    is throttled for 50-90 ms (/sys/fs/cgroup/cpu.stat nr_throttled).  Lower (never raise) the pool to what the
    container can run; SCIPNP_KEEP_TORCH_THREADS=1 leaves it alone."""
    import os
    import torch
    if os.environ.get('SCIPNP_KEEP_TORCH_THREADS'):
        return
    # one process per GPU (torchrun exports LOCAL_WORLD_SIZE): the ranks of a node share the quota
    n = max(1, usable_cpus() // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1)))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)


def stream_ptr():
    """the current HIP stream of the current device as a C pointer argument.  torch.cuda.current_stream() builds a Stream
    object through several Python layers (~8 us); the raw-handle query PyTorch uses internally costs a fraction of that,
    which matters where a kernel launch is a ctypes call of a few microseconds"""
    import torch
    get = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if get is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(get(torch.cuda.current_device()))


def require_gpu():
    global _gpu_ok
    if _gpu_ok:
        return
    import torch
    if torch.cuda.is_available():
        _gpu_ok = True
        cap_host_threads()
        return
    if not torch.cuda.is_available():
        raise ScipnpError('no MI355X/ROCm device visible: the scipnp hot path runs only on the GPU '
                          '(no CPU fallback by design)')

"""Online finetune of the plug-in denoiser on the measurement loss, on the GPU.

ffdnet_online_finetune     <- packages/ffdnet/test_ffdnet_ipol.py:248-300 (branch `updata_=True`)
fastdvdnet_online_finetune <- packages/fastdvdnet/test_fastdvdnet.py:343-451

Both are fully hand-written: forward convolutions that keep their activations, the loss gradient, the backward-data
convolutions (forward kernel with transposed / flipped weights and a ReLU-mask epilogue), the weight / bias gradients
(MFMA GEMM with the pixels as the K dimension) and torch.optim.Adam's update are HIP kernels.  With the engine in
precision 'f16x3' (default) everything convolution-shaped runs on the split-fp16 kernels (csrc/conv_split.hip,
csrc/wgrad_split.hip), gradients travelling pre-scaled by a power of two; with 'f32' on csrc/conv.hip, csrc/finetune.hip.
A fresh Adam state is created per call, like the reference (`torch.optim.Adam(model.parameters(), lr=lr_)` at :251); the
module's parameters are updated in place and the engine continues on the device-packed updated weights.
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _lib, ops
from .nets import ffdnet_layers

F32 = torch.float32


def _s():
    return _lib.stream_ptr()


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _split_slabs(cin):
    """persistent workgroups of the split weight-gradient kernel = slabs x (Cin/32 input-channel blocks): keep their
    product at ~255 so that every CU of the MI355X gets one workgroup whatever the layer's input width"""
    from . import config
    env = config.current().wgrad_slabs
    if env:
        return int(env)
    ncib = (cin + 31) // 32
    # when the slab count is a multiple of 8 the kernel places the ncib blocks of a slab on one XCD, where they share the
    # dZ tiles in L2 (HBM traffic 888 -> 496 MB per launch at 96 channels with 80 slabs) -- taken where it costs no
    # workgroups (64 / 128 channels: 128 x 2, 64 x 4 = 256); at 96 channels 85 x 3 = 255 workgroups without the sharing
    # are 3 % faster than 80 x 3 = 240 with it (280 vs 288 us: the kernel is occupancy-bound, one 9-wave workgroup per CU)
    if ncib > 1 and 256 % ncib == 0 and (256 // ncib) % 8 == 0:
        return 256 // ncib
    return max(1, 255 // ncib)


def _wino_slabs(cin):
    """persistent workgroups of the fp32 Winograd weight-gradient kernel (csrc/wgrad_wino.hip) = slabs x (Cin/32 blocks):
    one 16-wave workgroup per CU"""
    return max(1, 255 // ((cin + 31) // 32))


def _wino4_slabs(cin, cout):
    """persistent workgroups of the F(4x4)-domain weight-gradient kernel (csrc/wgrad_wino4.hip) = slabs x (Cout/32 x Cin/32
    blocks), one 12-wave workgroup per CU"""
    return max(1, 255 // (((cin + 31) // 32) * ((cout + 31) // 32)))


def wgrad_f4_enabled():
    """the fp32 FFDNet trainer takes the weight gradients of its layers (since round 5 all of them) in the Winograd F(4x4) domain
    (csrc/wgrad_wino4.hip: 382 us against 441 us of the F(2x2) form at 96 -> 96 on 8 x 256 x 256, rounding 4.9e-6 against
    1.5e-6 of the gradient's norm, gate 1e-4: DESIGN.md section 5); SCIPNP_F32_WGRAD=f2 keeps the F(2x2) form"""
    from . import config
    return config.current().f32_wgrad == 'f4'


def _wino_wgrad_fits(n, cin, cout, h, w):
    """csrc/wgrad_wino.hip / wgrad_wino4.hip address both tensors through 32-bit buffer offsets: each must stay below 2 GiB
    (the F(4x4) kernel's input descriptor spans w + 1 pixels more, and its masked lanes carry the offset 2^31)"""
    return n * max(cin, cout) * h * w * 4 + (w + 1) * 32 <= (1 << 31)


def _upload_flat(flat_dev, srcs):
    """fill the flat device buffer from the parameter tensors: ONE host->device copy when they live on the host
    (per-tensor pageable copies cost ~0.1-0.3 ms each), per-tensor device copies otherwise"""
    if any(t.is_cuda for t in srcs):
        off = 0
        for t in srcs:
            flat_dev[off:off + t.numel()].copy_(t.reshape(-1))
            off += t.numel()
    else:
        flat_dev.copy_(torch.from_numpy(ops.host_flat(srcs)))


def _download_flat(flat_dev, dsts):
    """inverse: the updated parameters back into the module's tensors with ONE device->host copy"""
    with torch.no_grad():
        if any(t.is_cuda for t in dsts):
            off = 0
            for t in dsts:
                t.copy_(flat_dev[off:off + t.numel()].view(t.shape))
                off += t.numel()
        else:
            host = flat_dev.cpu().numpy()
            off = 0
            for t in dsts:                      # NumPy copies: see ops.host_flat on why not Tensor.copy_
                n = t.numel()
                if t.dtype == torch.float32 and t.is_contiguous():
                    np.copyto(t.detach().numpy().reshape(-1), host[off:off + n])
                else:
                    t.copy_(torch.from_numpy(host[off:off + n]).view(t.shape))
                off += n


def _carve(flat, like):
    """consecutive views of `flat` shaped like the tensors in `like` (one Adam launch covers them all)"""
    out, off = [], 0
    for t in like:
        out.append(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    return out


def _repack_wino(packed_f32, wp):
    """Winograd-domain weights of a trainer layer (ops.WinoPacked with preallocated buffers) from its fp32 direct packing"""
    if wp.f4 is not None and wp.w is None:
        # a layer latched onto the F(4x4) kernel at construction (ops.WinoPacked with w = None: ops.conv3x3_c8w can only take
        # its F(4x4) form, whatever SCIPNP_WINO_F4 says later): no F(2x2) packing exists to go stale
        ops.pack_conv3x3_wino4(packed_f32, wp.cin, wp.cout, out=wp.f4)
        return
    ops.pack_conv3x3_wino(packed_f32, wp.cin, wp.cout, out=wp.w)
    if wp.f4 is not None:
        ops.pack_conv3x3_wino4(packed_f32, wp.cin, wp.cout, out=wp.f4)


class _DeviceMean:
    """a sum left on the device by scipnp_sum_rows_f64 and its divisor: `.item()` reads it back (the only sync of a step)"""

    def __init__(self, total, count):
        self.total, self.count = total, count

    def item(self):
        return float(self.total.item()) / self.count


# Test/diagnostic hook: callable({state-dict key: gradient tensor (a device copy)}) invoked after the FIRST backward pass of
# every online-finetune event, before the Adam step -- the quantity the reference holds in `.grad` after its first
# total_loss.backward() (test_ffdnet_ipol.py:296, test_fastdvdnet.py:444).  None in production.
GRAD_HOOK = None
# diagnostic tap (tools/probes/fastdvd_grad_debug.py): callable(block name, layer index, gradient at that layer's (BN) output)
TAP = None


_WB_TLS = threading.local()          # per host thread: {(device, floats): (pinned buffer, side stream)} of deferred write-backs


class _FFDNetTrainer:
    """Device master copies of the parameters, packed forward/backward weights, Adam state and the
    activation stash of one FFDNetEngine geometry."""

    NSLAB = 256

    def __init__(self, model, eng):
        self.eng = eng
        dev = eng.device
        self.split = eng.precision == 'f16x3'
        self.model = model
        self.layers = ffdnet_layers(model)                # [(weight, bias)] tensors of the module (any device)
        srcs = [w.detach() for w, _ in self.layers] + [b.detach() for _, b in self.layers]
        total = sum(t.numel() for t in srcs)
        self._synced = None                               # (data_ptr, _version) of every parameter when device == module
        self._host = None                                 # pinned staging buffer of the deferred write-back
        self._wb = None                                   # (stream, event) of a write-back in flight
        self.losses = []                                  # (device mean, trace list) of the Adam steps not reported yet
        self._host_loss, self._loss_on_host = None, 0     # pinned copies of their sums (deferred write-back)
        # master parameters, gradients and Adam moments live in four flat buffers: one Adam launch per step
        self.flat_p, self.flat_g = torch.empty(total, dtype=F32, device=dev), torch.empty(total, dtype=F32, device=dev)
        self.flat_m, self.flat_v = torch.zeros(total, dtype=F32, device=dev), torch.zeros(total, dtype=F32, device=dev)
        views = _carve(self.flat_p, srcs)
        _upload_flat(self.flat_p, srcs)
        self._synced = self._mark()                       # (self.layers is set: the module and the device master agree)
        nl = len(self.layers)
        self.w, self.b = views[:nl], views[nl:]
        self.nb, self.nc = eng.nb, eng.nc
        B, M, N, nc = eng.B, eng.M, eng.N, eng.nc
        self.cin = [16] + [nc] * (self.nb - 1)
        self.cout = [nc] * (self.nb - 1) + [16]
        lib = _lib.load()
        self.lib = lib
        self.fwd = [torch.empty(lib.scipnp_conv3x3_packed_floats(ci, co), dtype=F32, device=dev)
                    for ci, co in zip(self.cin, self.cout)]
        self.bwd = [None] + [torch.empty(lib.scipnp_conv3x3_packed_floats(co, ci), dtype=F32, device=dev)
                             for ci, co in list(zip(self.cin, self.cout))[1:]]
        # fp32 engine in Winograd form: the stash forward and the backward-data convolutions (stride 1, full 3x3: the same
        # operator with transposed / flipped weights) run on csrc/conv_wino.hip too; weight gradients stay on the direct MFMA kernel
        self.wino = (not self.split) and getattr(eng, 'packed_wino', None) is not None
        if self.wino:
            # (the 96 -> 96 layers in both directions also as F(4x4,3x3), csrc/conv_wino4.hip: conv3x3_c8w prefers that packing)
            def wp(ci, co):
                f4 = (torch.empty(lib.scipnp_conv3x3_wino4_packed_floats(ci, co), dtype=F32, device=dev)
                      if (ops.wino_f4_enabled() and ops.wino_f4_shape(ci, co)) else None)
                # a layer with an F(4x4) packing is LATCHED onto that kernel (w = None): no uninitialised F(2x2) buffer exists that a
                # later toggle of SCIPNP_WINO_F4, rows16 = True or a C-entry caller's data_ptr() could reach
                w2 = None if f4 is not None else torch.empty(lib.scipnp_conv3x3_wino_packed_floats(ci, co), dtype=F32, device=dev)
                return ops.WinoPacked(w2, ci, co, f4)
            self.fwd_w = [wp(ci, co) for ci, co in zip(self.cin, self.cout)]
            self.bwd_w = [None] + [wp(co, ci) for ci, co in list(zip(self.cin, self.cout))[1:]]
        self.acts = [torch.empty(B, nc // 8, M, N, 8, dtype=F32, device=dev) for _ in range(self.nb - 1)]
        self.dz = [torch.empty(B, nc // 8, M, N, 8, dtype=F32, device=dev) for _ in range(2)]
        self.gout = torch.empty(B, 2, M, N, 8, dtype=F32, device=dev)
        gviews = _carve(self.flat_g, srcs)
        self.dw, self.db = gviews[:nl], gviews[nl:]
        self.slabs = [_split_slabs(ci) if self.split else self.NSLAB for ci in self.cin]
        ws = max(lib.scipnp_conv3x3_wgrad_workspace_floats(ci, co, ns) for ci, co, ns in zip(self.cin, self.cout, self.slabs))
        if self.wino:        # fp32: weight gradients in the Winograd domain too (16 positions per slab instead of 9 taps)
            ws = max([ws] + [lib.scipnp_conv3x3_wgrad_wino_workspace_floats(ci, co, _wino_slabs(ci))
                             for ci, co in zip(self.cin, self.cout)])
        self.wino4 = self.wino and wgrad_f4_enabled()
        if self.wino4:
            ws = max([ws] + [lib.scipnp_conv3x3_wgrad_wino4_workspace_floats(ci, co, _wino4_slabs(ci, co))
                             for ci, co in zip(self.cin, self.cout)])
        self.ws = torch.empty(ws, dtype=F32, device=dev)
        self.bws = torch.empty((nc // 8) * 64 * 8, dtype=F32, device=dev)
        # F(4x4)-domain weight gradients: every layer keeps its own slabs and bias partials until the end of the backward pass,
        # where ONE slab-reduction, ONE back-transform and ONE bias-reduction launch finish all of them (round 5: three dependent
        # launches of 5-9 us per layer and step were 0.45 ms of a 12 ms tile event)
        self.ws_l = self.bws_l = None
        if self.wino4 and all(_wino_wgrad_fits(B, ci, co, M, N) for ci, co in zip(self.cin, self.cout)):
            self.ws_l = [torch.empty(lib.scipnp_conv3x3_wgrad_wino4_workspace_floats(ci, co, _wino4_slabs(ci, co)), dtype=F32, device=dev)
                         for ci, co in zip(self.cin, self.cout)]
            self.bws_l = [torch.empty((co // 8) * 64 * 8, dtype=F32, device=dev) for co in self.cout]
        nb_ = C.c_int(0)
        _lib.check(lib.scipnp_ffdnet_loss_grad(None, None, None, None, None, M, N, B, C.byref(nb_), None), 'loss size')
        self.loss_part = torch.empty(nb_.value, dtype=torch.float64, device=dev)
        self.step = 0
        # split-fp16 path (engine precision 'f16x3'): forward stash, backward-data convolutions and weight gradients on the
        # fp16 MFMA with error-compensated operands; gradients travel pre-scaled by the power of two nearest 2*M*N (the
        # measurement loss carries 1/(2MN), so the scaled output gradient is O(residual)) and are un-scaled, exactly, by
        # the slab reduction of the weight / bias gradient kernels
        if self.split:
            f16 = torch.float16
            self.fwd_s = [torch.empty(lib.scipnp_conv3x3_split_packed_bytes(ci, co), dtype=torch.uint8, device=dev)
                          for ci, co in zip(self.cin, self.cout)]
            self.bwd_s = [None] + [torch.empty(lib.scipnp_conv3x3_split_packed_bytes(co, ci), dtype=torch.uint8, device=dev)
                                   for ci, co in list(zip(self.cin, self.cout))[1:]]
            self.acts_s = [a.view(f16).view(B, nc // 8, 2, M, N, 8) for a in self.acts]      # same bytes as fp32 c8
            self.dz_s = [torch.empty(B, nc // 8, 2, M, N, 8, dtype=f16, device=dev) for _ in range(2)]
            self.gout_s = torch.empty(B, 2, 2, M, N, 8, dtype=f16, device=dev)
            self.gscale = float(2.0 ** round(np.log2(2.0 * M * N)))

    def _real(self, l):
        w = self.w[l]
        return w.shape[1], w.shape[0]      # Cin_real, Cout_real

    def pack(self, final=False):
        """pack the master weights on the device; final=True (after the last Adam step) skips the backward-data packs
        and also refreshes the fp32 forward packs the engine keeps next to the split ones"""
        if self.split:
            for l in range(self.nb):
                ops.pack_conv3x3_split_device(self.w[l], self.b[l], self.fwd_s[l], self.cin[l], self.cout[l])
                if l > 0 and not final:
                    ops.pack_conv3x3_split_device(self.w[l], None, self.bwd_s[l], self.cin[l], self.cout[l], transpose=True)
                if final:
                    ci_r, co_r = self._real(l)
                    _lib.check(self.lib.scipnp_pack_conv3x3_device(_ptr(self.w[l]), _ptr(self.b[l]), _ptr(self.fwd[l]), ci_r,
                                                                   co_r, self.cin[l], self.cout[l], 0, _s()), 'pack fwd')
            return
        # every layer's packs in two launches (scipnp_pack_conv3x3_device_multi, scipnp_pack_conv3x3_wino4_multi; round 5: the 46
        # launches of 4 us each per pack() -- three packs per event -- were 1.2 ms of kernels and 1.7 ms of gaps in a 25 ms event)
        dev_jobs = []                                         # (w, bias, packed, ci_r, co_r, cin, cout, transpose)
        w4_jobs, rest = [], []                                # (packed_f32, f4, cin, cout) / layers that keep an F(2x2) packing
        for l in range(self.nb):
            ci_r, co_r = self._real(l)
            dev_jobs.append((self.w[l], self.b[l], self.fwd[l], ci_r, co_r, self.cin[l], self.cout[l], 0))
            if l > 0:
                dev_jobs.append((self.w[l], None, self.bwd[l], ci_r, co_r, self.cin[l], self.cout[l], 1))
            if self.wino:
                for pk, wp in ((self.fwd[l], self.fwd_w[l]),) + (((self.bwd[l], self.bwd_w[l]),) if (l > 0 and not final) else ()):
                    if wp.f4 is not None and wp.w is None:
                        w4_jobs.append((pk, wp.f4, wp.cin, wp.cout))
                    else:
                        rest.append((pk, wp))
        ops.pack_conv3x3_device_multi(dev_jobs)
        ops.pack_conv3x3_wino4_multi(w4_jobs)
        for pk, wp in rest:
            _repack_wino(pk, wp)

    def forward_keep(self):
        eng = self.eng
        if self.split:
            x = eng.in_c8s
            for l in range(self.nb - 1):
                ops.conv3x3_c8s(x, self.fwd_s[l], self.nc, relu=True, out=self.acts_s[l], head=(l == 0))
                x = self.acts_s[l]
            ops.conv3x3_c8s(x, self.fwd_s[-1], 16, out=eng.out_c8, f32_out=True)
            return
        x = eng.in_c8
        conv, pk = (ops.conv3x3_c8w, self.fwd_w) if self.wino else (ops.conv3x3_c8, self.fwd)
        for l in range(self.nb - 1):
            conv(x, pk[l], self.nc, relu=True, out=self.acts[l], head=(l == 0))
            x = self.acts[l]
        conv(x, pk[-1], 16, relu=False, out=eng.out_c8)

    def loss_and_grad(self, y_pm, Phi_pm):
        eng = self.eng
        nb_ = C.c_int(0)
        _lib.check(self.lib.scipnp_ffdnet_loss_grad(_ptr(eng.out_c8), _ptr(Phi_pm), _ptr(y_pm), _ptr(self.gout),
                                                    _ptr(self.loss_part), eng.M, eng.N, eng.B, C.byref(nb_), _s()),
                   'scipnp_ffdnet_loss_grad')
        return _DeviceMean(ops.sum_rows_f64(self.loss_part), 4.0 * eng.M * eng.N)     # read back by .item()

    def backward(self):
        if self.split:
            return self._backward_split()
        eng = self.eng
        B, M, N = eng.B, eng.M, eng.N
        dz = self.gout
        if self.ws_l is not None:
            return self._backward_deferred(dz)
        for l in range(self.nb - 1, -1, -1):
            a_in = eng.in_c8 if l == 0 else self.acts[l - 1]
            ci_r, co_r = self._real(l)
            # (every layer, the 16-channel head and tail included -- the kernel pads the narrow side to its 32-channel block:
            # 84 / 82 us against 106 / 118 us of the F(2x2)-domain kernel on a tile, profiles/r05zw_wgrad_narrow.txt)
            if self.wino4 and _wino_wgrad_fits(B, self.cin[l], self.cout[l], M, N):
                _lib.check(self.lib.scipnp_conv3x3_wgrad_wino4(_ptr(a_in), _ptr(dz), _ptr(self.dw[l]), _ptr(self.ws),
                                                               _wino4_slabs(self.cin[l], self.cout[l]), B, ci_r, co_r,
                                                               self.cin[l], self.cout[l], M, N, _s()), 'wgrad wino4')
            elif self.wino and _wino_wgrad_fits(B, self.cin[l], self.cout[l], M, N):
                _lib.check(self.lib.scipnp_conv3x3_wgrad_wino(_ptr(a_in), _ptr(dz), _ptr(self.dw[l]), _ptr(self.ws),
                                                              _wino_slabs(self.cin[l]), B, ci_r, co_r, self.cin[l],
                                                              self.cout[l], M, N, _s()), 'wgrad wino')
            else:
                _lib.check(self.lib.scipnp_conv3x3_wgrad(_ptr(a_in), _ptr(dz), _ptr(self.dw[l]), _ptr(self.ws), self.NSLAB, B,
                                                         ci_r, co_r, self.cin[l], self.cout[l], M, N, _s()), 'wgrad')
            _lib.check(self.lib.scipnp_conv_bias_grad(_ptr(dz), _ptr(self.db[l]), _ptr(self.bws), B, co_r, self.cout[l],
                                                      M, N, _s()), 'bgrad')
            if l > 0:
                nxt = self.dz[l & 1]
                # backward-data: conv of dZ_l with the transposed/flipped weights, masked by ReLU'(A_{l-1})
                if self.wino:
                    ops.conv3x3_c8w(dz.view(B, self.cout[l] // 8, M, N, 8), self.bwd_w[l], self.cin[l], mask_src=self.acts[l - 1],
                                    out=nxt)
                else:
                    _lib.check(self.lib.scipnp_conv3x3_c8(_ptr(dz), _ptr(self.bwd[l]), _ptr(nxt), _ptr(self.acts[l - 1]), B,
                                                          self.cout[l], self.cin[l], M, N, 16, _s()), 'backward-data conv')
                dz = nxt

    def _backward_deferred(self, dz):
        """the same backward pass with the F(4x4)-domain weight gradient of every layer: slabs and bias partials stay in the
        layer's own workspace, three multi-layer launches at the end turn them into dW / db (same kernels' arithmetic: the
        gradients are bit-identical to finishing layer by layer)"""
        eng, lib = self.eng, self.lib
        B, M, N = eng.B, eng.M, eng.N
        for l in range(self.nb - 1, -1, -1):
            a_in = eng.in_c8 if l == 0 else self.acts[l - 1]
            ci_r, co_r = self._real(l)
            _lib.check(lib.scipnp_conv3x3_wgrad_wino4(_ptr(a_in), _ptr(dz), None, _ptr(self.ws_l[l]),
                                                      _wino4_slabs(self.cin[l], self.cout[l]), B, ci_r, co_r, self.cin[l],
                                                      self.cout[l], M, N, _s()), 'wgrad wino4 slabs')
            _lib.check(lib.scipnp_conv_bias_grad(_ptr(dz), None, _ptr(self.bws_l[l]), B, co_r, self.cout[l], M, N, _s()), 'bgrad partials')
            if l > 0:
                nxt = self.dz[l & 1]
                ops.conv3x3_c8w(dz.view(B, self.cout[l] // 8, M, N, 8), self.bwd_w[l], self.cin[l], mask_src=self.acts[l - 1], out=nxt)
                dz = nxt
        n = self.nb
        P, I = C.c_void_p * n, C.c_int * n
        real = [self._real(l) for l in range(n)]
        _lib.check(lib.scipnp_conv3x3_wgrad_wino4_finish_multi(
            n, P(*[t.data_ptr() for t in self.ws_l]), P(*[t.data_ptr() for t in self.dw]),
            I(*[_wino4_slabs(ci, co) for ci, co in zip(self.cin, self.cout)]), I(*[r[0] for r in real]), I(*[r[1] for r in real]),
            I(*self.cin), I(*self.cout), _s()), 'wgrad wino4 finish')
        _lib.check(lib.scipnp_conv_bias_grad_reduce_multi(n, P(*[t.data_ptr() for t in self.bws_l]), P(*[t.data_ptr() for t in self.db]),
                                                          I(*[r[1] for r in real]), _s()), 'bgrad reduce')

    def _backward_split(self):
        eng = self.eng
        B, M, N = eng.B, eng.M, eng.N
        inv = 1.0 / self.gscale
        dz_s = ops.c8_scale_to_c8s(self.gout, self.gout_s, self.gscale)
        for l in range(self.nb - 1, -1, -1):
            a_in = eng.in_c8s if l == 0 else self.acts_s[l - 1]
            ci_r, co_r = self._real(l)
            _lib.check(self.lib.scipnp_conv3x3_wgrad_split(_ptr(a_in), _ptr(dz_s), _ptr(self.dw[l]), _ptr(self.ws), self.slabs[l],
                                                           B, ci_r, co_r, self.cin[l], self.cout[l], M, N, inv, _s()),
                       'wgrad split')
            _lib.check(self.lib.scipnp_conv_bias_grad_split(_ptr(dz_s), _ptr(self.db[l]), _ptr(self.bws), B, co_r,
                                                            self.cout[l], M, N, inv, _s()), 'bgrad split')
            if l > 0:
                nxt = self.dz_s[l & 1]
                ops.conv3x3_c8s(dz_s, self.bwd_s[l], self.cin[l], out=nxt, mask=self.acts_s[l - 1])
                dz_s = nxt

    def adam(self, lr):
        self.step += 1
        _lib.check(self.lib.scipnp_adam_step(_ptr(self.flat_p), _ptr(self.flat_g), _ptr(self.flat_m), _ptr(self.flat_v),
                                             self.flat_p.numel(), float(lr), 0.9, 0.999, 1e-8, self.step, _s()),
                   'scipnp_adam_step')

    def _params(self):
        return [w for w, _ in self.layers] + [b for _, b in self.layers]

    def _mark(self):
        return [(t.data_ptr(), t._version) for t in self._params()]

    def reuse(self, model):
        """the trainer of an earlier event of the same engine for the next one (round 5: building it -- 40 allocations, the
        parameter upload, two Adam-state fills -- kept the GPU idle for 0.7 ms per event): same module and parameter tensors ->
        fresh Adam state (the reference builds a new optimiser per event, test_ffdnet_ipol.py:286-288), and the parameters are
        uploaded again unless the module still holds exactly what this trainer wrote back (tensor identity and version
        counters).  False: build a new trainer."""
        self.finish_write_back()
        if model is not self.model:
            return False
        # the module's parameters still live where this trainer reads / writes them (storage identity: state_dict() and
        # parameters() hand out new tensor objects over the same storage)
        mine = {t.data_ptr() for t in self._params()}
        cur = [q for q in model.parameters()]
        if len(cur) != len(mine) or any(q.data_ptr() not in mine for q in cur):
            return False
        if self._synced is None or self._synced != self._mark():
            _upload_flat(self.flat_p, [t.detach() for t in self._params()])
        self.flat_m.zero_()
        self.flat_v.zero_()
        self.step = 0
        return True

    def write_back(self, defer=False):
        """the reference mutates `model` in place and returns it (test_ffdnet_ipol.py:356-357).  defer: the device -> host copy
        goes to a pinned buffer on a side stream behind the last Adam step; finish_write_back() -- called by the solver once the
        evaluation pass is enqueued, and by anything else that touches the trainer -- waits for it and fills the module's
        tensors, so the host's 0.35 ms are spent while the GPU runs the denoiser"""
        dsts = self._params()
        if not defer or any(t.is_cuda for t in dsts):
            _download_flat(self.flat_p, dsts)
            self._synced = self._mark()
            return
        # pinned staging + side stream: one per HOST THREAD, shared by the trainers that thread steps.  Looked up on every call (never
        # cached on the trainer): a trainer first stepped by one lane thread and later by another must not carry the first thread's
        # buffer along -- two threads would then fill and read one pinned buffer at the same time.
        if self._wb is not None:                 # (a pending write-back of this trainer, possibly staged in another thread's buffer)
            self.finish_write_back()
        key = (self.flat_p.device.index, self.flat_p.numel())
        pool = getattr(_WB_TLS, 'pool', None)
        if pool is None:
            pool = _WB_TLS.pool = {}
        if key not in pool:
            pool[key] = [torch.empty(self.flat_p.numel(), dtype=F32, pin_memory=True), torch.cuda.Stream(self.flat_p.device), None]
        self._pool = pool[key]
        self._host, self._wb_stream = self._pool[0], self._pool[1]
        owner = self._pool[2]                    # the staging buffer is shared within the thread: whoever used it last has to be done with it
        if owner is not None and owner is not self:
            owner.finish_write_back()
        self._pool[2] = self
        cur = torch.cuda.current_stream(self.flat_p.device)
        self._wb_stream.wait_stream(cur)
        with torch.cuda.stream(self._wb_stream):
            self._host.copy_(self.flat_p, non_blocking=True)
            if self.losses:                          # the steps' losses ride along (an .item() on the solve's stream would wait
                if self._host_loss is None or self._host_loss.numel() < len(self.losses):     # for the evaluation pass)
                    self._host_loss = torch.empty(max(8, len(self.losses)), dtype=torch.float64, pin_memory=True)
                for i, (loss, _) in enumerate(self.losses):
                    self._host_loss[i:i + 1].copy_(loss.total.reshape(1), non_blocking=True)
                self._loss_on_host = len(self.losses)
            ev = torch.cuda.Event()
            ev.record(self._wb_stream)
        self._wb = ev

    def report_losses(self):
        """the reference prints the loss tensor of every Adam step (test_ffdnet_ipol.py:298-299)"""
        pending, self.losses = self.losses, []
        on_host, self._loss_on_host = self._loss_on_host, 0
        for i, (loss, trace) in enumerate(pending):
            val = float(self._host_loss[i]) / loss.count if i < on_host else float(loss.item())
            print('loss:', val)
            if trace is not None:
                trace.append(val)

    def finish_write_back(self):
        if self._wb is not None:
            self._wb.synchronize()
        self.report_losses()
        if self._wb is None:
            return
        self._wb = None
        host = self._host.numpy()
        off = 0
        with torch.no_grad():
            for t in self._params():                  # NumPy copies: see ops.host_flat on why not Tensor.copy_
                n = t.numel()
                if t.dtype == torch.float32 and t.is_contiguous():
                    np.copyto(t.detach().numpy().reshape(-1), host[off:off + n])
                else:
                    t.copy_(torch.from_numpy(host[off:off + n]).view(t.shape))
                off += n
        self._synced = self._mark()


def ffdnet_online_finetune(model, eng, y_pm, Phi_pm, sigma, lr_, update_per_iter, logf=None, trace=None, defer_write_back=False):
    """`update_per_iter` Adam steps on the measurement loss for the input currently held in eng.in_c8
    (written by scipnp_pm_pre_denoise); y_pm [4][M][N], Phi_pm [B][4][M][N] plane-major.  Leaves the
    engine's packed weights refreshed and the module's parameters updated; the caller then runs the
    evaluation forward (reference :303-315)."""
    _lib.require_gpu()
    if update_per_iter <= 0:                 # no Adam step: nothing changes (the trainer's fp32 forward packs are only
        return model                         # filled by the last step's pack(final=True); never adopt them unfilled)
    tr = getattr(eng, '_ft_trainer', None)                # (defer_write_back: the caller calls the returned trainer's
    if tr is None or not tr.reuse(model):                 # finish_write_back() once its evaluation pass is enqueued)
        tr = eng._ft_trainer = _FFDNetTrainer(model, eng)
    tr.pack()
    for it in range(update_per_iter):
        tr.forward_keep()
        loss = tr.loss_and_grad(y_pm, Phi_pm)
        tr.backward()
        if it == 0 and GRAD_HOOK is not None:
            GRAD_HOOK({**{f'model.{2 * l}.weight': tr.dw[l].clone() for l in range(tr.nb)},
                       **{f'model.{2 * l}.bias': tr.db[l].clone() for l in range(tr.nb)}})
        tr.adam(lr_)
        tr.pack(final=(it == update_per_iter - 1))
        tr.losses.append((loss, trace))                       # read back and printed once the whole event is enqueued (a
    tr.write_back(defer=defer_write_back)                     # read-back per step stalled the GPU for 0.1 ms each)
    eng.adopt(tr.fwd, tr.fwd_s if tr.split else None,     # the engine continues on the device-packed updated weights
              packed_wino=tr.fwd_w if tr.wino else None)
    if not defer_write_back:
        tr.report_losses()
    return tr if defer_write_back else model


# ============================================================================================ FastDVDnet
class _DenBlockTrainer:
    """Master parameters, folded/packed weights, gradients and Adam state of one DenBlock (temp1 / temp2)."""

    # per layer: (input buffer key, output-gradient resolution divisor, stride2, shuffle)
    def __init__(self, sd, prefix, device, lib, split=False, wino=False):
        from .fastdvd import _LAYERS, _BN_EPS
        self.lib, self.dev, self.prefix = lib, device, prefix
        self.split = split
        self.eps = _BN_EPS
        self.spec = _LAYERS
        self.sd_keys = []
        # parameter order = Adam order: all conv weights, then (gamma, beta) of every BatchNorm; parameters, gradients and
        # Adam moments live in four flat buffers (views below), so one Adam launch per step covers the whole block
        names, srcs = [], []
        for key, bn, *_r in _LAYERS:
            names.append(f'{prefix}.{key}.weight')
        for key, bn, *_r in _LAYERS:
            if bn is not None:
                names += [f'{prefix}.{bn}.weight', f'{prefix}.{bn}.bias']
        srcs = [sd[k].detach() for k in names]
        total = sum(t.numel() for t in srcs)
        self.flat_p, self.flat_g = (torch.empty(total, dtype=F32, device=device) for _ in range(2))
        self.flat_m, self.flat_v = (torch.zeros(total, dtype=F32, device=device) for _ in range(2))
        pv, gv = _carve(self.flat_p, srcs), _carve(self.flat_g, srcs)
        _upload_flat(self.flat_p, srcs)
        nl = len(_LAYERS)
        self.W = pv[:nl]
        self.gamma, self.beta, self.mean, self.var = [], [], [], []
        self.dgamma, self.dbeta = [], []
        bns = [bn for _k, bn, *_r in _LAYERS if bn is not None]
        stats = ops.device_params([sd[f'{prefix}.{bn}.running_mean'] for bn in bns] +
                                  [sd[f'{prefix}.{bn}.running_var'] for bn in bns], torch.device(device))   # one upload
        j, jb = nl, 0
        for key, bn, cin, cout, *_ in _LAYERS:
            if bn is not None:
                self.gamma.append(pv[j]); self.beta.append(pv[j + 1])
                self.dgamma.append(gv[j]); self.dbeta.append(gv[j + 1])
                j += 2
                self.mean.append(stats[jb]); self.var.append(stats[len(bns) + jb])
                jb += 1
            else:
                self.gamma.append(None); self.beta.append(None); self.mean.append(None); self.var.append(None)
                self.dgamma.append(None); self.dbeta.append(None)
        self.params = list(zip(names, pv))
        self.grads = list(zip(names, gv))
        self._gW = gv[:nl]
        n = len(_LAYERS)
        self.scale = [None if g is None else torch.empty_like(g) for g in self.gamma]
        self.shift = [None if g is None else torch.empty_like(g) for g in self.gamma]
        self.fwd = [torch.empty(lib.scipnp_conv3x3_packed_floats(ci, co), dtype=F32, device=device)
                    for _, _, ci, co, *_ in _LAYERS]
        self.bwd = [torch.empty(lib.scipnp_conv3x3_packed_floats(co, ci), dtype=F32, device=device)
                    for _, _, ci, co, *_ in _LAYERS]
        if split:
            self.fwd_s = [torch.empty(lib.scipnp_conv3x3_split_packed_bytes(ci, co), dtype=torch.uint8, device=device)
                          for _, _, ci, co, *_ in _LAYERS]
            self.bwd_s = [torch.empty(lib.scipnp_conv3x3_split_packed_bytes(co, ci), dtype=torch.uint8, device=device)
                          for _, _, ci, co, *_ in _LAYERS]
        # fp32 in Winograd form (csrc/conv_wino.hip): the stash forward of the stride-1 layers (PixelShuffle ones included) and
        # EVERY backward-data convolution (stride-2 layers convolve a zero-upsampled gradient, PixelShuffle layers an
        # un-shuffled one: all stride-1 3x3 convolutions with the transposed weights); weight gradients stay direct
        self.fwd_w = self.bwd_w = None
        if wino and not split:
            wf = lambda a, c: torch.empty(lib.scipnp_conv3x3_wino_packed_floats(a, c), dtype=F32, device=device)  # noqa: E731
            self.fwd_w = [None if s2 else wf(ci, co) for _, _, ci, co, _r, s2, _sh in _LAYERS]
            self.bwd_w = [wf(co, ci) for _, _, ci, co, *_ in _LAYERS]
        self.dense0 = torch.zeros(90, 12, 3, 3, dtype=F32, device=device)      # block-diagonal form of the grouped conv
        self.G = [torch.empty_like(self.dense0 if i == 0 else w) for i, w in enumerate(self.W)]
        # dW[i] (i >= 1) and the grouped form of dW[0] are the flat gradient views; dW[0] itself is the dense scratch
        self.dW = [torch.empty_like(self.dense0)] + self._gW[1:]
        self.dW0_grouped = self._gW[0]
        self.sdy = [None if g is None else torch.empty_like(g) for g in self.gamma]

    def dense_w(self, i):
        if i != 0:
            return self.W[i]
        for g in range(3):                                   # data movement only: grouped -> block-diagonal
            self.dense0[g * 30:(g + 1) * 30, g * 4:(g + 1) * 4] = self.W[0][g * 30:(g + 1) * 30]
        return self.dense0

    def pack(self):
        lib = self.lib
        for i, (key, bn, cin, cout, *_r) in enumerate(self.spec):
            w = self.dense_w(i)
            co_r, ci_r = w.shape[0], w.shape[1]
            sc = sh = None
            if bn is not None:
                _lib.check(lib.scipnp_bn_fold(_ptr(self.gamma[i]), _ptr(self.beta[i]), _ptr(self.mean[i]), _ptr(self.var[i]),
                                              self.eps, _ptr(self.scale[i]), _ptr(self.shift[i]), co_r, _s()), 'bn_fold')
                sc, sh = self.scale[i], self.shift[i]
            if self.split:
                ops.pack_conv3x3_split_device(w, sh, self.fwd_s[i], cin, cout, scale=sc)
                ops.pack_conv3x3_split_device(w, None, self.bwd_s[i], cin, cout, transpose=True, scale=sc)
                continue
            _lib.check(lib.scipnp_pack_conv3x3_device_scaled(_ptr(w), _ptr(sh), _ptr(sc), _ptr(self.fwd[i]), ci_r, co_r, cin,
                                                             cout, 0, _s()), 'pack fwd')
            _lib.check(lib.scipnp_pack_conv3x3_device_scaled(_ptr(w), None, _ptr(sc), _ptr(self.bwd[i]), ci_r, co_r, cin,
                                                             cout, 1, _s()), 'pack bwd')
            if self.bwd_w is not None:
                if self.fwd_w[i] is not None:
                    ops.pack_conv3x3_wino(self.fwd[i], cin, cout, out=self.fwd_w[i])
                ops.pack_conv3x3_wino(self.bwd[i], cout, cin, out=self.bwd_w[i])

    def grads_of_layer(self, i, x_in, dy, n, h, w, ws, bws, nslab, inv_scale=1.0):
        """parameter gradients of layer i from its input activation and the gradient at its (BN) output; in split mode
        x_in / dy are c8s tensors and dy carries the gradient times 1/inv_scale"""
        lib = self.lib
        key, bn, cin, cout, *_r = self.spec[i]
        wd = self.dense_w(i)
        co_r, ci_r = wd.shape[0], wd.shape[1]
        if self.split:
            _lib.check(lib.scipnp_conv3x3_wgrad_split(_ptr(x_in), _ptr(dy), _ptr(self.G[i]), _ptr(ws), nslab, n, ci_r, co_r, cin,
                                                      cout, h, w, inv_scale, _s()), 'wgrad split')
        elif self.bwd_w is not None and _wino_wgrad_fits(n, cin, cout, h, w):     # fp32 in Winograd form (csrc/wgrad_wino.hip)
            _lib.check(lib.scipnp_conv3x3_wgrad_wino(_ptr(x_in), _ptr(dy), _ptr(self.G[i]), _ptr(ws), _wino_slabs(cin), n, ci_r,
                                                     co_r, cin, cout, h, w, _s()), 'wgrad wino')
        else:
            _lib.check(lib.scipnp_conv3x3_wgrad(_ptr(x_in), _ptr(dy), _ptr(self.G[i]), _ptr(ws), nslab, n, ci_r, co_r, cin, cout,
                                                h, w, _s()), 'wgrad')
        if bn is not None and self.split:
            _lib.check(lib.scipnp_conv_bias_grad_split(_ptr(dy), _ptr(self.sdy[i]), _ptr(bws), n, co_r, cout, h, w, inv_scale,
                                                       _s()), 'bgrad split')
        elif bn is not None:
            _lib.check(lib.scipnp_conv_bias_grad(_ptr(dy), _ptr(self.sdy[i]), _ptr(bws), n, co_r, cout, h, w, _s()), 'bgrad')
        if bn is not None:
            _lib.check(lib.scipnp_bn_fold_grads(_ptr(wd), _ptr(self.G[i]), _ptr(self.sdy[i]), _ptr(self.gamma[i]),
                                                _ptr(self.mean[i]), _ptr(self.var[i]), self.eps, _ptr(self.dW[i]),
                                                _ptr(self.dgamma[i]), _ptr(self.dbeta[i]), co_r, ci_r * 9, _s()), 'bn grads')
        else:
            self.dW[i].copy_(self.G[i])
        if i == 0:
            for g in range(3):
                self.dW0_grouped[g * 30:(g + 1) * 30] = self.dW[0][g * 30:(g + 1) * 30, g * 4:(g + 1) * 4]


class _FastDVDTrainer:
    NSLAB = 256

    def __init__(self, model, eng):
        from .fastdvd import _strip, alloc_denblock_buffers
        self.eng = eng
        self.lib = _lib.load()
        dev = eng.device
        self.model_sd = model.state_dict()
        self.prefixed = any(k.startswith('module.') for k in self.model_sd)
        sd = _strip(self.model_sd)
        self.split = getattr(eng, 'precision', 'f32') == 'f16x3'
        B, H, W = eng.B, eng.H, eng.W
        from .nets import f32_conv_form
        wino = (not self.split) and f32_conv_form(H, W) == 'winograd'
        self.blocks = {p: _DenBlockTrainer(sd, p, dev, self.lib, self.split, wino) for p in ('temp1', 'temp2')}
        if self.split:
            # forward stash, backward-data convolutions and weight gradients on the split-fp16 kernels; gradients travel
            # pre-scaled by the power of two nearest H*W/2 (the loss carries 2/(H*W)) and are un-scaled exactly where
            # they leave the convolution chain (weight / bias gradients, input-frame gradient)
            from .fastdvd import alloc_denblock_buffers_split
            self.gscale = float(2.0 ** round(np.log2(H * W / 2.0)))
            self.stash = {p: alloc_denblock_buffers_split(B, H, W, dev, alias=False) for p in ('temp1', 'temp2')}
            f = lambda c, h, w: torch.empty(B, c // 8, 2, h, w, 8, dtype=torch.float16, device=dev)  # noqa: E731
        else:
            self.stash = {p: alloc_denblock_buffers(B, H, W, dev, alias=False) for p in ('temp1', 'temp2')}
            f = lambda c, h, w: torch.empty(B, c // 8, h, w, 8, dtype=F32, device=dev)  # noqa: E731
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        # gradient scratch (reused by both stages)
        self.g = dict(x8=f(8, H, W), f32a=f(32, H, W), f32b=f(32, H, W), f32c=f(32, H, W), f96=f(96, H, W), f16=f(16, H, W),
                      up64=f(64, H, W), h64a=f(64, H2, W2), h64b=f(64, H2, W2), h64c=f(64, H2, W2), h128=f(128, H2, W2),
                      q128a=f(128, H4, W4), q128b=f(128, H4, W4), q256=f(256, H4, W4))
        if self.split:
            self.g['x8_32'] = torch.empty(B, 1, H, W, 8, dtype=F32, device=dev)
            self.g['f16_32'] = torch.empty(B, 2, H, W, 8, dtype=F32, device=dev)
        self.s1 = torch.empty(B, 3, H, W, dtype=F32, device=dev)
        self.out = torch.empty_like(self.s1)
        self.dout = torch.empty_like(self.s1)
        self.ds1 = torch.empty_like(self.s1)
        ws = max(self.lib.scipnp_conv3x3_wgrad_workspace_floats(ci, co, _split_slabs(ci) if self.split else self.NSLAB)
                 for _, _, ci, co, *_ in self.blocks['temp1'].spec)
        if wino:
            ws = max([ws] + [self.lib.scipnp_conv3x3_wgrad_wino_workspace_floats(ci, co, _wino_slabs(ci))
                             for _, _, ci, co, *_ in self.blocks['temp1'].spec])
        self.ws = torch.empty(ws, dtype=F32, device=dev)
        self.bws = torch.empty(32 * 64 * 8, dtype=F32, device=dev)
        nb_ = C.c_int(0)
        _lib.check(self.lib.scipnp_fastdvd_loss_grad(None, None, None, None, None, H // 2, W // 2, B, C.byref(nb_), None), 'size')
        self.loss_part = torch.empty(nb_.value, dtype=torch.float64, device=dev)
        self.step = 0

    def pack(self):
        for b in self.blocks.values():
            b.pack()

    def forward(self, frames, sigma):
        from .fastdvd import denblock_forward, denblock_forward_split
        if self.split:
            denblock_forward_split(self.blocks['temp1'].fwd_s, frames, sigma, self.s1, self.stash['temp1'])
            denblock_forward_split(self.blocks['temp2'].fwd_s, self.s1, sigma, self.out, self.stash['temp2'])
            return
        denblock_forward(self.blocks['temp1'].fwd, frames, sigma, self.s1, self.stash['temp1'], self.blocks['temp1'].fwd_w)
        denblock_forward(self.blocks['temp2'].fwd, self.s1, sigma, self.out, self.stash['temp2'], self.blocks['temp2'].fwd_w)

    def loss_and_grad(self, y_pm, Phi_pm):
        eng = self.eng
        nb_ = C.c_int(0)
        _lib.check(self.lib.scipnp_fastdvd_loss_grad(_ptr(self.out), _ptr(Phi_pm), _ptr(y_pm), _ptr(self.dout),
                                                     _ptr(self.loss_part), eng.H // 2, eng.W // 2, eng.B, C.byref(nb_), _s()),
                   'scipnp_fastdvd_loss_grad')
        return _DeviceMean(ops.sum_rows_f64(self.loss_part), float(eng.H * eng.W))

    def _bwd(self, blk, i, dz, out, n, h, w, residual=None, mask=None):
        """backward-data of layer i: out = [mask]( conv(dz; W_i^T flipped, BN scale folded) [+ residual] );
        (h, w) = size of dz (already zero-upsampled for stride-2 layers)."""
        _k, _bn, cin, cout, *_r = blk.spec[i]
        if self.split:
            return ops.conv3x3_c8s(dz, blk.bwd_s[i], cin, out=out, mask=mask, residual=residual)
        if blk.bwd_w is not None:
            return ops.conv3x3_c8w(dz, blk.bwd_w[i], cin, residual=residual, mask_src=mask, out=out)
        flags = (2 if residual is not None else 0) | (16 if mask is not None else 0)
        _lib.check(self.lib.scipnp_conv3x3_c8_ex(_ptr(dz), _ptr(blk.bwd[i]), _ptr(out), _ptr(residual), _ptr(mask), n, cout,
                                                 cin, h, w, flags, _s()), 'backward-data conv')
        return out

    def backward_block(self, name, dout, dframes=None):
        """dout: planar gradient at the DenBlock output; fills the block's parameter gradients and, if `dframes`
        is given, the gradient w.r.t. the planar input frames (incl. the `center -` path)."""
        lib, eng, g = self.lib, self.eng, self.g
        blk, a = self.blocks[name], self.stash[name]
        B, H, W = eng.B, eng.H, eng.W
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        sp = self.split
        inv = 1.0 / self.gscale if sp else 1.0
        def gl(i, x_in, dy, h, w):
            if TAP is not None:
                TAP(name, i, dy)
            blk.grads_of_layer(i, x_in, dy, B, h, w, self.ws, self.bws, _split_slabs(blk.spec[i][2]) if sp else self.NSLAB, inv)

        def unshuffle(src, dst, cs, h, w):
            fn = lib.scipnp_pixel_shuffle_bwd_c8s if sp else lib.scipnp_pixel_shuffle_bwd_c8
            _lib.check(fn(_ptr(src), _ptr(dst), B, cs, h, w, _s()), 'unshuffle')

        def upzero(src, dst, c, h, w, hh, ww):
            fn = lib.scipnp_upsample_zero_c8s if sp else lib.scipnp_upsample_zero_c8
            _lib.check(fn(_ptr(src), _ptr(dst), B, c, h, w, hh, ww, _s()), 'upsample')

        if sp:
            _lib.check(lib.scipnp_fastdvd_finish_bwd(_ptr(dout), _ptr(g['x8_32']), B, H, W, _s()), 'finish_bwd')
            ops.c8_scale_to_c8s(g['x8_32'], g['x8'], self.gscale)
        else:
            _lib.check(lib.scipnp_fastdvd_finish_bwd(_ptr(dout), _ptr(g['x8']), B, H, W, _s()), 'finish_bwd')
        gl(15, a['o32'], g['x8'], H, W)
        dy14 = self._bwd(blk, 15, g['x8'], g['f32a'], B, H, W, mask=a['o32'])
        gl(14, a['s32'], dy14, H, W)
        d_s32 = self._bwd(blk, 14, dy14, g['f32b'], B, H, W)                          # skip: also the gradient of x0
        unshuffle(d_s32, g['h128'], 32, H2, W2)
        gl(13, a['c1'], g['h128'], H2, W2)
        dy12 = self._bwd(blk, 13, g['h128'], g['h64a'], B, H2, W2, mask=a['c1'])
        gl(12, a['c0'], dy12, H2, W2)
        dy11 = self._bwd(blk, 12, dy12, g['h64b'], B, H2, W2, mask=a['c0'])
        gl(11, a['s64'], dy11, H2, W2)
        d_s64 = self._bwd(blk, 11, dy11, g['h64c'], B, H2, W2)                        # skip: also the gradient of x1
        unshuffle(d_s64, g['q256'], 64, H4, W4)
        gl(10, a['u1'], g['q256'], H4, W4)
        dy9 = self._bwd(blk, 10, g['q256'], g['q128a'], B, H4, W4, mask=a['u1'])
        gl(9, a['u0'], dy9, H4, W4)
        dy8 = self._bwd(blk, 9, dy9, g['q128b'], B, H4, W4, mask=a['u0'])
        gl(8, a['x2'], dy8, H4, W4)
        dy7 = self._bwd(blk, 8, dy8, g['q128a'], B, H4, W4, mask=a['x2'])
        gl(7, a['d1'], dy7, H4, W4)
        dy6 = self._bwd(blk, 7, dy7, g['q128b'], B, H4, W4, mask=a['d1'])
        gl(6, a['d0'], dy6, H4, W4)
        dy5 = self._bwd(blk, 6, dy6, g['q128a'], B, H4, W4, mask=a['d0'])
        # stride-2 layer 5 (x1 @H/2 -> d0 @H/4): gradient = stride-1 backward of the zero-upsampled dy5
        upzero(dy5, g['h128'], 128, H4, W4, H2, W2)
        gl(5, a['x1'], g['h128'], H2, W2)
        dy4 = self._bwd(blk, 5, g['h128'], g['h64a'], B, H2, W2, residual=d_s64, mask=a['x1'])
        gl(4, a['a1'], dy4, H2, W2)
        dy3 = self._bwd(blk, 4, dy4, g['h64b'], B, H2, W2, mask=a['a1'])
        gl(3, a['a0'], dy3, H2, W2)
        dy2 = self._bwd(blk, 3, dy3, g['h64a'], B, H2, W2, mask=a['a0'])
        upzero(dy2, g['up64'], 64, H2, W2, H, W)
        gl(2, a['x0'], g['up64'], H, W)
        dy1 = self._bwd(blk, 2, g['up64'], g['f32c'], B, H, W, residual=d_s32, mask=a['x0'])
        gl(1, a['t96'], dy1, H, W)
        dy0 = self._bwd(blk, 1, dy1, g['f96'], B, H, W, mask=a['t96'])
        gl(0, a['t_in'], dy0, H, W)
        if dframes is not None:
            d_tin = self._bwd(blk, 0, dy0, g['f16'], B, H, W)
            if sp:
                d_tin = ops.c8s_to_c8(d_tin, g['f16_32'], inv)
            _lib.check(lib.scipnp_fastdvd_unpack_bwd(_ptr(d_tin), _ptr(dout), _ptr(dframes), B, H, W, _s()), 'unpack_bwd')

    def adam(self, lr):
        self.step += 1
        for blk in self.blocks.values():
            _lib.check(self.lib.scipnp_adam_step(_ptr(blk.flat_p), _ptr(blk.flat_g), _ptr(blk.flat_m), _ptr(blk.flat_v),
                                                 blk.flat_p.numel(), float(lr), 0.9, 0.999, 1e-8, self.step, _s()),
                       'scipnp_adam_step')

    def write_back(self):
        for blk in self.blocks.values():
            _download_flat(blk.flat_p, [self.model_sd[('module.' + key) if self.prefixed else key] for key, _p in blk.params])


def legacy_normal(loc, scale, shape):
    """np.random.normal(loc, scale, shape) on the GLOBAL legacy NumPy generator -- same values, same final generator
    state -- computed by libscipnp's host function (scipnp_host_legacy_normal) with the GIL released, so that the
    launching thread keeps feeding the GPU while 6.3 M deviates are drawn.  Falls back to NumPy itself when the global
    generator is not the default MT19937."""
    st = np.random.get_state()
    if st[0] != 'MT19937':
        return np.random.normal(loc, scale, shape)
    lib = _lib.load()
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos, has_gauss, cached = C.c_int(int(st[2])), C.c_int(int(st[3])), C.c_double(float(st[4]))
    out = np.empty(shape, dtype=np.float64)
    _lib.check(lib.scipnp_host_legacy_normal(C.c_void_p(key.ctypes.data), C.byref(pos), C.byref(has_gauss), C.byref(cached),
                                             float(loc), float(scale), C.c_void_p(out.ctypes.data), out.size),
               'scipnp_host_legacy_normal')
    np.random.set_state(('MT19937', key, pos.value, has_gauss.value, cached.value))
    return out


class NoisePrefetch:
    """The reference draws the finetune noise with np.random.normal(0, 5/255, (B,3,H,W)) from the GLOBAL NumPy RNG, one
    draw per finetune event (utils/utils_image.py:183-192) -- 6.3 M float64 normals, ~65 ms of host time at 512x512x8.
    When the solver knows from its schedule that `count` events WILL fire, this worker thread makes exactly those
    draws, in order, while the GPU is busy with the preceding iterations (NumPy releases the GIL while it fills the
    array); the stream of values is the one a synchronous caller would have obtained."""

    def __init__(self, shape, count):
        import queue
        import threading
        self.q = queue.Queue(maxsize=2)
        self._stop = threading.Event()
        self._states = []          # global RNG state in front of draw i
        self._taken = 0
        self.t = threading.Thread(target=self._run, args=(tuple(shape), int(count)), daemon=True)
        self.t.start()

    def _run(self, shape, count):
        import queue
        for _ in range(count):
            if self._stop.is_set():
                return
            self._states.append(np.random.get_state())
            item = legacy_normal(0, 5 / 255, shape)
            while not self._stop.is_set():
                try:
                    self.q.put(item, timeout=0.05)
                    break
                except queue.Full:
                    pass

    def get(self):
        """next draw, or None once every planned draw has been handed out (the caller then draws synchronously)"""
        import queue
        while True:
            try:
                item = self.q.get(timeout=0.05)
                self._taken += 1
                return item
            except queue.Empty:
                if not self.t.is_alive() and self.q.empty():
                    return None

    def close(self):
        """stop the worker and hand unconsumed draws back: the global NumPy RNG is left in the state a caller who drew
        synchronously (only the draws actually used) would have left it in.  While a prefetch is alive nothing else in
        the process may use np.random -- legacy_normal's get_state / draw / set_state is not atomic."""
        self._stop.set()
        self.t.join()
        if self._taken < len(self._states):
            np.random.set_state(self._states[self._taken])


def fastdvdnet_online_finetune(model, eng, frames, y_pm, Phi_pm, sigma, lr_, update_per_iter, logf=None, trace=None,
                               noise=None):
    """reference packages/fastdvdnet/test_fastdvdnet.py:343-451.  frames: planar (B,3,H,W) net input; the net is
    trained on  frames + float32(float64(frames) + N(0,(5/255)^2))  -- i.e. 2*frames + noise, the reference's
    helper already returns input+noise (utils/utils_image.py:183-192) -- with the noise drawn from the GLOBAL
    NumPy RNG like the reference (pass `noise` (B,3,H,W) float64 to override).  All conv weights and BatchNorm
    affine parameters are updated (BatchNorm statistics frozen: every BN is in eval(), :376-379)."""
    _lib.require_gpu()
    steps = [update_per_iter] if isinstance(update_per_iter, int) else list(update_per_iter)
    lrs = [lr_] if isinstance(update_per_iter, int) else list(lr_)
    if noise is None:
        noise = legacy_normal(0, 5 / 255, tuple(frames.shape))
    # frames + float32(float64(frames) + noise): the float64 sum and its rounding are done on the device (same IEEE result)
    noise_d = torch.from_numpy(np.ascontiguousarray(noise, dtype=np.float64)).to(frames.device)
    v_plus = ops.fastdvd_noisy_input(frames.contiguous(), noise_d)
    del noise_d
    tr = _FastDVDTrainer(model, eng)
    tr.pack()
    first = True
    for n_steps, lr_i in zip(steps, lrs):
        tr.step = 0
        for blk in tr.blocks.values():                        # a new Adam per lr group (:385)
            blk.flat_m.zero_(); blk.flat_v.zero_()
        for _ in range(n_steps):
            tr.forward(v_plus, sigma)
            loss = tr.loss_and_grad(y_pm, Phi_pm)
            tr.backward_block('temp2', tr.dout, tr.ds1)
            tr.backward_block('temp1', tr.ds1, None)
            if first and GRAD_HOOK is not None:
                GRAD_HOOK({k: g.clone() for blk in tr.blocks.values() for k, g in blk.grads})
            first = False
            tr.adam(lr_i)
            tr.pack()
            val = float(loss.item())
            print('loss:', val)
            if trace is not None:
                trace.append(val)
    tr.write_back()
    # the engine continues on the device-packed updated weights (no host repack)
    eng.set_packed({p: (b.fwd_s if tr.split else b.fwd) for p, b in tr.blocks.items()})
    return model

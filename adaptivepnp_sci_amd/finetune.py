"""Online finetune of the plug-in denoiser on the measurement loss, on the GPU.

ffdnet_online_finetune     <- packages/ffdnet/test_ffdnet_ipol.py:248-300 (branch `updata_=True`)
fastdvdnet_online_finetune <- packages/fastdvdnet/test_fastdvdnet.py:343-451

FFDNet: fully hand-written -- forward convs keep their activations, the loss gradient, the backward-data
convolutions (conv3x3_c8 with transposed/flipped weights + ReLU mask), the weight/bias gradients (MFMA
GEMM over pixels) and torch.optim.Adam's update are HIP kernels (csrc/finetune.hip, csrc/conv.hip).
A fresh Adam state is created per call, like the reference (`torch.optim.Adam(model.parameters(), lr=lr_)`
at :251); the module's parameters are updated in place.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops
from .nets import ffdnet_layers

F32 = torch.float32


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class _FFDNetTrainer:
    """Device master copies of the parameters, packed forward/backward weights, Adam state and the
    activation stash of one FFDNetEngine geometry."""

    NSLAB = 256

    def __init__(self, model, eng):
        self.eng = eng
        dev = eng.device
        self.layers = ffdnet_layers(model)                # [(weight, bias)] tensors of the module (any device)
        self.w = [w.detach().to(dev, F32).contiguous().clone() for w, _ in self.layers]
        self.b = [b.detach().to(dev, F32).contiguous().clone() for _, b in self.layers]
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
        self.acts = [torch.empty(B, nc // 8, M, N, 8, dtype=F32, device=dev) for _ in range(self.nb - 1)]
        self.dz = [torch.empty(B, nc // 8, M, N, 8, dtype=F32, device=dev) for _ in range(2)]
        self.gout = torch.empty(B, 2, M, N, 8, dtype=F32, device=dev)
        self.dw = [torch.empty_like(w) for w in self.w]
        self.db = [torch.empty_like(b) for b in self.b]
        self.m = [torch.zeros_like(t) for t in self.w + self.b]
        self.v = [torch.zeros_like(t) for t in self.w + self.b]
        ws = max(lib.scipnp_conv3x3_wgrad_workspace_floats(ci, co, self.NSLAB) for ci, co in zip(self.cin, self.cout))
        self.ws = torch.empty(ws, dtype=F32, device=dev)
        self.bws = torch.empty((nc // 8) * 64 * 8, dtype=F32, device=dev)
        nb_ = C.c_int(0)
        _lib.check(lib.scipnp_ffdnet_loss_grad(None, None, None, None, None, M, N, B, C.byref(nb_), None), 'loss size')
        self.loss_part = torch.empty(nb_.value, dtype=torch.float64, device=dev)
        self.step = 0

    def _real(self, l):
        w = self.w[l]
        return w.shape[1], w.shape[0]      # Cin_real, Cout_real

    def pack(self):
        for l in range(self.nb):
            ci_r, co_r = self._real(l)
            _lib.check(self.lib.scipnp_pack_conv3x3_device(_ptr(self.w[l]), _ptr(self.b[l]), _ptr(self.fwd[l]), ci_r, co_r,
                                                           self.cin[l], self.cout[l], 0, _s()), 'pack fwd')
            if l > 0:
                _lib.check(self.lib.scipnp_pack_conv3x3_device(_ptr(self.w[l]), None, _ptr(self.bwd[l]), ci_r, co_r,
                                                               self.cin[l], self.cout[l], 1, _s()), 'pack bwd')

    def forward_keep(self):
        eng = self.eng
        x = eng.in_c8
        for l in range(self.nb - 1):
            ops.conv3x3_c8(x, self.fwd[l], self.nc, relu=True, out=self.acts[l], head=(l == 0))
            x = self.acts[l]
        ops.conv3x3_c8(x, self.fwd[-1], 16, relu=False, out=eng.out_c8)

    def loss_and_grad(self, y_pm, Phi_pm):
        eng = self.eng
        nb_ = C.c_int(0)
        _lib.check(self.lib.scipnp_ffdnet_loss_grad(_ptr(eng.out_c8), _ptr(Phi_pm), _ptr(y_pm), _ptr(self.gout),
                                                    _ptr(self.loss_part), eng.M, eng.N, eng.B, C.byref(nb_), _s()),
                   'scipnp_ffdnet_loss_grad')
        return self.loss_part.sum() / (4.0 * eng.M * eng.N)        # device scalar (float64)

    def backward(self):
        eng = self.eng
        B, M, N = eng.B, eng.M, eng.N
        dz = self.gout
        for l in range(self.nb - 1, -1, -1):
            a_in = eng.in_c8 if l == 0 else self.acts[l - 1]
            ci_r, co_r = self._real(l)
            _lib.check(self.lib.scipnp_conv3x3_wgrad(_ptr(a_in), _ptr(dz), _ptr(self.dw[l]), _ptr(self.ws), self.NSLAB, B,
                                                     ci_r, co_r, self.cin[l], self.cout[l], M, N, _s()), 'wgrad')
            _lib.check(self.lib.scipnp_conv_bias_grad(_ptr(dz), _ptr(self.db[l]), _ptr(self.bws), B, co_r, self.cout[l],
                                                      M, N, _s()), 'bgrad')
            if l > 0:
                nxt = self.dz[l & 1]
                # backward-data: conv of dZ_l with the transposed/flipped weights, masked by ReLU'(A_{l-1})
                _lib.check(self.lib.scipnp_conv3x3_c8(_ptr(dz), _ptr(self.bwd[l]), _ptr(nxt), _ptr(self.acts[l - 1]), B,
                                                      self.cout[l], self.cin[l], M, N, 16, _s()), 'backward-data conv')
                dz = nxt

    def adam(self, lr):
        self.step += 1
        params = self.w + self.b
        grads = self.dw + self.db
        for p, g, m, v in zip(params, grads, self.m, self.v):
            _lib.check(self.lib.scipnp_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), 0.9, 0.999,
                                                 1e-8, self.step, _s()), 'scipnp_adam_step')

    def write_back(self):
        """the reference mutates `model` in place and returns it (test_ffdnet_ipol.py:356-357)"""
        with torch.no_grad():
            for (w, b), wd, bd in zip(self.layers, self.w, self.b):
                w.copy_(wd)
                b.copy_(bd)


def ffdnet_online_finetune(model, eng, y_pm, Phi_pm, sigma, lr_, update_per_iter, logf=None, trace=None):
    """`update_per_iter` Adam steps on the measurement loss for the input currently held in eng.in_c8
    (written by scipnp_pm_pre_denoise); y_pm [4][M][N], Phi_pm [B][4][M][N] plane-major.  Leaves the
    engine's packed weights refreshed and the module's parameters updated; the caller then runs the
    evaluation forward (reference :303-315)."""
    _lib.require_gpu()
    tr = _FFDNetTrainer(model, eng)
    tr.pack()
    for _ in range(update_per_iter):
        tr.forward_keep()
        loss = tr.loss_and_grad(y_pm, Phi_pm)
        tr.backward()
        tr.adam(lr_)
        tr.pack()
        val = float(loss.item())
        print('loss:', val)                                   # the reference prints the loss tensor (:298-299)
        if trace is not None:
            trace.append(val)
    tr.write_back()
    eng.packed = tr.fwd
    eng._ptrs = (C.c_void_p * eng.nb)(*[p.data_ptr() for p in eng.packed])
    return model


def fastdvdnet_online_finetune(model, eng, frames, y_pm, Phi_pm, sigma, lr_, update_per_iter, logf=None, trace=None):
    raise NotImplementedError('FastDVDnet online finetune (reference test_fastdvdnet.py:343-451) is not built yet: '
                              'run with update_=False for the fastdvd_color denoiser')

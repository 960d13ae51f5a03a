"""Denoiser networks of the plug-and-play prior, as the engine sees them.

`FFDNet` / `FastDVDnet` are parameter containers that are state-dict-key compatible with the
reference's checkpoints (models/network_ffdnet.py:27-69: keys `model.{0,2,..}.{weight,bias}`;
packages/fastdvdnet/models.py:146-253: keys `temp{1,2}.<block>.convblock.*`), so that a user can
`load_state_dict` the reference's .pth files, or hand the reference's own nn.Module to the solver.
Their `forward` runs on the HIP kernels only (no PyTorch convolution, no CPU path).

`FFDNetEngine` packs a module's weights for scipnp_conv3x3_c8 (c8 layout, channel groups of 8) and
owns the scratch activations of one forward pass over B frames.
"""
import ctypes as C
import re

import torch
import torch.nn as nn

from . import _lib, ops


class FFDNet(nn.Module):
    """KAIR FFDNet container (in_nc*4+1 -> nc -> ... -> out_nc*4, nb conv3x3 layers)."""

    def __init__(self, in_nc=3, out_nc=3, nc=96, nb=12, act_mode='R'):
        super().__init__()
        if act_mode != 'R':
            raise ValueError("only act_mode='R' (conv+ReLU, no BatchNorm) is supported, as the reference uses")
        self.in_nc, self.out_nc, self.nc, self.nb = in_nc, out_nc, nc, nb
        layers = [nn.Conv2d(in_nc * 4 + 1, nc, 3, 1, 1, bias=True), nn.ReLU(inplace=True)]
        for _ in range(nb - 2):
            layers += [nn.Conv2d(nc, nc, 3, 1, 1, bias=True), nn.ReLU(inplace=True)]
        layers += [nn.Conv2d(nc, out_nc * 4, 3, 1, 1, bias=True)]
        self.model = nn.Sequential(*layers)
        self._engine = None

    def forward(self, x, sigma):
        """x (n,in_nc,H,W) CUDA tensor, sigma (1,1,1,1) or float -> (n,out_nc,H,W); HIP kernels only."""
        from .denoisers import ffdnet_forward_nchw
        return ffdnet_forward_nchw(self, x, float(sigma.reshape(-1)[0]) if torch.is_tensor(sigma) else float(sigma))


def ffdnet_layers(model):
    """[(weight, bias)] of the conv stack of a (reference- or scipnp-) FFDNet module, in order."""
    sd = model.state_dict()
    keys = sorted((int(m.group(1)), k) for k in sd for m in [re.fullmatch(r'(?:module\.)?model\.(\d+)\.weight', k)] if m)
    if not keys:
        raise ValueError('model does not look like an FFDNet (no `model.<i>.weight` keys)')
    out = []
    for _, k in keys:
        out.append((sd[k], sd[k.replace('weight', 'bias')]))
    return out


def default_precision():
    """'f32' (the default since round 4: every product an exact fp32 product on the fp32 MFMA, the reference's arithmetic --
    Winograd F(4x4,3x3) / F(2x2,3x3) / direct kernels, csrc/conv_wino4.hip, conv_wino.hip, conv.hip) or the opt-in 'f16x3'
    (error-compensated split-fp16 operands on the fp16 MFMA, csrc/conv_split.hip: 22 significant bits per operand, fp32
    accumulation, ~1.2-1.3x faster, inside the 1e-5 / 1e-4 dB gates but NARROWER than the reference's fp32).
    Set with config.Config(precision=...) (config.use / AdmmRun(config=...)), the engines' `precision=` / the solvers'
    `conv_precision=` argument, or SCIPNP_CONV_PRECISION (alias: SCIPNP_FFDNET_PRECISION) as the process default.  Applies to the FFDNet / FastDVDnet / DDnet passes and their online finetune."""
    from . import config
    return config.current().precision


WINO_MAX_PIXELS = 1 << 25          # csrc/conv_wino.hip addresses a plane through 32-bit buffer offsets: h * w < 2^25


def f32_conv_form(h=None, w=None):
    """'winograd' (default: F(4x4,3x3) on the fp32 MFMA for layers of >= 16 input and >= 32 output channels,
    csrc/conv_wino4.hip -- 4x fewer products than the direct form, every one an exact fp32 product; F(2x2,3x3),
    csrc/conv_wino.hip, for the narrower layers or with SCIPNP_WINO_F4=0) or 'direct' (csrc/conv.hip) for the stride-1
    layers of the fp32 passes; SCIPNP_F32_CONV.
    Planes of h x w >= 2^25 pixels (FFDNet on frames beyond 11585 x 11585) take the direct form, which has no such bound."""
    from . import config
    f = config.current().f32_form
    if f == 'winograd' and h is not None and h * w >= WINO_MAX_PIXELS:
        return 'direct'
    return f


class FFDNetEngine:
    """Packed weights + scratch for B frames of M x N (half-resolution) activations."""

    def __init__(self, model, B, M, N, device, precision=None):
        self.device = device
        self.B, self.M, self.N = B, M, N
        self.precision = precision or default_precision()
        self.f32_form = f32_conv_form(M, N) if self.precision == 'f32' else None
        self.refresh(model)
        nc = self.nc
        self.scratch = [torch.empty(B * nc * M * N, dtype=torch.float32, device=device) for _ in range(2)]
        self.in_c8 = torch.empty(B, self.cin0 // 8, M, N, 8, dtype=torch.float32, device=device)
        self.in_c8s = (torch.empty(B, self.cin0 // 8, 2, M, N, 8, dtype=torch.float16, device=device)
                       if self.precision == 'f16x3' else None)
        self.out_c8 = torch.empty(B, self.cout_last // 8, M, N, 8, dtype=torch.float32, device=device)

    def refresh(self, model):
        """(Re)pack the weights; call after every optimizer step of the online finetune."""
        layers = ffdnet_layers(model)
        self.nb = len(layers)
        self.nc = layers[0][0].shape[0]
        self.in_ch, self.out_ch = layers[0][0].shape[1], layers[-1][0].shape[0]
        if (self.in_ch, self.out_ch) not in ((13, 12), (5, 4)) or self.nc % 8:
            raise ValueError('FFDNetEngine supports the colour (13 -> nc -> 12) and grayscale (5 -> nc -> 4) networks, nc % 8 == 0')
        self.cin0, self.cout_last = (self.in_ch + 7) // 8 * 8, (self.out_ch + 7) // 8 * 8
        # weights travel to the device in one upload and are packed there (scipnp_pack_conv3x3_device /
        # _split_device): both layouts are kept, the fp32 one serves the C entry scipnp_ffdnet_forward
        dev_t = ops.device_params([w for w, _ in layers] + [b for _, b in layers], torch.device(self.device))
        ws, bs = dev_t[:self.nb], dev_t[self.nb:]
        split = self.precision == 'f16x3'
        self.packed, self.packed_split = [], ([] if split else None)
        for i in range(self.nb):
            cin = self.cin0 if i == 0 else self.nc
            cout = self.cout_last if i == self.nb - 1 else self.nc
            self.packed.append(ops.pack_conv3x3_device(ws[i], bs[i], ops.packed_buffer(cin, cout, self.device, False), cin, cout))
            if split:
                self.packed_split.append(ops.pack_conv3x3_split_device(ws[i], bs[i], ops.packed_buffer(cin, cout, self.device, True),
                                                                       cin, cout))
        self._ptrs = (C.c_void_p * self.nb)(*[p.data_ptr() for p in self.packed])
        self._pack_wino()

    def _pack_wino(self):
        """Winograd-domain weights U = G g G^T of every layer, derived on the device from the fp32 packing."""
        self.packed_wino = None
        if self.f32_form == 'winograd':
            self.packed_wino = []
            for i in range(self.nb):
                cin = self.cin0 if i == 0 else self.nc
                cout = self.cout_last if i == self.nb - 1 else self.nc
                self.packed_wino.append(ops.pack_conv3x3_wino_both(self.packed[i], cin, cout))

    def adopt(self, packed_f32, packed_split=None, packed_wino=None):
        """Take over device-packed weights (the online finetune packs its updated master weights on the GPU,
        scipnp_pack_conv3x3_device / _split_device) instead of re-packing on the host.  packed_wino: the trainer's Winograd-domain
        packings of the same weights (ops.WinoPacked per layer; a layer latched onto the F(4x4) kernel carries only that form) --
        taken over as they are instead of 23 more pack launches per event."""
        self.packed = list(packed_f32)
        self._ptrs = (C.c_void_p * self.nb)(*[p.data_ptr() for p in self.packed])
        if self.precision == 'f16x3':
            self.packed_split = list(packed_split)
        if packed_wino is not None and self.f32_form == 'winograd':
            self.packed_wino = list(packed_wino)
        else:
            self._pack_wino()

    def forward(self, in_c8=None, out_c8=None, events=None):
        """12 conv launches on the current stream (same sequence as the C entry scipnp_ffdnet_forward).
        `events`, if a list, receives (start, end) torch.cuda.Event pairs bracketing the nb-2 body layers
        (bench.py's live roofline measurement)."""
        out_c8 = self.out_c8 if out_c8 is None else out_c8
        B, M, N, nc = self.B, self.M, self.N, self.nc
        if self.precision == 'f16x3':
            return self._forward_split(in_c8, out_c8, events)
        in_c8 = self.in_c8 if in_c8 is None else in_c8
        buf = [s.view(B, nc // 8, M, N, 8) for s in self.scratch]
        if self.packed_wino is not None:
            conv, pk = ops.conv3x3_c8w, self.packed_wino
        else:
            conv, pk = ops.conv3x3_c8, self.packed
        conv(in_c8, pk[0], nc, relu=True, out=buf[0], head=True)
        if events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        cur = 0
        for l in range(1, self.nb - 1):
            conv(buf[cur], pk[l], nc, relu=True, out=buf[cur ^ 1])
            cur ^= 1
        if events is not None:
            e1.record()
            events.append((e0, e1))
        # (head=True only selects the kernel symbol of the first / last layer, so that profiler statistics of the body
        # layers' symbol hold body launches only)
        conv(buf[cur], pk[self.nb - 1], self.cout_last, relu=False, out=out_c8, head=True)
        return out_c8

    def _forward_split(self, in_c8, out_c8, events):
        """same 12 layers on the split-fp16 kernels; the input is self.in_c8s (written by scipnp_pm_pre_denoise_ex)
        unless an fp32 c8 tensor is passed, the output is fp32 c8."""
        B, M, N, nc = self.B, self.M, self.N, self.nc
        x = self.in_c8s if in_c8 is None else ops.c8_to_c8s(in_c8)
        buf = [s.view(torch.float16).view(B, nc // 8, 2, M, N, 8) for s in self.scratch]   # same bytes as fp32 c8
        pk = self.packed_split

        def layers(sl):
            ops.conv3x3_c8s(x[sl], pk[0], nc, relu=True, out=buf[0][sl], head=True)
            cur = 0
            for l in range(1, self.nb - 1):
                ops.conv3x3_c8s(buf[cur][sl], pk[l], nc, relu=True, out=buf[cur ^ 1][sl])
                cur ^= 1
            ops.conv3x3_c8s(buf[cur][sl], pk[self.nb - 1], self.cout_last, relu=False, out=out_c8[sl], f32_out=True)

        if events is not None:
            # bench.py's live roofline measurement: one stream, an event pair around the body layers
            ops.conv3x3_c8s(x, pk[0], nc, relu=True, out=buf[0], head=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            cur = 0
            for l in range(1, self.nb - 1):
                ops.conv3x3_c8s(buf[cur], pk[l], nc, relu=True, out=buf[cur ^ 1])
                cur ^= 1
            e1.record()
            events.append((e0, e1))
            ops.conv3x3_c8s(buf[cur], pk[self.nb - 1], self.cout_last, relu=False, out=out_c8, f32_out=True)
            return out_c8
        # the frames are independent: two half-batches on two HIP streams (SCIPNP_STREAMS) -- the body layer's grid is 2.67
        # generations of workgroups, the other stream's launches fill the CUs its last generation leaves idle
        ops.on_side_streams(B, layers)
        return out_c8

    def forward_c_entry(self, in_c8=None, out_c8=None):
        """Same pass through the single C entry point scipnp_ffdnet_forward (what a C/C++ host would call)."""
        if self.in_ch != 13:
            raise _lib.ScipnpError('scipnp_ffdnet_forward is the colour network (13 -> nc -> 12)')
        in_c8 = self.in_c8 if in_c8 is None else in_c8
        out_c8 = self.out_c8 if out_c8 is None else out_c8
        lib = _lib.load()
        _lib.require_gpu()
        if self.packed_wino is not None:                  # the engine's own form: Winograd layers, one C call
            ptrs = (C.c_void_p * self.nb)(*[(p.w.data_ptr() if p.w is not None else None) for p in self.packed_wino])   # (None: F(4x4) only)
            p4 = (C.c_void_p * self.nb)(*[(p.f4.data_ptr() if (p.f4 is not None and ops.wino_f4_enabled()) else None)
                                         for p in self.packed_wino])
            rc = lib.scipnp_ffdnet_forward_c8w4(C.c_void_p(in_c8.data_ptr()), C.c_void_p(out_c8.data_ptr()), ptrs, p4, self.nb,
                                                self.nc, C.c_void_p(self.scratch[0].data_ptr()),
                                                C.c_void_p(self.scratch[1].data_ptr()), self.B, self.M, self.N,
                                                _lib.stream_ptr())
            _lib.check(rc, 'scipnp_ffdnet_forward_c8w4')
            return out_c8
        rc = lib.scipnp_ffdnet_forward(C.c_void_p(in_c8.data_ptr()), C.c_void_p(out_c8.data_ptr()), self._ptrs,
                                       self.nb, self.nc, C.c_void_p(self.scratch[0].data_ptr()),
                                       C.c_void_p(self.scratch[1].data_ptr()), self.B, self.M, self.N,
                                       _lib.stream_ptr())
        _lib.check(rc, 'scipnp_ffdnet_forward')
        return out_c8

    def forward_c_entry_split(self, in_c8s=None, out_c8=None, side=None):
        """The split-fp16 pass through its single C entry point scipnp_ffdnet_forward_c8s; side = (torch.cuda.Stream,
        torch.cuda.Event, torch.cuda.Event): the caller's side stream and fork / join events for scipnp_ffdnet_forward_c8s_2s
        (half of the frames on that stream; the library creates none of its own)."""
        if self.precision != 'f16x3' or self.in_ch != 13:
            raise _lib.ScipnpError('scipnp_ffdnet_forward_c8s needs the colour network and precision f16x3')
        in_c8s = self.in_c8s if in_c8s is None else in_c8s
        out_c8 = self.out_c8 if out_c8 is None else out_c8
        lib = _lib.load()
        _lib.require_gpu()
        ptrs = (C.c_void_p * self.nb)(*[p.data_ptr() for p in self.packed_split])
        if side is not None:
            st, fork, join = side
            for ev in (fork, join):                   # torch creates the hipEvent_t lazily, at the first record
                if not ev.cuda_event:
                    ev.record(st)
            rc = lib.scipnp_ffdnet_forward_c8s_2s(C.c_void_p(in_c8s.data_ptr()), C.c_void_p(out_c8.data_ptr()), ptrs, self.nb,
                                                  self.nc, C.c_void_p(self.scratch[0].data_ptr()),
                                                  C.c_void_p(self.scratch[1].data_ptr()), self.B, self.M, self.N,
                                                  _lib.stream_ptr(), C.c_void_p(st.cuda_stream), C.c_void_p(fork.cuda_event),
                                                  C.c_void_p(join.cuda_event))
            _lib.check(rc, 'scipnp_ffdnet_forward_c8s_2s')
            return out_c8
        rc = lib.scipnp_ffdnet_forward_c8s(C.c_void_p(in_c8s.data_ptr()), C.c_void_p(out_c8.data_ptr()), ptrs, self.nb,
                                           self.nc, C.c_void_p(self.scratch[0].data_ptr()),
                                           C.c_void_p(self.scratch[1].data_ptr()), self.B, self.M, self.N,
                                           _lib.stream_ptr())
        _lib.check(rc, 'scipnp_ffdnet_forward_c8s')
        return out_c8

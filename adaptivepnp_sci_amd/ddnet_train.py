"""Online finetune of the deep demosaicker on the GPU -- the `args.dm_update` branch of the reference's `test_ddnet`
(packages/DDnet/DDnet_test.py:248-296):

    for update_per_iter steps:
        out  = DDnet on every frame's 5-frame window                       (ddnet_seqdenoise, :166-206)
        loss = MSE(input CFA-site cube, CFA samples of out)                 (gen_bayer_img :208-216, :273-275)
        a NEW torch.optim.Adam(model.parameters(), lr=dm_lr); zero_grad; backward; step       (:277-280)
    then the pass itself without gradients                                  (:285-288)

The network has no BatchNorm and no dropout (every nn.BatchNorm2d of models/network_demosaicking.py is commented out), so
`model.train()` changes nothing and the forward is the engine's.  Everything runs in fp32 on the library's direct convolution
kernels (csrc/conv.hip forward / backward-data with transposed weights and the ReLU-mask epilogue, csrc/finetune.hip weight
gradients) plus the DDnet glue adjoints of csrc/ddnet.hip; the engine's precision only matters for the final pass.

Backward graph (B frames, E = 3B first-stage evaluations):
    d out -> mix^T            -> d s2 (2B), d weight_tensor_out
    temp2 (2B evaluations)    : `in1 + x` and the U-Net -> its weight gradients, d s1 (6B stage-1 outputs)
    temp1 (E, full res)       : its weight gradients, d weight_tensor_in  (gates: sum g * raw input frame)
    temp11 (E, half res)      : fusion convolutions, bilinear x2 adjoint, `in1 + x`, U-Net -> weight gradients, d weight_tensor_in2
A NEW Adam per step means zero moments and step count 1 every time: the update is lr * g / (|g| + 1e-8) element by element.
No driver and no default of the two solver entry points enables this branch; it is reached through the plug-in
`denoisers.test_ddnet(..., args)` only."""
import ctypes as C

import torch

from . import _lib, ops
from .ddnet import C0, C1, C2, CI, _LAYERS, _p8, _strip, _unet_convs
from .finetune import _DeviceMean, _carve, _download_flat, _ptr, _s, _upload_flat

F32 = torch.float32
NSLAB = 256
GRAD_HOOK = None        # test hook: callable({state-dict key: device gradient}) after the FIRST backward of a call


class _Stash:
    """every activation of one DenBlock pass over E evaluations at (h, w), each in its own buffer (the engine's _Bufs reuses
    dead ones; the backward pass needs them all)"""

    def __init__(self, E, cin, h, w, device):
        def act(c, hh, ww):
            return torch.empty(E, c // 8, hh, ww, 8, dtype=F32, device=device)
        h2, w2, h4, w4 = h // 2, w // 2, h // 4, w // 4
        self.t_in, self.t96 = act(cin, h, w), act(CI, h, w)
        self.x0, self.s0, self.o0 = act(C0, h, w), act(C0, h, w), act(C0, h, w)
        self.a0, self.a1, self.x1, self.s1, self.c0, self.c1 = (act(C1, h2, w2) for _ in range(6))
        self.d0, self.d1, self.x2, self.u0, self.u1 = (act(C2, h4, w4) for _ in range(5))
        self.x8 = act(8, h, w)


class _Scratch:
    """gradient scratch of one DenBlock backward over E evaluations at (h, w)"""

    def __init__(self, E, cin, h, w, device):
        def act(c, hh, ww):
            return torch.empty(E, c // 8, hh, ww, 8, dtype=F32, device=device)
        h2, w2, h4, w4 = h // 2, w // 2, h // 4, w // 4
        self.f8, self.f96 = act(8, h, w), act(CI, h, w)
        self.fin = {8: act(8, h, w), 16: act(16, h, w)}                       # d t_in by the block's padded input width
        self.fa, self.fb, self.fc = (act(C0, h, w) for _ in range(3))
        self.up1 = act(C1, h, w)
        self.ha, self.hb, self.hc = (act(C1, h2, w2) for _ in range(3))
        self.h96, self.up2 = act(4 * C0, h2, w2), act(C2, h2, w2)
        self.qa, self.qb = act(C2, h4, w4), act(C2, h4, w4)
        self.q160 = act(4 * C1, h4, w4)


class _Block:
    """master weights (views of the trainer's flat parameter buffer), packed forward / backward-data weights and the
    gradients of one DDnet DenBlock (no BatchNorm; optional fusion convolutions of temp11)"""

    def __init__(self, prefix, ch_each, weights, grads, device, lib):
        self.prefix, self.ch, self.lib = prefix, ch_each, lib
        self.W, self.dW = weights, grads                     # lists over the 16 (+2) layers, OIHW
        self.cin0 = _p8(3 * ch_each)
        self.spec = [(self.cin0 if cin is None else cin, cout, relu, s2, sh) for _k, cin, cout, relu, s2, sh in _LAYERS]
        if len(weights) > len(_LAYERS):
            self.spec += [(8, 8, True, False, False), (8, 8, False, False, False)]          # fusion: 4 -> 4 (ReLU), 4 -> 3
        per = weights[0].shape[0] // 3
        self.per = per
        self.dense0 = torch.zeros(3 * per, 3 * ch_each, 3, 3, dtype=F32, device=device)     # block-diagonal grouped conv
        self.G0 = torch.empty_like(self.dense0)
        self.fwd = [torch.empty(lib.scipnp_conv3x3_packed_floats(ci, co), dtype=F32, device=device) for ci, co, *_ in self.spec]
        self.bwd = [torch.empty(lib.scipnp_conv3x3_packed_floats(co, ci), dtype=F32, device=device) for ci, co, *_ in self.spec]

    def dense_w(self, i):
        if i != 0:
            return self.W[i]
        per, ch = self.per, self.ch
        for g in range(3):                                   # data movement only: grouped -> block-diagonal
            self.dense0[g * per:(g + 1) * per, g * ch:(g + 1) * ch] = self.W[0][g * per:(g + 1) * per]
        return self.dense0

    def pack(self):
        for i, (cin, cout, *_r) in enumerate(self.spec):
            w = self.dense_w(i)
            co_r, ci_r = w.shape[0], w.shape[1]
            for buf, tr in ((self.fwd[i], 0), (self.bwd[i], 1)):
                _lib.check(self.lib.scipnp_pack_conv3x3_device_scaled(_ptr(w), None, None, _ptr(buf), ci_r, co_r, cin, cout, tr,
                                                                      _s()), 'pack')

    def wgrad(self, i, x_in, dy, n, h, w, ws):
        """dW of layer i from its input activation and the gradient at its (pre-ReLU) output, summed over the n evaluations"""
        cin, cout, *_r = self.spec[i]
        wd = self.dense_w(i) if i == 0 else self.W[i]
        co_r, ci_r = wd.shape[0], wd.shape[1]
        dst = self.G0 if i == 0 else self.dW[i]
        _lib.check(self.lib.scipnp_conv3x3_wgrad(_ptr(x_in), _ptr(dy), _ptr(dst), _ptr(ws), NSLAB, n, ci_r, co_r, cin, cout, h, w,
                                                 _s()), 'scipnp_conv3x3_wgrad')
        if i == 0:
            per, ch = self.per, self.ch
            for g in range(3):
                self.dW[0][g * per:(g + 1) * per] = self.G0[g * per:(g + 1) * per, g * ch:(g + 1) * ch]

    def bwd_data(self, i, dz, out, n, h, w, residual=None, mask=None):
        """out = [mask]( conv(dz; W_i transposed and flipped) [+ residual] ); (h, w) = size of dz"""
        cin, cout, *_r = self.spec[i]
        flags = (2 if residual is not None else 0) | (16 if mask is not None else 0)
        _lib.check(self.lib.scipnp_conv3x3_c8_ex(_ptr(dz), _ptr(self.bwd[i]), _ptr(out), _ptr(residual), _ptr(mask), n, cout, cin,
                                                 h, w, flags, _s()), 'backward-data conv')
        return out


class DDnetTrainer:
    def __init__(self, model, eng):
        self.eng, self.lib = eng, _lib.load()
        dev = self.dev = eng.device
        self.model_sd = model.state_dict()
        self.prefixed = any(k.startswith('module.') for k in self.model_sd)
        sd = _strip(self.model_sd)
        B, H, W = eng.B, eng.H, eng.W
        self.B, self.H, self.W, self.E = B, H, W, 3 * B
        # ---- parameters that receive a gradient, in ONE flat buffer (the unused `inc` blocks keep .grad = None in the reference:
        # Adam skips them)
        names = []
        for pre in ('temp1', 'temp2', 'temp11'):
            names += [f'{pre}.{key}.weight' for key, *_r in _LAYERS]
            if pre == 'temp11':
                names += ['temp11.fusion.convblock.0.weight', 'temp11.fusion.convblock.2.weight']
        names += ['weight_tensor_in', 'weight_tensor_in2', 'weight_tensor_out']
        self.names = names
        srcs = [sd[k].detach() for k in names]
        total = sum(t.numel() for t in srcs)
        self.flat_p, self.flat_g, self.flat_m, self.flat_v = (torch.empty(total, dtype=F32, device=dev) for _ in range(4))
        pv, gv = _carve(self.flat_p, srcs), _carve(self.flat_g, srcs)
        _upload_flat(self.flat_p, srcs)
        self.params, self.grads = dict(zip(names, pv)), dict(zip(names, gv))
        nl = len(_LAYERS)
        self.blocks = {'temp1': _Block('temp1', 1, pv[:nl], gv[:nl], dev, self.lib),
                       'temp2': _Block('temp2', 3, pv[nl:2 * nl], gv[nl:2 * nl], dev, self.lib),
                       'temp11': _Block('temp11', 4, pv[2 * nl:3 * nl + 2], gv[2 * nl:3 * nl + 2], dev, self.lib)}
        self.a, self.a2, self.a3 = pv[-3], pv[-2], pv[-1]
        self.da, self.da2, self.da3 = gv[-3], gv[-2], gv[-1]
        E, h, w = self.E, H // 2, W // 2
        self.st1 = _Stash(E, 8, H, W, dev)
        self.st11 = _Stash(E, 16, h, w, dev)
        self.st2 = _Stash(2 * B, 16, H, W, dev)
        self.gfull = _Scratch(E, 16, H, W, dev)              # temp1 (E) and temp2 (2B evaluations: leading slices)
        self.ghalf = _Scratch(E, 16, h, w, dev)
        f = lambda *shape: torch.empty(*shape, dtype=F32, device=dev)          # noqa: E731
        self.p4, self.fu_in, self.fu_mid = f(E, 4, h, w), f(E, 1, H, W, 8), f(E, 1, H, W, 8)
        self.s1, self.s2, self.out = f(2 * E, 3, H, W), f(2 * B, 3, H, W), f(B, 3, H, W)
        self.dout, self.ds2, self.ds1, self.dp4 = f(B, 3, H, W), f(2 * B, 3, H, W), f(2 * E, 3, H, W), f(E, 4, h, w)
        self.dfu_a, self.dfu_b = f(E, 1, H, W, 8), f(E, 1, H, W, 8)
        self.scale1, self.scale11 = f(E, 3, 1), f(E, 3, 4)
        ws = max(self.lib.scipnp_conv3x3_wgrad_workspace_floats(ci, co, NSLAB) for blk in self.blocks.values() for ci, co, *_ in blk.spec)
        self.ws = f(ws)
        n = C.c_int(0)
        _lib.check(self.lib.scipnp_ddnet_loss_grad(None, None, None, None, H, W, B, C.byref(n), None), 'size')
        self.loss_part = torch.empty(n.value, dtype=torch.float64, device=dev)
        _lib.check(self.lib.scipnp_ddnet_mix_bwd(None, None, None, None, None, B, H, W, C.byref(n), None), 'size')
        self.part_mix = torch.empty(6, n.value, dtype=torch.float64, device=dev)
        _lib.check(self.lib.scipnp_ddnet_gather_bwd(None, None, 0, None, None, None, None, None, E, B, 1, H, W, C.byref(n), None), 'size')
        self.part1 = torch.empty(9, n.value, dtype=torch.float64, device=dev)
        _lib.check(self.lib.scipnp_ddnet_gather_bwd(None, None, 0, None, None, None, None, None, E, B, 4, h, w, C.byref(n), None), 'size')
        self.part11 = torch.empty(36, n.value, dtype=torch.float64, device=dev)

    # ------------------------------------------------------------------ forward with every activation kept
    def pack(self):
        for blk in self.blocks.values():
            blk.pack()
        B = self.B
        # gate scalars per evaluation e = j*B + n and slot i (as DDnetEngine.refresh; device-side views and expands only)
        self.scale1.copy_(self.a.reshape(3, 3, 1)[:, None].expand(3, B, 3, 1).reshape(3 * B, 3, 1))
        self.scale11.copy_(self.a2.reshape(3, 3, 4)[:, None].expand(3, B, 3, 4).reshape(3 * B, 3, 4))

    def forward(self, planes, mosaic):
        eng, E, B, H, W = self.eng, self.E, self.B, self.H, self.W
        h, w = H // 2, W // 2
        b1, b11, b2 = self.blocks['temp1'], self.blocks['temp11'], self.blocks['temp2']
        ops.ddnet_gather(mosaic, eng.idx1, self.scale1, self.st1.t_in, 1, H, W)
        _unet_convs(b1.fwd, self.st1, False)
        ops.ddnet_finish(mosaic, eng.idx1, self.scale1, self.st1.x8, self.s1[:E], 1, 3, H, W)
        ops.ddnet_gather(planes, eng.idx1, self.scale11, self.st11.t_in, 4, h, w)
        _unet_convs(b11.fwd, self.st11, False)
        ops.ddnet_finish(planes, eng.idx1, self.scale11, self.st11.x8, self.p4, 4, 4, h, w)
        ops.bilinear_up2_c8(self.p4, self.fu_in)
        ops.conv3x3_c8(self.fu_in, b11.fwd[16], 8, relu=True, out=self.fu_mid)
        ops.conv3x3_c8(self.fu_mid, b11.fwd[17], 8, out=self.gfull.f8)
        ops.ddnet_finish(None, None, None, self.gfull.f8, self.s1[E:], 3, 3, H, W)
        ops.ddnet_gather(self.s1, eng.idx2, None, self.st2.t_in, 3, H, W)
        _unet_convs(b2.fwd, self.st2, False)
        ops.ddnet_finish(self.s1, eng.idx2, None, self.st2.x8, self.s2, 3, 3, H, W)
        return ops.ddnet_mix(self.s2, self.a3.reshape(2, 3), self.out)

    # ------------------------------------------------------------------ backward
    def loss_and_grad(self, mosaic):
        n = C.c_int(0)
        _lib.check(self.lib.scipnp_ddnet_loss_grad(_ptr(self.out), _ptr(mosaic), _ptr(self.dout), _ptr(self.loss_part), self.H,
                                                   self.W, self.B, C.byref(n), _s()), 'scipnp_ddnet_loss_grad')
        return _DeviceMean(ops.sum_rows_f64(self.loss_part), float(3 * self.B * self.H * self.W))

    def _unet_backward(self, blk, st, g, n, h, w):
        """gradients of a DenBlock's 16 layers from g.f8 = the gradient at its 8-channel tail; returns d t_in (c8)"""
        lib = self.lib
        h2, w2, h4, w4 = h // 2, w // 2, h // 4, w // 4
        sl = lambda t: t[:n]                                  # noqa: E731  (scratch sized for E evaluations, temp2 runs 2B)

        def unshuffle(src, dst, cs, hh, ww):
            _lib.check(lib.scipnp_pixel_shuffle_bwd_c8(_ptr(src), _ptr(dst), n, cs, hh, ww, _s()), 'unshuffle')

        def upzero(src, dst, c, hh, ww, HH, WW):
            _lib.check(lib.scipnp_upsample_zero_c8(_ptr(src), _ptr(dst), n, c, hh, ww, HH, WW, _s()), 'upsample')
        f8, fa, fb, fc, f96, fin, up1 = (sl(t) for t in (g.f8, g.fa, g.fb, g.fc, g.f96, g.fin[blk.cin0], g.up1))
        ha, hb, hc, h96, up2 = (sl(t) for t in (g.ha, g.hb, g.hc, g.h96, g.up2))
        qa, qb, q160 = (sl(t) for t in (g.qa, g.qb, g.q160))
        ws = self.ws
        blk.wgrad(15, st.o0, f8, n, h, w, ws)
        dz14 = blk.bwd_data(15, f8, fa, n, h, w, mask=st.o0)
        blk.wgrad(14, st.s0, dz14, n, h, w, ws)
        d_s0 = blk.bwd_data(14, dz14, fb, n, h, w)                            # skip: also the gradient of x0
        unshuffle(d_s0, h96, C0, h2, w2)
        blk.wgrad(13, st.c1, h96, n, h2, w2, ws)
        dz12 = blk.bwd_data(13, h96, ha, n, h2, w2, mask=st.c1)
        blk.wgrad(12, st.c0, dz12, n, h2, w2, ws)
        dz11 = blk.bwd_data(12, dz12, hb, n, h2, w2, mask=st.c0)
        blk.wgrad(11, st.s1, dz11, n, h2, w2, ws)
        d_s1 = blk.bwd_data(11, dz11, hc, n, h2, w2)                          # skip: also the gradient of x1
        unshuffle(d_s1, q160, C1, h4, w4)
        blk.wgrad(10, st.u1, q160, n, h4, w4, ws)
        dz9 = blk.bwd_data(10, q160, qa, n, h4, w4, mask=st.u1)
        blk.wgrad(9, st.u0, dz9, n, h4, w4, ws)
        dz8 = blk.bwd_data(9, dz9, qb, n, h4, w4, mask=st.u0)
        blk.wgrad(8, st.x2, dz8, n, h4, w4, ws)
        dz7 = blk.bwd_data(8, dz8, qa, n, h4, w4, mask=st.x2)
        blk.wgrad(7, st.d1, dz7, n, h4, w4, ws)
        dz6 = blk.bwd_data(7, dz7, qb, n, h4, w4, mask=st.d1)
        blk.wgrad(6, st.d0, dz6, n, h4, w4, ws)
        dz5 = blk.bwd_data(6, dz6, qa, n, h4, w4, mask=st.d0)
        upzero(dz5, up2, C2, h4, w4, h2, w2)                                  # stride-2 layer 5: zero-upsampled gradient
        blk.wgrad(5, st.x1, up2, n, h2, w2, ws)
        dz4 = blk.bwd_data(5, up2, ha, n, h2, w2, residual=d_s1, mask=st.x1)
        blk.wgrad(4, st.a1, dz4, n, h2, w2, ws)
        dz3 = blk.bwd_data(4, dz4, hb, n, h2, w2, mask=st.a1)
        blk.wgrad(3, st.a0, dz3, n, h2, w2, ws)
        dz2 = blk.bwd_data(3, dz3, ha, n, h2, w2, mask=st.a0)
        upzero(dz2, up1, C1, h2, w2, h, w)                                    # stride-2 layer 2
        blk.wgrad(2, st.x0, up1, n, h, w, ws)
        dz1 = blk.bwd_data(2, up1, fc, n, h, w, residual=d_s0, mask=st.x0)
        blk.wgrad(1, st.t96, dz1, n, h, w, ws)
        dz0 = blk.bwd_data(1, dz1, f96, n, h, w, mask=st.t96)
        blk.wgrad(0, st.t_in, dz0, n, h, w, ws)
        return blk.bwd_data(0, dz0, fin, n, h, w)

    def backward(self, planes, mosaic):
        lib, eng, E, B, H, W = self.lib, self.eng, self.E, self.B, self.H, self.W
        h, w = H // 2, W // 2
        b1, b11, b2 = self.blocks['temp1'], self.blocks['temp11'], self.blocks['temp2']
        n = C.c_int(0)
        # mix
        _lib.check(lib.scipnp_ddnet_mix_bwd(_ptr(self.dout), _ptr(self.s2), _ptr(self.a3), _ptr(self.ds2), _ptr(self.part_mix), B, H, W,
                                            C.byref(n), _s()), 'scipnp_ddnet_mix_bwd')
        self.da3.view(-1).copy_(ops.sum_rows_f64(self.part_mix))
        # second stage: temp2 on both branches (2B evaluations)
        g = self.gfull
        _lib.check(lib.scipnp_ddnet_finish_bwd(_ptr(self.ds2), _ptr(g.f8), 2 * B, 3, H, W, _s()), 'finish_bwd')
        d_tin = self._unet_backward(b2, self.st2, g, 2 * B, H, W)
        _lib.check(lib.scipnp_ddnet_gather_bwd(_ptr(d_tin), _ptr(self.ds2), 3, _ptr(self.s1), _ptr(eng.idx2), None, _ptr(self.ds1), None,
                                               2 * B, 2 * B, 3, H, W, C.byref(n), _s()), 'gather_bwd temp2')
        # first stage, mosaic branch: temp1
        _lib.check(lib.scipnp_ddnet_finish_bwd(_ptr(self.ds1[:E]), _ptr(g.f8), E, 3, H, W, _s()), 'finish_bwd')
        d_tin = self._unet_backward(b1, self.st1, g, E, H, W)
        _lib.check(lib.scipnp_ddnet_gather_bwd(_ptr(d_tin), _ptr(self.ds1[:E]), 3, _ptr(mosaic), _ptr(eng.idx1), None, None,
                                               _ptr(self.part1), E, B, 1, H, W, C.byref(n), _s()), 'gather_bwd temp1')
        self.da.view(-1).copy_(ops.sum_rows_f64(self.part1))
        # first stage, Bayer-plane branch: fusion convolutions, bilinear x2, temp11 at half resolution
        _lib.check(lib.scipnp_ddnet_finish_bwd(_ptr(self.ds1[E:]), _ptr(g.f8), E, 3, H, W, _s()), 'finish_bwd')
        b11.wgrad(17, self.fu_mid, g.f8, E, H, W, self.ws)
        dzf = b11.bwd_data(17, g.f8, self.dfu_a, E, H, W, mask=self.fu_mid)
        b11.wgrad(16, self.fu_in, dzf, E, H, W, self.ws)
        d_fu_in = b11.bwd_data(16, dzf, self.dfu_b, E, H, W)
        _lib.check(lib.scipnp_bilinear_up2_bwd_c8(_ptr(d_fu_in), _ptr(self.dp4), E, h, w, _s()), 'bilinear_bwd')
        gh = self.ghalf
        _lib.check(lib.scipnp_ddnet_finish_bwd(_ptr(self.dp4), _ptr(gh.f8), E, 4, h, w, _s()), 'finish_bwd')
        d_tin = self._unet_backward(b11, self.st11, gh, E, h, w)
        _lib.check(lib.scipnp_ddnet_gather_bwd(_ptr(d_tin), _ptr(self.dp4), 4, _ptr(planes), _ptr(eng.idx1), None, None,
                                               _ptr(self.part11), E, B, 4, h, w, C.byref(n), _s()), 'gather_bwd temp11')
        self.da2.view(-1).copy_(ops.sum_rows_f64(self.part11))

    def adam_fresh(self, lr):
        """torch.optim.Adam created for THIS step (DDnet_test.py:277): zero moments, step count 1"""
        self.flat_m.zero_()
        self.flat_v.zero_()
        _lib.check(self.lib.scipnp_adam_step(_ptr(self.flat_p), _ptr(self.flat_g), _ptr(self.flat_m), _ptr(self.flat_v),
                                             self.flat_p.numel(), float(lr), 0.9, 0.999, 1e-8, 1, _s()), 'scipnp_adam_step')

    def write_back(self):
        _download_flat(self.flat_p, [self.model_sd[('module.' + k) if self.prefixed else k] for k in self.names])


def ddnet_online_finetune(model, eng, planes, mosaic, lr, update_per_iter, logf=None):
    """The `dm_update` steps of test_ddnet on the HIP kernels: `model`'s parameters are updated in place (module tensors and the
    engine's packed weights).  planes (B,4,H/2,W/2), mosaic (B,H,W) as DDnetEngine.forward takes them.  Returns the losses."""
    tr = DDnetTrainer(model, eng)
    losses = []
    for it in range(int(update_per_iter)):
        tr.pack()
        tr.forward(planes, mosaic)
        loss = tr.loss_and_grad(mosaic)
        tr.backward(planes, mosaic)
        if it == 0 and GRAD_HOOK is not None:
            GRAD_HOOK({k: v.clone() for k, v in tr.grads.items()})
        tr.adam_fresh(lr)
        val = loss.item()
        losses.append(val)
        print('ddn loss: {}'.format(val))
        if logf is not None:
            logf.write('ddn loss: {}\n'.format(val))
    tr.write_back()
    eng.refresh(model)
    return losses

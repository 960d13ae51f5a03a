"""DDnet deep demosaicking on the HIP kernels.

`DDnet` is a parameter container with the reference's state-dict keys (models/network_demosaicking.py:381-463:
temp1 / temp2 / temp11 DenBlocks without BatchNorm, base width 20, and the gate scalars weight_tensor_in /
weight_tensor_in2 / weight_tensor_out); wrap it in nn.DataParallel or not -- an optional `module.` prefix is stripped.

`DDnetEngine.forward(planes, mosaic)` = packages/DDnet/DDnet_test.py:166-216 (sliding 5-frame window, circular temporal
indexing) + DDnet.forward for every output frame n of the cube, all frames at once:
  stage 1a  temp1  on the 3B mosaic triplets   (frames n-2+j .. n+j, j = 0..2, gate scalars a[3j..3j+2])   full res
  stage 1b  temp11 on the 3B Bayer-plane triplets (gates a2), `in1 + x`, bilinear x2, fusion block          half res
  stage 2   temp2  on the 2B triples of stage-1 outputs, `in1 + x`
  mix       a3[0]*out_a + a3[1]*out_b
The gate scalars differ between the three windows of a frame, so (unlike FastDVDnet) the 3B stage-1 evaluations are
all distinct.  Channel widths 20 / 40 / 80 / 90 are zero-padded to the 8-channel groups of the c8 layout
(24 / 40 / 80 / 96); padded weights are zero, so padded activations stay zero.
"""
import torch
import torch.nn as nn

from . import ops

BASE = 20


def _cr(cin, cout, stride=1, groups=1):
    return [nn.Conv2d(cin, cout, 3, stride=stride, padding=1, groups=groups, bias=False), nn.ReLU(inplace=True)]


class _Blk(nn.Module):
    def __init__(self, *mods):
        super().__init__()
        self.convblock = nn.Sequential(*mods)


class DDDenBlock(nn.Module):
    def __init__(self, num_input_frames=3, ch_each_frame=3, bayer4=False):
        super().__init__()
        f, c0 = num_input_frames, BASE
        c1, c2 = 2 * c0, 4 * c0
        self.inc = _Blk(*_cr(f * 4, f * 30, groups=f), *_cr(f * 30, c0))          # unused (noise-map input); key parity
        self.inc_1 = _Blk(*_cr(f * ch_each_frame, f * 30, groups=f), *_cr(f * 30, c0))
        self.downc0 = _Blk(*_cr(c0, c1, stride=2), _Blk(*_cr(c1, c1), *_cr(c1, c1)))
        self.downc1 = _Blk(*_cr(c1, c2, stride=2), _Blk(*_cr(c2, c2), *_cr(c2, c2)))
        self.upc2 = _Blk(_Blk(*_cr(c2, c2), *_cr(c2, c2)), nn.Conv2d(c2, c1 * 4, 3, padding=1, bias=False), nn.PixelShuffle(2))
        self.upc1 = _Blk(_Blk(*_cr(c1, c1), *_cr(c1, c1)), nn.Conv2d(c1, c0 * 4, 3, padding=1, bias=False), nn.PixelShuffle(2))
        self.outc = _Blk(*_cr(c0, c0), nn.Conv2d(c0, 4 if bayer4 else 3, 3, padding=1, bias=False))
        if bayer4:
            self.fusion = _Blk(*_cr(4, 4), nn.Conv2d(4, 3, 3, padding=1, bias=False))


class DDnet(nn.Module):
    """Parameter container (state-dict compatible with the reference's DDnet)."""

    def __init__(self, num_input_frames=5):
        super().__init__()
        if num_input_frames != 5:
            raise ValueError('only the 5-frame DDnet of the reference is supported')
        self.num_input_frames = num_input_frames
        self.temp1 = DDDenBlock(3, 1)
        self.temp2 = DDDenBlock(3, 3)
        self.temp11 = DDDenBlock(3, 4, bayer4=True)
        self.weight_tensor_in = nn.Parameter(torch.ones((9, 1, 1, 1, 1)))
        self.weight_tensor_in2 = nn.Parameter(torch.ones((9, 1, 4, 1, 1)))
        self.weight_tensor_out = nn.Parameter(torch.ones((2, 1, 3, 1, 1)))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, nonlinearity='relu')


def _p8(c):
    return (c + 7) // 8 * 8


C0, C1, C2, CI = _p8(BASE), 2 * BASE, 4 * BASE, 96          # 24, 40, 80, 96 (90 padded)

# (key, Cin padded, Cout padded, relu, stride2, shuffle); the first entry is the grouped input conv
_LAYERS = [
    ('inc_1.convblock.0', None, CI, True, False, False),
    ('inc_1.convblock.2', CI, C0, True, False, False),
    ('downc0.convblock.0', C0, C1, True, True, False),
    ('downc0.convblock.2.convblock.0', C1, C1, True, False, False),
    ('downc0.convblock.2.convblock.2', C1, C1, True, False, False),
    ('downc1.convblock.0', C1, C2, True, True, False),
    ('downc1.convblock.2.convblock.0', C2, C2, True, False, False),
    ('downc1.convblock.2.convblock.2', C2, C2, True, False, False),
    ('upc2.convblock.0.convblock.0', C2, C2, True, False, False),
    ('upc2.convblock.0.convblock.2', C2, C2, True, False, False),
    ('upc2.convblock.1', C2, 4 * C1, False, False, True),
    ('upc1.convblock.0.convblock.0', C1, C1, True, False, False),
    ('upc1.convblock.0.convblock.2', C1, C1, True, False, False),
    ('upc1.convblock.1', C1, 4 * C0, False, False, True),          # 80 real conv channels in 96: 20 of 24 after the shuffle
    ('outc.convblock.0', C0, C0, True, False, False),
    ('outc.convblock.2', C0, 8, False, False, False),
]


def _strip(sd):
    return {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}


def _pack(w, cin, cout, device, split):
    buf = ops.packed_buffer(cin, cout, device, split)
    if split:
        return ops.pack_conv3x3_split_device(w, None, buf, cin, cout)
    return ops.pack_conv3x3_device(w, None, buf, cin, cout)


def pack_block(sd, prefix, ch_each_frame, device, split):
    """one upload of the block's weights, packing on the device"""
    device = torch.device(device)
    keys = [f'{prefix}.{key}.weight' for key, *_r in _LAYERS]
    fusion = f'{prefix}.fusion.convblock.0.weight' in sd
    if fusion:
        keys += [f'{prefix}.fusion.convblock.0.weight', f'{prefix}.fusion.convblock.2.weight']
    dev = ops.device_params([sd[k] for k in keys], device)
    packed = []
    for (key, cin, cout, _r, _s, _sh), w in zip(_LAYERS, dev):
        if cin is None:
            dense = torch.zeros(w.shape[0], w.shape[1] * 3, 3, 3, dtype=w.dtype, device=device)
            per = w.shape[0] // 3
            for g in range(3):                                   # grouped -> block-diagonal (data movement only)
                dense[g * per:(g + 1) * per, g * w.shape[1]:(g + 1) * w.shape[1]] = w[g * per:(g + 1) * per]
            w, cin = dense, _p8(3 * ch_each_frame)
        packed.append(_pack(w.contiguous(), cin, cout, device, split))
    if fusion:
        packed.append(_pack(dev[-2].contiguous(), 8, 8, device, split))
        packed.append(_pack(dev[-1].contiguous(), 8, 8, device, split))
    return packed


def wino_packs(packed, ch_each_frame):
    """Winograd-domain weights (csrc/conv_wino.hip) of a block's stride-1 layers (PixelShuffle layers included) and of the two
    fusion convolutions, if present, None for the stride-2 ones; derived on the device from the fp32 direct packing"""
    out = []
    for i, pk in enumerate(packed):
        if i < len(_LAYERS):
            _k, cin, cout, _r, s2, shuf = _LAYERS[i]
            cin = _p8(3 * ch_each_frame) if cin is None else cin
            out.append(None if (s2 or (shuf and cout % 32)) else ops.pack_conv3x3_wino_both(pk, cin, cout))
        else:
            out.append(ops.pack_conv3x3_wino(pk, 8, 8))
    return out


class _Bufs:
    """activation buffers of one DenBlock pass over E evaluations at (h, w); dead buffers are reused."""

    def __init__(self, E, cin, h, w, device, split):
        def act(c, hh, ww):
            if split:
                return torch.empty(E, c // 8, 2, hh, ww, 8, dtype=torch.float16, device=device)
            return torch.empty(E, c // 8, hh, ww, 8, dtype=torch.float32, device=device)

        def f32(c, hh, ww):
            return torch.empty(E, c // 8, hh, ww, 8, dtype=torch.float32, device=device)
        h2, w2, h4, w4 = h // 2, w // 2, h // 4, w // 4
        self.t_in, self.t96 = act(cin, h, w), act(CI, h, w)
        self.x0, self.s0, self.o0 = act(C0, h, w), act(C0, h, w), act(C0, h, w)
        a = [act(C1, h2, w2) for _ in range(3)]
        d = [act(C2, h4, w4) for _ in range(3)]
        self.a0, self.a1, self.x1, self.s1, self.c0, self.c1 = a[0], a[1], a[0], a[1], a[2], a[1]
        self.d0, self.d1, self.x2, self.u0, self.u1 = d[0], d[1], d[0], d[1], d[2]
        self.x8 = f32(8, h, w)


def unet_forward(pk, b, split, pkw=None):
    """the U-Net body of a DenBlock from the packed input b.t_in to the 8-channel fp32 tail b.x8
    (reference models/network_demosaicking.py:223-238); the evaluations are independent and run as two half-batches on
    two HIP streams (ops.on_side_streams, SCIPNP_STREAMS) like the FastDVDnet DenBlocks."""
    ops.on_side_streams(b.t_in.shape[0], lambda sl: _unet_convs(pk, _View(b, sl), split, pkw))
    return b.x8


def _unet_convs(pk, b, split, pkw=None):
    if split:
        c = ops.conv3x3_c8s
        c(b.t_in, pk[0], CI, relu=True, out=b.t96)
        c(b.t96, pk[1], C0, relu=True, out=b.x0)
        c(b.x0, pk[2], C1, relu=True, stride2=True, out=b.a0)
        c(b.a0, pk[3], C1, relu=True, out=b.a1)
        c(b.a1, pk[4], C1, relu=True, out=b.x1)
        c(b.x1, pk[5], C2, relu=True, stride2=True, out=b.d0)
        c(b.d0, pk[6], C2, relu=True, out=b.d1)
        c(b.d1, pk[7], C2, relu=True, out=b.x2)
        c(b.x2, pk[8], C2, relu=True, out=b.u0)
        c(b.u0, pk[9], C2, relu=True, out=b.u1)
        c(b.u1, pk[10], 4 * C1, shuffle=True, residual=b.x1, out=b.s1)
        c(b.s1, pk[11], C1, relu=True, out=b.c0)
        c(b.c0, pk[12], C1, relu=True, out=b.c1)
        c(b.c1, pk[13], 4 * C0, shuffle=True, residual=b.x0, out=b.s0)
        c(b.s0, pk[14], C0, relu=True, out=b.o0)
        c(b.o0, pk[15], 8, out=b.x8, f32_out=True)
    else:
        def c(x, w, cout, **kw):                       # fp32: the stride-1 layers in Winograd form where pkw holds them
            i = next(j for j, p_ in enumerate(pk) if p_ is w)
            if pkw is None or pkw[i] is None:
                return ops.conv3x3_c8(x, w, cout, **kw)
            return ops.conv3x3_c8w(x, pkw[i], cout, **kw)
        c(b.t_in, pk[0], CI, relu=True, out=b.t96)
        c(b.t96, pk[1], C0, relu=True, out=b.x0)
        c(b.x0, pk[2], C1, relu=True, stride2=True, out=b.a0)
        c(b.a0, pk[3], C1, relu=True, out=b.a1)
        c(b.a1, pk[4], C1, relu=True, out=b.x1)
        c(b.x1, pk[5], C2, relu=True, stride2=True, out=b.d0)
        c(b.d0, pk[6], C2, relu=True, out=b.d1)
        c(b.d1, pk[7], C2, relu=True, out=b.x2)
        c(b.x2, pk[8], C2, relu=True, out=b.u0)
        c(b.u0, pk[9], C2, relu=True, out=b.u1)
        c(b.u1, pk[10], 4 * C1, shuffle=True, residual=b.x1, out=b.s1)
        c(b.s1, pk[11], C1, relu=True, out=b.c0)
        c(b.c0, pk[12], C1, relu=True, out=b.c1)
        c(b.c1, pk[13], 4 * C0, shuffle=True, residual=b.x0, out=b.s0)
        c(b.s0, pk[14], C0, relu=True, out=b.o0)
        c(b.o0, pk[15], 8, out=b.x8)
    return b.x8


class DDnetEngine:
    def __init__(self, model, B, H, W, device, precision=None, units=1):
        """B frames in all; units > 1: a unit batch of `units` sequences of B / units frames (frame t of unit u at t * units + u):
        the temporal windows of stage 1 stay inside a unit -- only the gather tables change"""
        from .nets import default_precision
        if B % units:
            raise ValueError(f'{B} frames do not split into {units} units')
        self.units = units
        if H % 8 or W % 8:
            raise ValueError('DDnet on the HIP path needs H and W to be multiples of 8 (half-resolution U-Net)')
        self.B, self.H, self.W, self.device = B, H, W, device
        self.precision = precision or default_precision()
        self.split = self.precision == 'f16x3'
        h, w = H // 2, W // 2
        E = 3 * B
        n = torch.arange(B)
        # stage 1: evaluation e = j*B + n uses frames (n - 2 + j + i) mod B, i = 0..2  (DDnet_test.py:177-179 window,
        # network_demosaicking.py:441-449 triplets)
        Bu = B // units
        tt, uu = n // units, n % units
        idx1 = torch.stack([torch.stack([((tt - 2 + j + i) % Bu) * units + uu for i in range(3)], 1) for j in range(3)]).reshape(E, 3)
        # stage 2: evaluation n (branch a) / B + n (branch b) uses stage-1 outputs j*B + n of its branch
        idx2 = torch.cat([torch.stack([j * B + n for j in range(3)], 1) + br * E for br in range(2)])
        self.idx1 = idx1.to(torch.int32).contiguous().to(device)
        self.idx2 = idx2.to(torch.int32).contiguous().to(device)
        self.bufs_full = _Bufs(E, 16, H, W, device, self.split)        # temp1 (8 ch in) and temp2 (16 ch in) share these
        self.t_in16 = self.bufs_full.t_in
        self.bufs_half = _Bufs(E, 16, h, w, device, self.split)
        self.t_in8 = (torch.empty(E, 1, 2, H, W, 8, dtype=torch.float16, device=device) if self.split else
                      torch.empty(E, 1, H, W, 8, dtype=torch.float32, device=device))
        self.fu_in = torch.empty_like(self.t_in8)
        self.fu_mid = torch.empty_like(self.t_in8)
        self.p4 = torch.empty(E, 4, h, w, dtype=torch.float32, device=device)
        self.s1 = torch.empty(2 * E, 3, H, W, dtype=torch.float32, device=device)      # stage-1 outputs, branch a then b
        self.s2 = torch.empty(2 * B, 3, H, W, dtype=torch.float32, device=device)
        self.refresh(model)

    def refresh(self, model):
        sd = _strip(model.state_dict())
        dev, sp = self.device, self.split
        self.pk1 = pack_block(sd, 'temp1', 1, dev, sp)
        self.pk2 = pack_block(sd, 'temp2', 3, dev, sp)
        self.pk11 = pack_block(sd, 'temp11', 4, dev, sp)
        from .nets import f32_conv_form
        self.pkw1 = self.pkw2 = self.pkw11 = None
        if not sp and f32_conv_form(self.H, self.W) == 'winograd':
            self.pkw1, self.pkw2, self.pkw11 = wino_packs(self.pk1, 1), wino_packs(self.pk2, 3), wino_packs(self.pk11, 4)
        B = self.B
        a = sd['weight_tensor_in'].detach().float().reshape(3, 3, 1)           # [j][i][c]
        a2 = sd['weight_tensor_in2'].detach().float().reshape(3, 3, 4)
        self.scale1 = a[:, None].expand(3, B, 3, 1).reshape(3 * B, 3, 1).contiguous().to(dev)
        self.scale11 = a2[:, None].expand(3, B, 3, 4).reshape(3 * B, 3, 4).contiguous().to(dev)
        self.gates = sd['weight_tensor_out'].detach().float().reshape(2, 3).contiguous().to(dev)

    def forward(self, planes, mosaic, out):
        """planes (B,4,H/2,W/2) Bayer planes and mosaic (B,H,W) of the same cube -> out (B,3,H,W) demosaicked frames."""
        B, H, W, E = self.B, self.H, self.W, 3 * self.B
        h, w = H // 2, W // 2
        bf, bh, sp = self.bufs_full, self.bufs_half, self.split
        # stage 1a: mosaic DenBlocks
        bf.t_in = self.t_in8
        ops.ddnet_gather(mosaic, self.idx1, self.scale1, bf.t_in, 1, H, W)
        unet_forward(self.pk1, bf, sp, self.pkw1)
        ops.ddnet_finish(mosaic, self.idx1, self.scale1, bf.x8, self.s1[:E], 1, 3, H, W)
        # stage 1b: Bayer-plane DenBlocks at half resolution, bilinear x2, fusion
        ops.ddnet_gather(planes, self.idx1, self.scale11, bh.t_in, 4, h, w)
        unet_forward(self.pk11, bh, sp, self.pkw11)
        ops.ddnet_finish(planes, self.idx1, self.scale11, bh.x8, self.p4, 4, 4, h, w)
        ops.bilinear_up2_c8(self.p4, self.fu_in)
        if sp:
            ops.conv3x3_c8s(self.fu_in, self.pk11[16], 8, relu=True, out=self.fu_mid)
            ops.conv3x3_c8s(self.fu_mid, self.pk11[17], 8, out=bf.x8, f32_out=True)
        elif self.pkw11 is not None:
            ops.conv3x3_c8w(self.fu_in, self.pkw11[16], 8, relu=True, out=self.fu_mid)
            ops.conv3x3_c8w(self.fu_mid, self.pkw11[17], 8, out=bf.x8)
        else:
            ops.conv3x3_c8(self.fu_in, self.pk11[16], 8, relu=True, out=self.fu_mid)
            ops.conv3x3_c8(self.fu_mid, self.pk11[17], 8, out=bf.x8)
        ops.ddnet_finish(None, None, None, bf.x8, self.s1[E:], 3, 3, H, W)
        # stage 2 on both branches (2B evaluations of temp2)
        bf.t_in = self.t_in16
        ops.ddnet_gather(self.s1, self.idx2, None, bf.t_in[:2 * B], 3, H, W)
        unet_forward(self.pk2, _View(bf, 2 * B), sp, self.pkw2)
        ops.ddnet_finish(self.s1, self.idx2, None, bf.x8[:2 * B], self.s2, 3, 3, H, W)
        return ops.ddnet_mix(self.s2, self.gates, out)


class _View:
    """view of a _Bufs on the evaluations n (first n) or a slice of them (a leading-dimension slice keeps contiguity)."""

    def __init__(self, b, n):
        sl = n if isinstance(n, slice) else slice(0, n)
        for k, t in vars(b).items():
            if isinstance(t, torch.Tensor):
                setattr(self, k, t[sl])

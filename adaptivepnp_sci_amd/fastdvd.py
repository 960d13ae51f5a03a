"""FastDVDnet on the HIP kernels.

`FastDVDnet` is a parameter container with the reference's state-dict keys
(packages/fastdvdnet/models.py:146-253: temp{1,2}.{inc,downc0,downc1,upc2,upc1,outc}.convblock.*); wrap it
in nn.DataParallel like the reference driver does (two_stage_ADMM_Online_FastDVD_Warm.py:240-241) or not --
the engine strips an optional `module.` prefix.

`FastDVDEngine.forward(frames, sigma)` = packages/fastdvdnet/fastdvdnet.py:82-146 (sliding 5-frame window with
circular temporal indexing) + models.py:227-251 (three stage-1 DenBlocks + one stage-2 DenBlock per output
frame).  With circular indexing the 3*B stage-1 evaluations of the reference are only B distinct ones
(triplets centred on each frame): each is computed once, all B at a time, then stage 2 runs on the B triplets
of stage-1 outputs -- bit-identical to the reference's 4*B DenBlock calls.

Per DenBlock: 16 conv3x3 launches (BatchNorm folded into the packed weights, ReLU / skip-add / PixelShuffle /
stride 2 fused into the conv kernel) + pack + residual.
"""
import torch
import torch.nn as nn

from . import ops

_BN_EPS = 1e-5


def _cbr(cin, cout, stride=1, groups=1):
    return [nn.Conv2d(cin, cout, 3, stride=stride, padding=1, groups=groups, bias=False),
            nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]


class _Blk(nn.Module):
    def __init__(self, *mods):
        super().__init__()
        self.convblock = nn.Sequential(*mods)


class DenBlock(nn.Module):
    def __init__(self, num_input_frames=3, ncolor=3):
        super().__init__()
        f = num_input_frames
        self.inc = _Blk(*_cbr(f * (ncolor + 1), f * 30, groups=f), *_cbr(f * 30, 32))
        self.downc0 = _Blk(*_cbr(32, 64, stride=2), _Blk(*_cbr(64, 64), *_cbr(64, 64)))
        self.downc1 = _Blk(*_cbr(64, 128, stride=2), _Blk(*_cbr(128, 128), *_cbr(128, 128)))
        self.upc2 = _Blk(_Blk(*_cbr(128, 128), *_cbr(128, 128)), nn.Conv2d(128, 256, 3, padding=1, bias=False),
                         nn.PixelShuffle(2))
        self.upc1 = _Blk(_Blk(*_cbr(64, 64), *_cbr(64, 64)), nn.Conv2d(64, 128, 3, padding=1, bias=False),
                         nn.PixelShuffle(2))
        self.outc = _Blk(*_cbr(32, 32), nn.Conv2d(32, ncolor, 3, padding=1, bias=False))


class FastDVDnet(nn.Module):
    """Parameter container (Kaiming-normal init like the reference, models.py:213-220)."""

    def __init__(self, num_input_frames=5, num_color_channels=3):
        super().__init__()
        if num_input_frames != 5 or num_color_channels != 3:
            raise ValueError('only the 5-frame colour FastDVDnet of the reference is supported')
        self.num_input_frames = num_input_frames
        self.num_color_channels = num_color_channels
        self.temp1 = DenBlock(3, num_color_channels)
        self.temp2 = DenBlock(3, num_color_channels)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, nonlinearity='relu')

    def forward(self, frames, sigma):
        """frames (B,3,H,W) CUDA tensor of a whole sequence -> denoised (B,3,H,W); HIP kernels only."""
        B, _, H, W = frames.shape
        return FastDVDEngine(self, B, H, W, frames.device).forward(frames.float().contiguous(), float(sigma))


# (key prefix, bn key or None, Cin, Cout, relu, stride2, shuffle)
_LAYERS = [
    ('inc.convblock.0', 'inc.convblock.1', 16, 96, True, False, False),          # grouped 12->90 as dense 16->96
    ('inc.convblock.3', 'inc.convblock.4', 96, 32, True, False, False),
    ('downc0.convblock.0', 'downc0.convblock.1', 32, 64, True, True, False),
    ('downc0.convblock.3.convblock.0', 'downc0.convblock.3.convblock.1', 64, 64, True, False, False),
    ('downc0.convblock.3.convblock.3', 'downc0.convblock.3.convblock.4', 64, 64, True, False, False),
    ('downc1.convblock.0', 'downc1.convblock.1', 64, 128, True, True, False),
    ('downc1.convblock.3.convblock.0', 'downc1.convblock.3.convblock.1', 128, 128, True, False, False),
    ('downc1.convblock.3.convblock.3', 'downc1.convblock.3.convblock.4', 128, 128, True, False, False),
    ('upc2.convblock.0.convblock.0', 'upc2.convblock.0.convblock.1', 128, 128, True, False, False),
    ('upc2.convblock.0.convblock.3', 'upc2.convblock.0.convblock.4', 128, 128, True, False, False),
    ('upc2.convblock.1', None, 128, 256, False, False, True),
    ('upc1.convblock.0.convblock.0', 'upc1.convblock.0.convblock.1', 64, 64, True, False, False),
    ('upc1.convblock.0.convblock.3', 'upc1.convblock.0.convblock.4', 64, 64, True, False, False),
    ('upc1.convblock.1', None, 64, 128, False, False, True),
    ('outc.convblock.0', 'outc.convblock.1', 32, 32, True, False, False),
    ('outc.convblock.3', None, 32, 8, False, False, False),
]


def _strip(sd):
    return {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}


def pack_denblock(sd, prefix, device, split=False):
    """Packed weights of one DenBlock from a (stripped) state dict; eval-mode BatchNorm folded:
    y = conv(x) * gamma/sqrt(var+eps) + (beta - mean*gamma/sqrt(var+eps)).  The parameters travel to the device in one
    upload; the fold (scipnp_bn_fold) and the packing (scipnp_pack_conv3x3_device_scaled / _split_device_scaled) run
    there.  split=True packs for the error-compensated split-fp16 kernels (conv_split.hip)."""
    device = torch.device(device)
    names = []
    for key, bn, *_r in _LAYERS:
        names.append(f'{prefix}.{key}.weight')
        if bn is not None:
            names += [f'{prefix}.{bn}.weight', f'{prefix}.{bn}.bias', f'{prefix}.{bn}.running_mean', f'{prefix}.{bn}.running_var']
    dev = dict(zip(names, ops.device_params([sd[k] for k in names], device)))
    packed = []
    for key, bn, cin, cout, _relu, _s2, _shuf in _LAYERS:
        w = dev[f'{prefix}.{key}.weight']
        if key == 'inc.convblock.0':
            dense = torch.zeros(w.shape[0], w.shape[1] * 3, 3, 3, dtype=w.dtype, device=device)
            per = w.shape[0] // 3
            for g in range(3):                                   # grouped -> block-diagonal (data movement only)
                dense[g * per:(g + 1) * per, g * w.shape[1]:(g + 1) * w.shape[1]] = w[g * per:(g + 1) * per]
            w = dense
        scale = shift = None
        if bn is not None:
            scale, shift = torch.empty_like(dev[f'{prefix}.{bn}.weight']), torch.empty_like(dev[f'{prefix}.{bn}.weight'])
            ops.bn_fold(dev[f'{prefix}.{bn}.weight'], dev[f'{prefix}.{bn}.bias'], dev[f'{prefix}.{bn}.running_mean'],
                        dev[f'{prefix}.{bn}.running_var'], _BN_EPS, scale, shift)
        buf = ops.packed_buffer(cin, cout, device, split)
        if split:
            packed.append(ops.pack_conv3x3_split_device(w, shift, buf, cin, cout, scale=scale))
        else:
            packed.append(ops.pack_conv3x3_device(w, shift, buf, cin, cout, scale=scale))
    return packed


BUF_KEYS = ('t_in', 't96', 'x0', 'a0', 'a1', 'x1', 'd0', 'd1', 'x2', 'u0', 'u1', 's64', 'c0', 'c1', 's32', 'o32', 'x8')


def alloc_denblock_buffers(B, H, W, device, alias=True):
    """c8 activation buffers of one DenBlock evaluation on B triplets.  alias=True reuses dead buffers
    (inference); alias=False keeps every activation (the finetune's backward pass needs them)."""
    f = lambda c, h, w: torch.empty(B, c // 8, h, w, 8, dtype=torch.float32, device=device)  # noqa: E731
    H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
    b = dict(t_in=f(16, H, W), t96=f(96, H, W), x0=f(32, H, W), s32=f(32, H, W), o32=f(32, H, W), x8=f(8, H, W))
    if alias:
        a = [f(64, H2, W2) for _ in range(3)]
        d = [f(128, H4, W4) for _ in range(3)]
        b.update(a0=a[0], a1=a[1], x1=a[0], d0=d[0], d1=d[1], x2=d[0], u0=d[1], u1=d[2], s64=a[1], c0=a[2], c1=a[1])
    else:
        for k in ('a0', 'a1', 'x1', 's64', 'c0', 'c1'):
            b[k] = f(64, H2, W2)
        for k in ('d0', 'd1', 'x2', 'u0', 'u1'):
            b[k] = f(128, H4, W4)
    return b


def wino_packs(packed):
    """Winograd-domain weights (csrc/conv_wino.hip) of the stride-1 layers of one DenBlock (the PixelShuffle layers
    included: the shuffle is an epilogue of that kernel too), None for the stride-2 ones; derived on the device from the
    fp32 direct packing"""
    # (layers of at least 16 input and 32 output channels also get the F(4x4,3x3) packing of csrc/conv_wino4.hip, which conv3x3_c8w
    # then prefers -- the PixelShuffle store is an epilogue of both kernels)
    return [None if s2 else ops.pack_conv3x3_wino_both(packed[i], cin, cout)
            for i, (_k, _bn, cin, cout, _relu, s2, _shuf) in enumerate(_LAYERS)]


def denblock_forward(pk, frames, sigma, out, b, pkw=None, units=1):
    """out[n] = DenBlock(frames[n-1], frames[n], frames[n+1]) for all n (circular); pk = 16 packed layers,
    b = buffers from alloc_denblock_buffers; pkw = wino_packs(pk) to run the stride-1 layers as fp32 Winograd.
    reference packages/fastdvdnet/models.py:179-198."""
    if pkw is not None:
        direct, full = ops.conv3x3_c8, pk

        def c(x, w, cout, **kw):                       # w: the direct packing; its layer index selects the form
            i = next(j for j, p_ in enumerate(full) if p_ is w)
            if pkw[i] is None:
                return direct(x, w, cout, **kw)
            return ops.conv3x3_c8w(x, pkw[i], cout, **kw)
    else:
        c = ops.conv3x3_c8
    ops.fastdvd_pack_triplets(frames, sigma, b['t_in'], units=units)

    def convs(b):
        c(b['t_in'], pk[0], 96, relu=True, out=b['t96'], head=True)
        c(b['t96'], pk[1], 32, relu=True, out=b['x0'])
        c(b['x0'], pk[2], 64, relu=True, stride2=True, out=b['a0'])
        c(b['a0'], pk[3], 64, relu=True, out=b['a1'])
        c(b['a1'], pk[4], 64, relu=True, out=b['x1'])
        c(b['x1'], pk[5], 128, relu=True, stride2=True, out=b['d0'])
        c(b['d0'], pk[6], 128, relu=True, out=b['d1'])
        c(b['d1'], pk[7], 128, relu=True, out=b['x2'])
        c(b['x2'], pk[8], 128, relu=True, out=b['u0'])
        c(b['u0'], pk[9], 128, relu=True, out=b['u1'])
        c(b['u1'], pk[10], 256, shuffle=True, residual=b['x1'], out=b['s64'])     # x1 + upc2(x2)
        c(b['s64'], pk[11], 64, relu=True, out=b['c0'])
        c(b['c0'], pk[12], 64, relu=True, out=b['c1'])
        c(b['c1'], pk[13], 128, shuffle=True, residual=b['x0'], out=b['s32'])     # x0 + upc1(.)
        c(b['s32'], pk[14], 32, relu=True, out=b['o32'])
        c(b['o32'], pk[15], 8, out=b['x8'])

    # two half-batches of frames on two HIP streams, as denblock_forward_split below
    ops.on_side_streams(b['t_in'].shape[0], lambda sl: convs({k: v[sl] for k, v in b.items()}))
    return ops.fastdvd_finish(frames, b['x8'], out)


def alloc_denblock_buffers_split(B, H, W, device, alias=True):
    """buffers of the split-fp16 DenBlock: c8s activations (float16, same bytes as fp32 c8) plus the two fp32
    8-channel tail.  alias=False keeps every activation (finetune stash)."""
    h16 = lambda c, h, w: torch.empty(B, c // 8, 2, h, w, 8, dtype=torch.float16, device=device)  # noqa: E731
    f32 = lambda c, h, w: torch.empty(B, c // 8, h, w, 8, dtype=torch.float32, device=device)  # noqa: E731
    H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
    b = dict(t_in=h16(16, H, W), t96=h16(96, H, W), x0=h16(32, H, W), s32=h16(32, H, W), o32=h16(32, H, W),
             x8=f32(8, H, W))
    if alias:
        a = [h16(64, H2, W2) for _ in range(3)]
        d = [h16(128, H4, W4) for _ in range(3)]
        b.update(a0=a[0], a1=a[1], x1=a[0], d0=d[0], d1=d[1], x2=d[0], u0=d[1], u1=d[2], s64=a[1], c0=a[2], c1=a[1])
    else:
        for k in ('a0', 'a1', 'x1', 's64', 'c0', 'c1'):
            b[k] = h16(64, H2, W2)
        for k in ('d0', 'd1', 'x2', 'u0', 'u1'):
            b[k] = h16(128, H4, W4)
    return b


def _denblock_convs_split(pk, b):
    c = ops.conv3x3_c8s
    c(b['t_in'], pk[0], 96, relu=True, out=b['t96'], head=True)
    c(b['t96'], pk[1], 32, relu=True, out=b['x0'])
    c(b['x0'], pk[2], 64, relu=True, stride2=True, out=b['a0'])
    c(b['a0'], pk[3], 64, relu=True, out=b['a1'])
    c(b['a1'], pk[4], 64, relu=True, out=b['x1'])
    c(b['x1'], pk[5], 128, relu=True, stride2=True, out=b['d0'])
    c(b['d0'], pk[6], 128, relu=True, out=b['d1'])
    c(b['d1'], pk[7], 128, relu=True, out=b['x2'])
    c(b['x2'], pk[8], 128, relu=True, out=b['u0'])
    c(b['u0'], pk[9], 128, relu=True, out=b['u1'])
    c(b['u1'], pk[10], 256, shuffle=True, residual=b['x1'], out=b['s64'])      # x1 + upc2(x2), PixelShuffle fused
    c(b['s64'], pk[11], 64, relu=True, out=b['c0'])
    c(b['c0'], pk[12], 64, relu=True, out=b['c1'])
    c(b['c1'], pk[13], 128, shuffle=True, residual=b['x0'], out=b['s32'])      # x0 + upc1(.)
    c(b['s32'], pk[14], 32, relu=True, out=b['o32'])
    c(b['o32'], pk[15], 8, out=b['x8'], f32_out=True)


def denblock_forward_split(pk, frames, sigma, out, b, units=1):
    """denblock_forward on the split-fp16 kernels: c8s activations; the two UpBlock convs store their PixelShuffle-d
    result plus the skip tensor straight into c8s (epilogue flag bit6).  The 16 convolutions run as two half-batches of
    frames on two HIP streams (SCIPNP_STREAMS=1 keeps one): the quarter- and half-resolution layers are grids of 1.3 - 2.7
    generations of workgroups, and the second stream's launches fill the CUs the first one's last generation leaves idle."""
    ops.fastdvd_pack_triplets_c8s(frames, sigma, b['t_in'], units=units)
    ops.on_side_streams(b['t_in'].shape[0], lambda sl: _denblock_convs_split(pk, {k: v[sl] for k, v in b.items()}))
    return ops.fastdvd_finish(frames, b['x8'], out)


class FastDVDEngine:
    def __init__(self, model, B, H, W, device, precision=None, units=1):
        """B frames in all; units > 1: a unit batch of `units` sequences of B / units frames, frame t of unit u at t * units + u --
        the temporal windows stay inside a unit (scipnp_fastdvd_pack_triplets_units), everything else is per frame"""
        if B % units:
            raise ValueError(f'{B} frames do not split into {units} units')
        self.units = units
        if H % 4 or W % 4:
            raise ValueError('FastDVDnet needs H and W to be multiples of 4 (the reference reflect-pads otherwise; '
                             'its padding of the noise map breaks for more than one frame, fastdvdnet.py:126)')
        from .nets import default_precision
        self.B, self.H, self.W, self.device = B, H, W, device
        self.precision = precision or default_precision()
        self.refresh(model)
        self.bufs = (alloc_denblock_buffers_split(B, H, W, device) if self.precision == 'f16x3' else
                     alloc_denblock_buffers(B, H, W, device, alias=True))
        self.s1 = torch.empty(B, 3, H, W, dtype=torch.float32, device=device)
        self.out = torch.empty_like(self.s1)

    def refresh(self, model):
        sd = _strip(model.state_dict())
        self.set_packed({p: pack_denblock(sd, p, self.device, split=self.precision == 'f16x3') for p in ('temp1', 'temp2')})

    def set_packed(self, packed):
        """install packed weights {'temp1': [...16], 'temp2': [...]} (refresh, or the finetune's device-packed updated
        weights); the fp32 engine derives the Winograd packs of its stride-1 layers from them"""
        from .nets import f32_conv_form
        self.packed = packed
        self.packed_wino = None
        if self.precision == 'f32' and f32_conv_form(self.H, self.W) == 'winograd':
            self.packed_wino = {p: wino_packs(pk) for p, pk in packed.items()}

    def forward(self, frames, sigma):
        """frames planar (B,3,H,W) -> denoised planar (B,3,H,W) (owned by the engine, overwritten per call)."""
        if self.precision == 'f16x3':
            denblock_forward_split(self.packed['temp1'], frames, sigma, self.s1, self.bufs, units=self.units)
            return denblock_forward_split(self.packed['temp2'], self.s1, sigma, self.out, self.bufs, units=self.units)
        w = self.packed_wino or {'temp1': None, 'temp2': None}
        denblock_forward(self.packed['temp1'], frames, sigma, self.s1, self.bufs, w['temp1'], units=self.units)
        return denblock_forward(self.packed['temp2'], self.s1, sigma, self.out, self.bufs, w['temp2'], units=self.units)

"""Oracle (test infrastructure): PyTorch-CPU definitions of the two plug-in denoiser networks,
state-dict-key compatible with the reference's checkpoints so that the same weights load.

FFDNet  -- reference models/network_ffdnet.py:27-69 (+ models/basicblock.py:61-126): replicate-pad
           to even size, 2x2 pixel-unshuffle (channel order c*4 + dy*2 + dx), noise-level map
           appended as the LAST channel, nb conv3x3(pad 1, bias)+ReLU layers (no ReLU after the
           last), PixelShuffle(2), crop.  Keys: model.{0,2,...,2(nb-1)}.{weight,bias}.
FastDVDnet -- reference packages/fastdvdnet/models.py:16-253: two cascaded 3-frame U-Net DenBlocks.
           Keys: temp{1,2}.{inc,downc0,downc1,upc2,upc1,outc}.convblock.* (prefix `module.` when
           wrapped in nn.DataParallel as the reference driver does).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class OracleFFDNet(nn.Module):
    def __init__(self, in_nc=3, out_nc=3, nc=96, nb=12):
        super().__init__()
        layers = [nn.Conv2d(in_nc * 4 + 1, nc, 3, 1, 1, bias=True), nn.ReLU(inplace=True)]
        for _ in range(nb - 2):
            layers += [nn.Conv2d(nc, nc, 3, 1, 1, bias=True), nn.ReLU(inplace=True)]
        layers += [nn.Conv2d(nc, out_nc * 4, 3, 1, 1, bias=True)]
        self.model = nn.Sequential(*layers)

    def forward(self, x, sigma):
        n, c, h, w = x.shape
        x = F.pad(x, (0, (-w) % 2, 0, (-h) % 2), mode='replicate')
        hh, ww = x.shape[-2] // 2, x.shape[-1] // 2
        x = x.reshape(n, c, hh, 2, ww, 2).permute(0, 1, 3, 5, 2, 4).reshape(n, c * 4, hh, ww)
        x = torch.cat((x, sigma.repeat(1, 1, hh, ww)), 1)
        x = self.model(x)
        x = F.pixel_shuffle(x, 2)
        return x[..., :h, :w]


def _cbr(cin, cout, stride=1, groups=1):
    return [nn.Conv2d(cin, cout, 3, stride=stride, padding=1, groups=groups, bias=False),
            nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]


class _Wrap(nn.Module):
    """Holds an nn.Sequential under the attribute name `convblock` (checkpoint key layout)."""

    def __init__(self, *mods):
        super().__init__()
        self.convblock = nn.Sequential(*mods)

    def forward(self, x):
        return self.convblock(x)


class OracleDenBlock(nn.Module):
    """reference packages/fastdvdnet/models.py:146-198."""

    def __init__(self, num_input_frames=3, ncolor=3):
        super().__init__()
        f = num_input_frames
        self.inc = _Wrap(*_cbr(f * (ncolor + 1), f * 30, groups=f), *_cbr(f * 30, 32))
        self.downc0 = _Wrap(*_cbr(32, 64, stride=2), _Wrap(*_cbr(64, 64), *_cbr(64, 64)))
        self.downc1 = _Wrap(*_cbr(64, 128, stride=2), _Wrap(*_cbr(128, 128), *_cbr(128, 128)))
        self.upc2 = _Wrap(_Wrap(*_cbr(128, 128), *_cbr(128, 128)),
                          nn.Conv2d(128, 256, 3, padding=1, bias=False), nn.PixelShuffle(2))
        self.upc1 = _Wrap(_Wrap(*_cbr(64, 64), *_cbr(64, 64)),
                          nn.Conv2d(64, 128, 3, padding=1, bias=False), nn.PixelShuffle(2))
        self.outc = _Wrap(*_cbr(32, 32), nn.Conv2d(32, ncolor, 3, padding=1, bias=False))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, nonlinearity='relu')

    def forward(self, in0, in1, in2, noise_map):
        x0 = self.inc(torch.cat((in0, noise_map, in1, noise_map, in2, noise_map), dim=1))
        x1 = self.downc0(x0)
        x2 = self.downc1(x1)
        x2 = self.upc2(x2)
        x1 = self.upc1(x1 + x2)
        x = self.outc(x0 + x1)
        return in1 - x


class OracleFastDVDnet(nn.Module):
    """reference packages/fastdvdnet/models.py:200-253."""

    def __init__(self, num_input_frames=5, ncolor=3):
        super().__init__()
        self.num_input_frames = num_input_frames
        self.ncolor = ncolor
        self.temp1 = OracleDenBlock(3, ncolor)
        self.temp2 = OracleDenBlock(3, ncolor)

    def forward(self, x, noise_map):
        C = self.ncolor
        f = [x[:, m * C:m * C + C] for m in range(self.num_input_frames)]
        a = self.temp1(f[0], f[1], f[2], noise_map)
        b = self.temp1(f[1], f[2], f[3], noise_map)
        c = self.temp1(f[2], f[3], f[4], noise_map)
        return self.temp2(a, b, c, noise_map)


def synth_fastdvdnet_weights(seed=0):
    """Seeded synthetic FastDVDnet weights (the reference's model.pth is not in the snapshot,
    .MISSING_LARGE_BLOBS): Kaiming-normal convs (activations stay O(1) through the U-Net), the last
    conv of each DenBlock scaled by 0.05 so the predicted residual is small and the PnP loop stays
    bounded, BatchNorm affine/running statistics randomised so that the BN fold is exercised."""
    g = torch.Generator().manual_seed(seed)
    net = OracleFastDVDnet()
    sd = net.state_dict()
    for k, v in sd.items():
        if k.endswith('num_batches_tracked'):
            continue
        if v.dim() == 4:
            fan_in = v.shape[1] * 9
            last = 0.05 if k.endswith('outc.convblock.3.weight') else 1.0
            sd[k] = torch.randn(v.shape, generator=g) * (last * (2.0 / fan_in) ** 0.5)
        elif k.endswith('running_var'):
            sd[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith('running_mean'):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith('weight'):
            sd[k] = 0.75 + 0.5 * torch.rand(v.shape, generator=g)
        elif k.endswith('bias'):
            sd[k] = 0.05 * torch.randn(v.shape, generator=g)
    net.load_state_dict(sd)
    return net


# ----------------------------------------------------------------------------- DDnet (deep demosaicking)
def _cr(cin, cout, stride=1, groups=1):
    return [nn.Conv2d(cin, cout, 3, stride=stride, padding=1, groups=groups, bias=False), nn.ReLU(inplace=True)]


class OracleDDDenBlock(nn.Module):
    """DenBlock of the demosaicking network: the FastDVDnet U-Net without BatchNorm, base width 20, residual ADDED
    (reference models/network_demosaicking.py:186-244; 4-channel Bayer variant :310-379 when `bayer4`).  The
    unused `inc` block (noise-map input) is kept so that state-dict keys match the reference."""
    BASE = 20

    def __init__(self, num_input_frames=3, ch_each_frame=3, bayer4=False):
        super().__init__()
        f, c0 = num_input_frames, self.BASE
        c1, c2 = 2 * c0, 4 * c0
        self.inc = _Wrap(*_cr(f * 4, f * 30, groups=f), *_cr(f * 30, c0))
        self.inc_1 = _Wrap(*_cr(f * ch_each_frame, f * 30, groups=f), *_cr(f * 30, c0))
        self.downc0 = _Wrap(*_cr(c0, c1, stride=2), _Wrap(*_cr(c1, c1), *_cr(c1, c1)))
        self.downc1 = _Wrap(*_cr(c1, c2, stride=2), _Wrap(*_cr(c2, c2), *_cr(c2, c2)))
        self.upc2 = _Wrap(_Wrap(*_cr(c2, c2), *_cr(c2, c2)), nn.Conv2d(c2, c1 * 4, 3, padding=1, bias=False),
                          nn.PixelShuffle(2))
        self.upc1 = _Wrap(_Wrap(*_cr(c1, c1), *_cr(c1, c1)), nn.Conv2d(c1, c0 * 4, 3, padding=1, bias=False),
                          nn.PixelShuffle(2))
        self.bayer4 = bayer4
        self.outc = _Wrap(*_cr(c0, c0), nn.Conv2d(c0, 4 if bayer4 else 3, 3, padding=1, bias=False))
        if bayer4:
            self.upscale = nn.UpsamplingBilinear2d(scale_factor=2)
            self.fusion = _Wrap(*_cr(4, 4), nn.Conv2d(4, 3, 3, padding=1, bias=False))

    def forward(self, in0, in1, in2):
        x0 = self.inc_1(torch.cat((in0, in1, in2), dim=1))
        x1 = self.downc0(x0)
        x2 = self.downc1(x1)
        x2 = self.upc2(x2)
        x1 = self.upc1(x1 + x2)
        x = in1 + self.outc(x0 + x1)
        if self.bayer4:
            x = self.fusion(self.upscale(x))
        return x


class OracleDDnet(nn.Module):
    """reference models/network_demosaicking.py:381-463: per output frame, three mosaic-domain DenBlocks (temp1) and
    three Bayer-plane DenBlocks at half resolution (temp11, bilinear x2 + fusion), each on learnable-scalar-weighted
    frame triplets, then the shared second stage (temp2) on both triples, mixed by weight_tensor_out."""

    def __init__(self, num_input_frames=5):
        super().__init__()
        self.num_input_frames = num_input_frames
        self.temp1 = OracleDDDenBlock(3, 1)
        self.temp2 = OracleDDDenBlock(3, 3)
        self.temp11 = OracleDDDenBlock(3, 4, bayer4=True)
        self.weight_tensor_in = nn.Parameter(torch.ones((9, 1, 1, 1, 1)))
        self.weight_tensor_in2 = nn.Parameter(torch.ones((9, 1, 4, 1, 1)))
        self.weight_tensor_out = nn.Parameter(torch.ones((2, 1, 3, 1, 1)))

    def forward(self, x, noise_map=None):
        from .sci_ops import bayer_split
        a, a2, a3 = self.weight_tensor_in, self.weight_tensor_in2, self.weight_tensor_out
        fr = [torch.sum(x[:, 3 * m:3 * m + 3], dim=1) for m in range(self.num_input_frames)]      # (N,H,W) mosaics
        four = [bayer_split(f.permute(1, 2, 0)).permute(2, 3, 0, 1) for f in fr]                    # (N,4,H/2,W/2)
        one = [f.unsqueeze(1) for f in fr]
        s1 = [self.temp1(one[j] * a[3 * j], one[j + 1] * a[3 * j + 1], one[j + 2] * a[3 * j + 2]) for j in range(3)]
        s1b = [self.temp11(four[j] * a2[3 * j], four[j + 1] * a2[3 * j + 1], four[j + 2] * a2[3 * j + 2])
               for j in range(3)]
        return a3[0] * self.temp2(*s1) + a3[1] * self.temp2(*s1b)


def synth_ddnet_weights(seed=0):
    """Seeded synthetic DDnet weights (the reference's checkpoint is not in the snapshot): Kaiming-normal convs, the
    last conv of each DenBlock scaled down so that the residual stays small, and NON-trivial learnable gate scalars
    (the checkpoint initialises them to 1) so that every weight_tensor_* entry is exercised."""
    g = torch.Generator().manual_seed(seed)
    net = OracleDDnet()
    sd = net.state_dict()
    for k, v in sd.items():
        if v.dim() == 4:
            fan_in = v.shape[1] * 9
            last = 0.05 if k.endswith('outc.convblock.2.weight') else 1.0
            sd[k] = torch.randn(v.shape, generator=g) * (last * (2.0 / fan_in) ** 0.5)
        elif k == 'weight_tensor_out':
            sd[k] = 0.5 + 0.05 * torch.randn(v.shape, generator=g)           # the two branches are averaged
        else:
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
    # the fusion block is not residual: give it "identity, then (R, (G1+G2)/2, B)" centre taps plus noise so that the
    # Bayer-plane branch behaves like a (crude) demosaicker and the PnP loop stays in a meaningful regime
    f0, f2 = sd['temp11.fusion.convblock.0.weight'] * 0.1, sd['temp11.fusion.convblock.2.weight'] * 0.1
    for c in range(4):
        f0[c, c, 1, 1] += 1.0
    f2[0, 0, 1, 1] += 1.0
    f2[1, 1, 1, 1] += 0.5
    f2[1, 2, 1, 1] += 0.5
    f2[2, 3, 1, 1] += 1.0
    sd['temp11.fusion.convblock.0.weight'], sd['temp11.fusion.convblock.2.weight'] = f0, f2
    net.load_state_dict(sd)
    return net


def cpu_data_parallel(module):
    """nn.DataParallel exactly as the reference driver wraps its networks (two_stage_ADMM_Online_FastDVD_Warm.py:240-241),
    but pinned to the CPU: with exactly one visible GPU the constructor moves the wrapped module to cuda:0 and forward()
    scatters the inputs there, which would turn the oracle into a PyTorch-GPU computation on the MI355X boxes.  The wrapper
    keeps the `.module` attribute and the `module.`-prefixed state-dict keys the reference code relies on."""
    dp = nn.DataParallel(module)
    dp.device_ids = []            # forward(): `if not self.device_ids: return self.module(*inputs, **kwargs)`
    dp.module.cpu()
    return dp

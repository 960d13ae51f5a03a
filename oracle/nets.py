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

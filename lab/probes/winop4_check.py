#!/usr/bin/env python3
"""GPU box: the producer / consumer F(4x4,3x3) kernel (scipnp_conv3x3_c8wp, Cout % 64 == 0) against scipnp_conv3x3_c8w4:
bit-identity on ragged shapes and every epilogue, then the times of FastDVDnet's 64 -> 64 / 128 -> 128 layer shapes and of a
4-layer chain (each launch reads what the previous one wrote)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'lab'))
import lablib as diaglib  # noqa: E402  (lab/libscipnp_lab.so)

g = torch.Generator().manual_seed(2)
ok = True
for (n, cin, cout, h, w) in ((1, 8, 64, 8, 64), (2, 16, 64, 13, 70), (1, 64, 64, 37, 131), (3, 32, 128, 20, 64), (1, 128, 128, 9, 9), (2, 64, 64, 64, 64),
                             (1, 24, 192, 16, 128)):
    x = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
    pk = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) * 0.1, torch.randn(cout, generator=g), Cin=cin, Cout=cout, device='cuda')
    p4 = ops.pack_conv3x3_wino4(pk, cin, cout)
    res = ops.to_c8(torch.randn(n, cout, h, w, generator=g).cuda())
    msk = ops.to_c8(torch.randn(n, cout, h, w, generator=g).cuda())
    for kw in ({}, {'relu': True}, {'relu': True, 'residual': res}, {'mask_src': msk, 'residual': res}):
        a = ops.conv3x3_c8w4(x, p4, cout, **kw)
        b = diaglib.conv3x3_c8wp(x, p4, cout, **kw)
        same = bool(torch.equal(a, b))
        ok &= same
        if not same:
            d = (a - b).abs()
            print('MISMATCH', (n, cin, cout, h, w), sorted(kw), 'max', float(d.max()), 'rel', float((a - b).norm() / a.norm()),
                  'bad rows', sorted(set((d.sum(dim=(0, 1, 3, 4)) > 0).nonzero().flatten().tolist()))[:12],
                  'bad groups', sorted(set((d.sum(dim=(0, 2, 3, 4)) > 0).nonzero().flatten().tolist()))[:12])
print('bit-identical to scipnp_conv3x3_c8w4 on every shape and epilogue:', ok)


def timed(f, reps=3, inner=30):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]


for (n, c, h, w) in ((8, 64, 256, 256), (8, 128, 128, 128), (8, 64, 512, 512)):
    x8 = ops.to_c8(torch.randn(n, c, h, w, generator=g).cuda())
    pk = ops.pack_conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.randn(c, generator=g), Cin=c, Cout=c, device='cuda')
    p4 = ops.pack_conv3x3_wino4(pk, c, c)
    bufs = [torch.empty_like(x8) for _ in range(2)]
    flop = 2.0 * 36 / 16 * c * c * h * w * n               # executed F(4x4) matrix work

    def chain(fn):
        cur = x8
        for l in range(4):
            cur = fn(cur, p4, c, relu=True, out=bufs[l & 1])
    t4, tp = timed(lambda: ops.conv3x3_c8w4(x8, p4, c, relu=True, out=bufs[0])), timed(lambda: diaglib.conv3x3_c8wp(x8, p4, c, relu=True, out=bufs[0]))
    c4, cp = timed(lambda: chain(ops.conv3x3_c8w4)) / 4, timed(lambda: chain(diaglib.conv3x3_c8wp)) / 4
    print(f'{c:3d} -> {c:3d} on {n} x {h} x {w}:  isolated  c8w4 {t4:7.1f} us ({flop / t4 / 1e6 / 157.3:.3f})   c8wp {tp:7.1f} us ({flop / tp / 1e6 / 157.3:.3f})'
          f'   |  per layer of a 4-layer chain  c8w4 {c4:7.1f} us   c8wp {cp:7.1f} us ({flop / cp / 1e6 / 157.3:.3f} of the fp32 MFMA peak)')

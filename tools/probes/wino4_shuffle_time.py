import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adaptivepnp_sci_amd import ops
g = torch.Generator().manual_seed(0)
for name, n, cin, cout, h, w in (('64->128 shuffle @256', 8, 64, 128, 256, 256), ('128->256 shuffle @128', 8, 128, 256, 128, 128)):
    x8 = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
    pk = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) * 0.05, torch.randn(cout, generator=g), Cin=cin, Cout=cout, device='cuda')
    p4, p2 = ops.pack_conv3x3_wino4(pk, cin, cout), ops.pack_conv3x3_wino(pk, cin, cout)
    res = torch.randn(n, cout // 32, 2 * h, 2 * w, 8, generator=g).cuda()
    o = torch.empty_like(res)
    for nm, f in (('F(2x2)', lambda: ops.conv3x3_c8w(x8, p2, cout, residual=res, out=o, shuffle=True)), ('F(4x4)', lambda: ops.conv3x3_c8w4(x8, p4, cout, residual=res, out=o, shuffle=True))):
        for _ in range(3): f()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(name, nm, round(sorted(ts)[2], 1), 'us', flush=True)

import os, sys
import torch
sys.path.insert(0, '/root/repo')
from adaptivepnp_sci_amd import ops
g = torch.Generator().manual_seed(0)
for (n,h,w) in ((8,64,64),(4,64,64)):
    pts=[]
    for cin in (8,24,48,96,192):
        cout=96
        x8 = ops.to_c8(torch.randn(n, cin, h, w, generator=g).cuda())
        wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        pw = ops.pack_conv3x3_wino(ops.pack_conv3x3(wt, None, Cin=cin, Cout=cout, device='cuda'), cin, cout)
        out = torch.empty(n, cout // 8, h, w, 8, device='cuda')
        f = lambda: ops.conv3x3_c8w(x8, pw, cout, relu=True, out=out)
        for _ in range(5): f()
        ts=[]
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1)/50*1e3)
        pts.append((cin//8, sorted(ts)[2]))
    print((n,h,w), 'units', n*(h//8)*(w//32)*3, ' '.join(f'CG={c}: {t:.2f}us' for c,t in pts))

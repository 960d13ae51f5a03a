import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from adaptivepnp_sci_amd import ops
rng = np.random.default_rng(7); g = torch.Generator().manual_seed(7)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
worst = 0
for it in range(160):
    n = int(rng.integers(1, 4)); cin = 8 * int(rng.integers(1, 17)); cout = 8 * int(rng.integers(1, 17))
    h, w = int(rng.integers(1, 140)), int(rng.integers(1, 300))
    shuf = cout % 32 == 0 and rng.random() < 0.3
    x = torch.randn(n, cin, h, w, generator=g); wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g)
    pk = ops.pack_conv3x3(wt, b, Cin=cin, Cout=cout, device='cuda'); p4 = ops.pack_conv3x3_wino4(pk, cin, cout)
    xc = ops.to_c8(x.cuda())
    d = ops.conv3x3_c8(xc, pk, cout, relu=True, shuffle=shuf)
    f = ops.conv3x3_c8w4(xc, p4, cout, relu=True, shuffle=shuf)
    e = rel(f.cpu(), d.cpu()); worst = max(worst, e)
    assert e < 4e-6, (n, cin, cout, h, w, shuf, e)
print('160 random shapes ok; worst rel-L2 vs the direct kernel', worst)

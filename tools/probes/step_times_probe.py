import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
net, _ = bench.load_weights()
y, Phi, orig = synth.make_problem(512, 512, 8, seed=0)
tv = AdmmRun(y, Phi, 'tv', False)
for _ in range(40): tv.step(0)
warm = tv.result_mosaic()
dev = torch.device('cuda')
run = AdmmRun(torch.from_numpy(y).to(dev), torch.from_numpy(Phi).to(dev), 'ffdnet_color', True, x0_bayer=warm, X_orig=torch.from_numpy(orig).to(dev), model=net)
ts = []
for k in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run.step(bench.SIGMA); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('per-step ms (synchronised):', ' '.join(f'{t:.2f}' for t in ts))

#!/bin/bash
# round 5: the F(4x4) kernel's whole-line stores with the nt hint (build/variants/libscipnp_w4nt.so) against the product, on the
# network iterations (one python process per row: DDnet, FastDVDnet, the headline FFDNet pass, the 256x256x16 tile)
set -u
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-product w4nt}; do
  unset SCIPNP_LIB
  [ $v != product ] && export SCIPNP_LIB=$GRAFT_REPO_ROOT/build/variants/libscipnp_$v.so
  echo "== $v"
  DD_STEPS=6 timeout -k 10 200 python tools/ddnet_bench.py 2>&1 | grep "ms/iteration"
  FD_STEPS=6 timeout -k 10 200 python tools/fastdvd_bench.py 2>&1 | grep "ms/iteration"
  timeout -k 10 300 python - <<'PY'
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
from adaptivepnp_sci_amd.nets import FFDNet
g = np.load('tests/golden/ffdnet_color_weights.npz')
net = FFDNet(); net.load_state_dict({k: torch.from_numpy(g[k]) for k in g.files})
for (H, W, B) in ((512, 512, 8), (256, 256, 16)):
    y, Phi, orig = synth.make_problem(H, W, B, 0)
    run = AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=net)
    for _ in range(10):
        run.step(25 / 255)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(25):
            run.step(25 / 255)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 25 * 1e3)
    print(f'FFDNet iteration {H}x{W}x{B}: {best:.3f} ms')
PY
done

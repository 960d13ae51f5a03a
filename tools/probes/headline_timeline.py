#!/usr/bin/env python3
"""GPU box, under `rocprofv3 --kernel-trace`: 12 iterations of the headline solve (two-stage ADMM + FFDNet, 512x512x8); with a trace
directory as argv[1]: the timeline of the LAST iteration -- every launch with its stream (queue), duration and the gap to the
previous launch of its queue, the union of busy time against the iteration's span."""
import csv, glob, os, sys
if len(sys.argv) > 1:
    f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('scipnp::', '')[:46], r.get('Queue_Id', '?'))
                  for r in csv.DictReader(open(f)))
    idx = [i for i, r in enumerate(rows) if 'pm_project_kernel' in r[2]]
    seg = rows[idx[-2]:idx[-1]]
    t0 = seg[0][0]
    last_end = {}
    for s, e, n, q in seg:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f'{(s - t0) / 1e3:9.1f} us  q{q:>3}  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {n}')
    ev = sorted([(s, 1) for s, e, n, q in seg] + [(e, -1) for s, e, n, q in seg])
    busy, depth, prev = 0, 0, ev[0][0]
    for t, d in ev:
        if depth > 0:
            busy += t - prev
        depth += d
        prev = t
    span = max(e for s, e, n, q in seg) - t0
    print(f'iteration span {span / 1e3:.1f} us, some kernel running {busy / 1e3:.1f} us, sum of kernel durations {sum(e - s for s, e, n, q in seg) / 1e3:.1f} us')
    sys.exit(0)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adaptivepnp_sci_amd import synth
from adaptivepnp_sci_amd.solver import AdmmRun
from adaptivepnp_sci_amd.nets import FFDNet
g = np.load(os.path.join(ROOT, 'tests/golden/ffdnet_color_weights.npz'))
net = FFDNet(); net.load_state_dict({k: torch.from_numpy(g[k]) for k in g.files})
y, Phi, orig = synth.make_problem(512, 512, 8, 0)
run = AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=net)
for _ in range(12):
    run.step(25 / 255)
torch.cuda.synchronize()
print('done')

#!/usr/bin/env python3
"""GPU box: where the time of an FFDNet online-finetune iteration goes -- per-phase HIP-event times of the trainer's
methods over several iterations (host enqueue vs device), allocator activity."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import synth, finetune
from adaptivepnp_sci_amd.solver import AdmmRun
from adaptivepnp_sci_amd.nets import FFDNet
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'ffdnet_color_weights.npz'))
net = FFDNet(); net.load_state_dict({k: torch.from_numpy(g[k]) for k in g.files})
y, Phi, orig = synth.make_problem(512, 512, 8, 0)
run = AdmmRun(y, Phi, 'ffdnet_color', True, X_orig=orig, model=net, update_=True, lr_=2e-6, update_per_iter=2,
              inital_iter=0, interval_iter=1)
marks = []


def wrap(cls, name):
    f = getattr(cls, name)

    def g_(self, *a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        t0 = time.perf_counter()
        r = f(self, *a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        marks.append((name, e0, e1, time.perf_counter() - t0))
        return r
    setattr(cls, name, g_)


for n_ in ('__init__', 'pack', 'forward_keep', 'loss_and_grad', 'backward', 'adam', 'write_back'):
    wrap(finetune._FFDNetTrainer, n_)
run.step(25 / 255); torch.cuda.synchronize()
import gc
gc.callbacks.append(lambda phase, info: phase == 'stop' and info['generation'] == 2 and print('  [gc] full collection'))
if os.environ.get('NO_GC'):
    gc.disable()
for rep in range(8):
    marks.clear()
    t0 = time.perf_counter(); run.step(25 / 255); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    ph = {}
    for name, e0, e1, host in marks:
        d = ph.setdefault(name, [0.0, 0.0]); d[0] += e0.elapsed_time(e1); d[1] += host * 1e3
    print(f'rep {rep}: enqueue {1e3 * (t1 - t0):6.1f} ms, total {1e3 * (t2 - t0):6.1f} ms; device mallocs {st["num_device_alloc"]}; ' +
          ' '.join(f'{k}={v[0]:.1f}/{v[1]:.1f}' for k, v in ph.items()) + '  (device ms / host ms)')

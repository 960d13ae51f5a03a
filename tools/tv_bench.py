#!/usr/bin/env python3
"""ON THE GPU BOX: the Chambolle TV step, tiled kernel (a launch per iteration) vs whole-plane kernel (one launch), and the
ADMM-TV iteration / whole call built on it (configs[0])."""
import io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adaptivepnp_sci_amd import ops, synth, admm_denoise_bayer_demosaic_pre
from adaptivepnp_sci_amd.solver import AdmmRun

dev = torch.device('cuda', 0)
for (C_, M, N) in ((32, 128, 128), (64, 128, 128), (32, 64, 64), (32, 256, 256)):
    x = torch.rand(C_, M, N, device=dev)
    b = torch.randn(C_, M, N, device=dev) * 0.1
    out = torch.empty_like(x)
    plan = ops.TvPlan(M, N, C_, 5, dev)
    for kernel in ((1, 2, 3, 4) if M <= 128 else (1, 3, 4)):
        for _ in range(5):
            ops.tv_chambolle(x, b, -1.0, out, plan, 0.1, kernel=kernel)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.tv_chambolle(x, b, -1.0, out, plan, 0.1, kernel=kernel)
        e1.record(); torch.cuda.synchronize()
        print(f'TV 5 iterations, {C_} planes of {M}x{N}, kernel={kernel}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us')
for (H, W, B) in ((256, 256, 8), (256, 256, 16), (128, 128, 8)):
    y, Phi, orig = synth.make_problem(H, W, B, 0)
    for two, defer in ((False, '1'), (False, '0'), (True, '1'), (True, '0')):
        os.environ['SCIPNP_TV_DEFER'] = defer
        run = AdmmRun(y, Phi, 'tv', two, X_orig=orig)
        for _ in range(3):
            run.step(0)
        # two timed loops per configuration, the slowest single step of each: round 3's profile carried a 743 us/iteration
        # line for the FIRST configuration of the process (host enqueue 736 us, i.e. one ~37 ms host-side stall inside its 50
        # steps); the second loop shows whether it belongs to the configuration or to the process's first loop
        for loop in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            worst = 0.0
            for _ in range(50):
                s0 = time.perf_counter()
                run.step(0)
                worst = max(worst, time.perf_counter() - s0)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            print(f'ADMM-TV {"two" if two else "one"}-stage {H}x{W}x{B} defer={defer} loop {loop}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} '
                  f'us/iteration (host enqueue {(t1 - t0) / 50 * 1e6:.1f} us, slowest single step {worst * 1e6:.0f} us)')
os.environ['SCIPNP_TV_DEFER'] = '1'
y0, Phi0, orig0 = synth.make_problem(256, 256, 8, 0)
for mode in ('1', '0'):
    os.environ['SCIPNP_HIPGRAPH'] = mode
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        admm_denoise_bayer_demosaic_pre(y0, Phi0, 1, 0.01, 'tv', [50], False, [0], X_orig=orig0, logf=io.StringIO())
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f'whole ADMM-TV call 256x256x8, 50 iterations, hipGraph={mode}: {min(ts):.2f} ms (runs: {" ".join(f"{t:.1f}" for t in ts)})')

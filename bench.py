#!/usr/bin/env python3
"""Headline benchmark: ADMM iterations/s (and reconstructed frames/s) of the two-stage PnP-ADMM +
FFDNet-colour solver on a 512x512x8 Bayer cube per GPU (BASELINE.json configs[1]).  The FFDNet convolutions run
on the error-compensated split-fp16 MFMA kernels by default (per-iterate parity <= 1e-5 verified by the GPU tests);
SCIPNP_CONV_PRECISION=f32 selects the fp32 MFMA kernels.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One step = one ADMM iteration over one cube (projection, mosaic+Malvar+w fusion, FFDNet on 8 frames,
theta/b/w updates, on-device PSNR partials) -- exactly `AdmmRun.step`, the code path behind
`twoStageAdmm_denoise_bayer`.  Inputs are resident in HBM when the timed region starts.  With N GPUs every
rank reconstructs its own cube (weak scaling, no collective inside the solve) and the (H,W,B)
mosaics are gathered to rank 0 with ONE RCCL gather at the end of the timed region.

The JSON line also carries
  roofline     : the dominant kernel (FFDNet body layer conv3x3), ALGORITHMIC FLOP/s measured with
                 HIP events around the body-layer launches inside the timed region;
  phi_step     : HBM roofline of the Phi / Phi^T Phi projection launch (events inside the timed region);
  cpu_baseline : the CPU oracle (bit-exact restatement of the reference) timed on this host on a
                 bounded sample of the same workload (rank 0, N=1 only); when the budget allows it runs the
                 very iterations the GPU ran and the line carries their parity (`cpu_baseline.parity`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 512
B = 8
SIGMA = 25 / 255
NB, NC = 12, 96
BODY_FLOP_PER_LAUNCH = 2.0 * 9 * NC * NC * (H // 2) * (W // 2) * B              # one body layer, 8 frames
FFDNET_FLOP_PER_ITER = 2.0 * 9 * (13 * NC + (NB - 2) * NC * NC + NC * 12) * (H // 2) * (W // 2) * B
PEAK_FP32_MFMA = 157.3e12                                                       # MI355X_MICROARCH.md
PEAK_F16_MFMA = 2500e12                                                         # dense f16/bf16 MFMA, MI355X_MICROARCH.md
# split-fp16 kernel: 14 MFMA 32x32x16 (32768 FLOP each) per (8 in-ch x 9 taps x 32x32 outputs) = 147456 algorithmic FLOP
SPLIT_EXEC_PER_ALGO = 14 * 32768 / 147456.0


def load_weights():
    from adaptivepnp_sci_amd.nets import FFDNet
    net = FFDNet()
    path = os.path.join(ROOT, 'tests', 'golden', 'ffdnet_color_weights.npz')
    if os.path.exists(path):
        g = np.load(path)
        net.load_state_dict({k: torch.from_numpy(g[k]) for k in g.files})
        return net, 'ffdnet_color.pth (reference weights, committed fixture)'
    torch.manual_seed(0)
    return net, 'random init'


def _usable_cpus():
    from adaptivepnp_sci_amd._lib import usable_cpus
    return usable_cpus()


def cpu_baseline(y, Phi, warm, orig, sd, budget_s=20.0, gpu_iters=None, gpu_mosaic=None, gpu_psnr=None):
    """The CPU oracle on the SAME cube and schedule, bounded to ~budget_s of CPU work.  Thread count:
    the fastest of a short calibration over {8,16,32,64} <= usable CPUs (PyTorch-CPU per-frame
    convolutions do not scale to hundreds of threads)."""
    from oracle import nets as ON
    from oracle import solver as OS
    onet = ON.OracleFFDNet()
    onet.load_state_dict(sd)
    onet.eval()
    usable = _usable_cpus()
    frame = torch.rand(1, 3, H, W)
    sig = torch.full((1, 1, 1, 1), SIGMA)
    best = (1e9, 1)
    with torch.no_grad():
        for n in [c for c in (8, 16, 32, 64) if c <= usable] or [usable]:
            torch.set_num_threads(n)
            onet(frame, sig)
            t0 = time.perf_counter()
            onet(frame, sig)
            best = min(best, (time.perf_counter() - t0, n))
        cores = best[1]
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        OS.two_stage_admm(y, Phi, 'ffdnet_color', [1], [SIGMA], x0_bayer=warm, X_orig=orig, model_denoise=onet)
        t1 = time.perf_counter() - t0
        iters = int(min(30, max(1, budget_s // max(t1, 1e-3))))
        # when the budget allows, run exactly as many iterations as the GPU did: the sample then doubles as a
        # full-size, free-running parity check of the timed run (north_star gates: 1e-5 rel-L2, 1e-4 dB)
        check = gpu_iters is not None and gpu_iters * t1 <= 2.5 * budget_s
        if check:
            iters = gpu_iters
        t0 = time.perf_counter()
        o = OS.two_stage_admm(y, Phi, 'ffdnet_color', [iters], [SIGMA], x0_bayer=warm, X_orig=orig, model_denoise=onet)
        dt = time.perf_counter() - t0
    try:
        cpu_model = next(l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name'))
    except Exception:
        cpu_model = 'unknown'
    out = dict(value=iters / dt, unit='ADMM iterations/s', cores=cores, kind='port', cpu_model=cpu_model,
               sample=f'{iters} two-stage ADMM+FFDNet iteration(s) of the same 512x512x8 cube (sigma 25/255, TV warm '
                      f'start), PyTorch-CPU oracle, {cores} of {usable} usable CPU threads, {dt:.1f} s')
    # one iteration on ONE thread (SURVEY 8d asks for both figures), if it fits the sample budget
    per_iter = dt / iters
    if per_iter * cores * 0.6 <= 15.0:
        torch.set_num_threads(1)
        with torch.no_grad():
            t0 = time.perf_counter()
            OS.two_stage_admm(y, Phi, 'ffdnet_color', [1], [SIGMA], x0_bayer=warm, X_orig=orig, model_denoise=onet)
        out['value_1_thread'] = 1.0 / (time.perf_counter() - t0)
        torch.set_num_threads(cores)
    if check:
        ref = o['x_bayer']
        out['parity'] = {'iterations': iters,
                         'rel_l2_final_iterate': float(np.linalg.norm(gpu_mosaic - ref) / np.linalg.norm(ref)),
                         'max_abs_psnr_diff_db': float(np.max(np.abs(np.array(gpu_psnr) - np.array(o['psnr_all'])))),
                         'gates': {'rel_l2': 1e-5, 'psnr_db': 1e-4}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=25)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--preheat', type=int, default=40, help='untimed denoiser passes before the warm-up steps (device clocks)')
    ap.add_argument('--cpu-budget', type=float, default=20.0, help='seconds of CPU-oracle work for cpu_baseline')
    args = ap.parse_args()

    # PyTorch sizes its intra-op pool by the visible cores (256 on the MI355X boxes) while the container's CPU quota is 16:
    # any CPU-side tensor op above the grain size wakes the pool, whose idle spinning exhausts the quota and gets this
    # (launching) thread throttled for tens of milliseconds -- keep the pool inside the quota
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    torch.set_num_threads(max(1, min(torch.get_num_threads(), _usable_cpus() // max(1, world))))   # ranks share the quota
    dist = None
    if world > 1 or os.environ.get('SCIPNP_BENCH_FORCE_DIST'):       # (the env var exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        local_rank %= max(1, torch.cuda.device_count())     # (a launcher may already have masked the devices per rank)
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())

    from adaptivepnp_sci_amd import synth
    from adaptivepnp_sci_amd.solver import AdmmRun
    net, wdesc = load_weights()
    y, Phi, orig = synth.make_problem(H, W, B, seed=rank)
    # TV warm start, as the reference driver does (two_stage_ADMM_Online_FFD_Warm.py:259-263); untimed
    tv = AdmmRun(y, Phi, 'tv', False)
    for _ in range(40):
        tv.step(0)
    warm = tv.result_mosaic()
    y_d, Phi_d, orig_d = (torch.from_numpy(a).to(dev) for a in (y, Phi, orig))
    run = AdmmRun(y_d, Phi_d, 'ffdnet_color', True, x0_bayer=warm, X_orig=orig_d, model=net)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # device pre-heat (untimed set-up, like the TV warm start above): after the light TV phase the part needs ~25 ms of
    # matrix load to reach its steady clocks (tools/probes/step_times_probe.py: 2.83, 2.89, 2.66, 2.56, 2.46, 2.43, 2.38, ...
    # 2.27 ms for the first twelve iterations); a production reconstruction lives in the steady state, a --steps 5 run would
    # measure the ramp.  The denoiser pass on a zeroed input, ~100 ms, solver state untouched.
    (run.eng.in_c8s if run.eng.precision == 'f16x3' else run.eng.in_c8).zero_()
    for _ in range(args.preheat):
        run.eng.forward()
    for _ in range(args.warmup):
        run.step(SIGMA)
    from adaptivepnp_sci_amd import shard
    if dist is not None:
        # untimed: the first gather sets up RCCL's point-to-point connections over xGMI
        shard.gather_units({rank: run.result_mosaic()}, world, (H, W, B), dev, dst=0)
        import ctypes
        ctypes.CDLL(None).fflush(None)          # every rank's RCCL banner (NCCL_DEBUG=VERSION) leaves its C stdout buffer now
    events = []
    run.profile_events = events
    phi_events = []
    run.phi_events = phi_events
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run.step(SIGMA)
    mosaic = run.result_mosaic()
    if dist is not None:                              # unit `rank` lives on this rank; ONE RCCL gather for the job
        gathered = shard.gather_units({rank: mosaic}, world, (H, W, B), dev, dst=0)
        assert rank != 0 or len(gathered) == world
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tpath):                           # measured in separate rocprofv3 --pmc passes, see the file's note
        traffic = json.load(open(tpath)).get(run.eng.precision, {}).get('hbm_bytes_per_launch')
    measured = {}
    mpath = os.path.join(ROOT, 'profiles', 'measured_peaks.json')
    if os.path.exists(mpath):                          # tools/peaks_bench.py on an MI355X of the pool: register-resident MFMA loop
        measured = json.load(open(mpath))              # on random operands, HBM read stream
    # the same projection kernel on a state large enough to leave the launch-latency regime (8 frames of 2048x2048, the
    # same B = 8 register path: 570 MB algorithmic per launch), 20 launches between one event pair
    phi_large = None
    if rank == 0:
        from adaptivepnp_sci_amd import ops
        Bl, Hl = 8, 2048
        th = torch.rand(Bl, 4, Hl // 2, Hl // 2, device=dev)
        bb, ph = torch.rand_like(th), (torch.rand_like(th) > 0.5).float()
        yy, ps = torch.rand(4, Hl // 2, Hl // 2, device=dev) * Bl / 2, torch.full((4, Hl // 2, Hl // 2), Bl / 2.0, device=dev)
        xo = torch.empty_like(th)
        for _ in range(3):
            ops.pm_project(th, bb, ph, yy, ps, 0, 1.0, 1.0, out=xo)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.pm_project(th, bb, ph, yy, ps, 0, 1.0, 1.0, out=xo)
        e1.record()
        torch.cuda.synchronize()
        lb = 16.0 * Hl * Hl * Bl + 8.0 * Hl * Hl
        ls = e0.elapsed_time(e1) / 20 * 1e-3
        phi_large = {'cube': [Hl, Hl, Bl], 'algorithmic_bytes_per_launch': lb, 'launch_us': ls * 1e6, 'achieved': lb / ls / 1e9,
                     'unit': 'GB/s', 'frac': lb / ls / 8e12}
        del th, bb, ph, yy, ps, xo
        # and on the bench's own 512 x 512 x 8 state, 50 launches between one event pair (no per-launch event overhead)
        xs = torch.empty_like(run.x)
        for _ in range(3):
            ops.pm_project(run.theta, run.b, run.Phi, run.y, run.Phisum, 0, 1.0, 1.0, out=xs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.pm_project(run.theta, run.b, run.Phi, run.y, run.Phisum, 0, 1.0, 1.0, out=xs)
        e1.record()
        torch.cuda.synchronize()
        phi_b2b_s = e0.elapsed_time(e1) / 50 * 1e-3
        del xs
    body_ms = [a.elapsed_time(b) for a, b in events]
    body_launch_s = float(np.mean(body_ms)) / 1e3 / (NB - 2)
    psnr = run.psnr_all()
    phi_s = float(np.median([a.elapsed_time(b) for a, b in phi_events])) / 1e3
    phi_bytes = 16.0 * H * W * B + 8.0 * H * W        # theta, b, Phi read + x written (4 E) + y, Phi_sum (2 HW)
    precision = run.eng.precision
    if rank == 0:
        iters_per_s = world * args.steps / dt
        achieved = BODY_FLOP_PER_LAUNCH / body_launch_s
        if precision == 'f16x3':
            peak, kname, dtype = PEAK_F16_MFMA, ('conv3x3_c8s_kernel<COB=3,TAG=0> (FFDNet body layer 96->96, 8 frames of '
                                                 '256x256; error-compensated split-fp16: 3 exact fp16 products per fp32 '
                                                 'product on v_mfma_f32_32x32x16_f16, fp32 accumulate)'), 'f16x3'
        else:
            peak, kname, dtype = PEAK_FP32_MFMA, ('conv3x3_c8_kernel<COB=3,TAG=0> (FFDNet body layer 96->96, 8 frames of '
                                                  '256x256, v_mfma_f32_32x32x2_f32)'), 'f32'
        peak_meas = measured.get('mfma_f16_32x32x16_2wave_per_simd_TFLOPs' if precision == 'f16x3'
                                 else 'mfma_f32_32x32x2_2wave_per_simd_TFLOPs')
        line = {
            'metric': 'admm_iters_per_s', 'value': iters_per_s, 'unit': 'ADMM iterations/s',
            'frames_per_s': iters_per_s * B,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype,
            'data': f'synthetic (seeded moving-sinusoid cube, Bernoulli(0.5) mask, noise-free y); weights: {wdesc}',
            'config': {'workload': 'two-stage ADMM + FFDNet-color, one 512x512x8 Bayer cube per GPU, Malvar demosaic, '
                                   'sigma=25/255, TV warm start, per-iteration PSNR on device', 'cube': [H, W, B],
                       'parallelism': f'{world} independent cube(s), one per GPU, one RCCL gather at the end'},
            'roofline': {'bound': 'mfma', 'achieved': achieved / 1e12, 'peak': peak / 1e12, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_unit': 'bytes/launch (PMC, profiles/pmc_traffic.json)',
                         'algorithmic_bytes_per_launch': 2.0 * B * NC * (H // 2) * (W // 2) * 4, 'kernel': kname,
                         'flop_per_launch': BODY_FLOP_PER_LAUNCH, 'avg_launch_ms': body_launch_s * 1e3,
                         'denoiser_flop_per_iter': FFDNET_FLOP_PER_ITER,
                         # honest bookkeeping for the split kernel: `achieved` counts ALGORITHMIC fp32-conv FLOPs; the
                         # matrix pipes execute 3.11x that in fp16 products
                         'mfma_flop_executed_over_algorithmic': SPLIT_EXEC_PER_ALGO if precision == 'f16x3' else 1.0,
                         'matrix_pipe_frac_of_peak': achieved * (SPLIT_EXEC_PER_ALGO if precision == 'f16x3' else 1.0) / peak,
                         'achieved_over_fp32_mfma_peak': achieved / PEAK_FP32_MFMA,
                         # the ceiling MEASURED with a register-resident MFMA loop on random operands (no memory traffic):
                         # what the part sustains under its own clock management, vs the 2.5 PFLOP/s vendor figure
                         'peak_measured': peak_meas,
                         'matrix_pipe_frac_of_measured_peak': (
                             achieved / 1e12 * (SPLIT_EXEC_PER_ALGO if precision == 'f16x3' else 1.0) / peak_meas) if peak_meas else None},
            # the Phi / Phi^T Phi projection step (north_star: HBM fraction), one launch per iteration, HIP events
            'phi_step': {'bound': 'hbm', 'kernel': 'pm_project_kernel<4,8,0> (p = theta - b/rho; x = p + Phi^T((y - Phi p)/(alpha rho + Phi_sum)))',
                         'algorithmic_bytes_per_launch': phi_bytes, 'launch_us': phi_s * 1e6,
                         'achieved': phi_bytes / phi_s / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': phi_bytes / phi_s / 8e12,
                         'peak_measured': measured.get('hbm_read_GBs'), 'large_state': phi_large,
                         'back_to_back': {'launches': 50, 'launch_us': phi_b2b_s * 1e6, 'achieved': phi_bytes / phi_b2b_s / 1e9,
                                          'frac': phi_bytes / phi_b2b_s / 8e12},
                         'note': 'event pair around one ~10 us launch includes ~2-3 us of event/launch overhead; rocprofv3 '
                                 'kernel time is in profiles/'},
            'preheat': f'{args.preheat} untimed denoiser passes before the warm-up steps (clock ramp after the TV phase)',
            'psnr_db_first_last': [psnr[args.warmup] if len(psnr) > args.warmup else None, psnr[-1] if psnr else None],
        }
        if world == 1:
            # SURVEY 8(d)'s other reading of the metric: whole solver calls with the reference driver's schedule
            # (sigma [25,12,6]/255 x [15,6,4] iterations), inputs as NumPy arrays, outputs read back to the host
            import io
            from adaptivepnp_sci_amd.solver import twoStageAdmm_denoise_bayer
            kw = dict(denoiser='ffdnet_color', iter_max=[15, 6, 4], sigma=[25 / 255, 12 / 255, 6 / 255], x0_bayer=warm,
                      X_orig=orig, model_denoise=net, logf=io.StringIO())
            twoStageAdmm_denoise_bayer(y, Phi, **kw)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); tr0 = time.perf_counter()
                twoStageAdmm_denoise_bayer(y, Phi, **kw)
                ts.append(time.perf_counter() - tr0)
            line['whole_reconstruction'] = {'schedule': 'two-stage ADMM + FFDNet-color, 25 iterations ([15,6,4] at sigma [25,12,6]/255), '
                                                        'H2D of y/Phi and D2H of the RGB cube + mosaic included, no finetune',
                                            'ms': 1e3 * min(ts), 'reconstructed_frames_per_s': B / min(ts)}
        if world == 1 and not args.no_cpu_baseline:
            sd = net.state_dict()
            line['cpu_baseline'] = cpu_baseline(y, Phi, warm.cpu().numpy(), orig, sd, args.cpu_budget,
                                                gpu_iters=args.warmup + args.steps, gpu_mosaic=mosaic.cpu().numpy(),
                                                gpu_psnr=psnr)
        else:
            line['cpu_baseline'] = None
        # RCCL (NCCL_DEBUG=VERSION on the boxes) writes its banner to the C-level stdout buffer: flush it first so that the
        # JSON line is the last thing this rank prints
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

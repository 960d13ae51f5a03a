#!/usr/bin/env python3
"""Headline benchmark: ADMM iterations/s (and reconstructed frames/s) of the two-stage PnP-ADMM +
FFDNet-colour solver on a 512x512x8 Bayer cube per GPU (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this process starts N rank processes itself (one per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in their environment) BEFORE it touches the GPU, waits for them and relays rank 0's
JSON line; under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` it is one of
the ranks.  `--gpus` must equal the number of ranks (non-zero exit otherwise).

One step = one ADMM iteration over one cube (projection, mosaic+Malvar+w fusion, FFDNet on 8 frames,
theta/b/w updates, on-device PSNR partials) -- exactly `AdmmRun.step`, the code path behind
`twoStageAdmm_denoise_bayer`.  Inputs are resident in HBM when the timed region starts.  With N GPUs every
rank reconstructs its own cube (weak scaling, no collective inside the solve) and the (H,W,B)
mosaics are gathered to rank 0 with ONE RCCL gather at the end of the timed region.

The SAME invocation times both convolution precisions on the same cube:
  headline (`value`, `dtype: "f32"`) : the FFDNet convolutions in fp32 arithmetic on the fp32 MFMA
                 (the reference's precision) as Winograd F(4x4,3x3) (csrc/conv_wino4.hip; F(2x2,3x3) with
                 SCIPNP_WINO_F4=0): every product an exact fp32 product, 4x fewer of them than the direct form;
                 `roofline.achieved` / `frac` count the products of the algorithm RUN (the matrix pipes' duty, <= 1),
                 the direct-form-equivalent rate rides beside it;
  `f32_direct_form` : the same pass with direct-form convolutions (executed = algorithmic FLOPs);
  `fast_path`  : the opt-in SCIPNP_CONV_PRECISION=f16x3 path, error-compensated split-fp16 operands on the fp16 MFMA (22 significant
                 bits per operand, fp32 accumulation; meets the 1e-5 / 1e-4 dB gates but is narrower than fp32,
                 so it is reported beside the headline, not as it).
The LAST stdout line is ONE compact JSON object (< 4 KB: `compact_line`; the driver parses it); everything below
that does not fit -- per-layer tables, the alternative forms' rooflines, run lists -- goes to the detail file the line
names (`detail`: gpurun_out/bench_detail_<mode>_n<N>.json; copies of judged runs are committed under profiles/).
The record carries
  roofline     : the dominant kernel (FFDNet body layer conv3x3), FLOP/s measured with HIP events around the
                 body-layer launches inside the timed region, `peak_measured` by the library's MFMA / HBM
                 micro-benchmarks in the same invocation; `traffic` from rocprofv3
                 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate passes) run as child processes of this
                 invocation when rocprofv3 is available, else from the committed profile it names;
  phi_step     : HBM roofline of the Phi / Phi^T Phi projection launch on an HBM-RESIDENT state (2048x2048x8, 570 MB per
                 launch): `frac` is an HBM fraction; the bench's own 36 MB state replays out of the Infinity Cache and is
                 reported apart (`cache_resident`, `cache_resident_frac`); `frac_rocprof` (here and in `roofline`) = the
                 same work over the kernel's rocprofv3 --kernel-trace average of a child pass of this run;
  configs      : the other single-GPU BASELINE configurations (ADMM-TV 256x256x8, FastDVDnet 512x512x8,
                 a 256x256x16 tile with the online finetune): ms/iteration, a per-layer-class table of the
                 convolution launches (kernel, launches, us, executed FLOPs, frac) and parity against the CPU
                 oracle for <= 3 iterations;
  cpu_baseline : the CPU oracle (bit-exact restatement of the reference) timed on this host on a
                 bounded sample of the same workload (rank 0, N=1 only); when the budget allows it runs the
                 very iterations the GPU ran and the line carries their parity (`cpu_baseline.parity`).
"""
import argparse
import contextlib
import gc
import io
import json
import os
import shutil
import subprocess
import sys
import tempfile
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
BODY_BYTES_PER_LAUNCH = 2.0 * B * NC * (H // 2) * (W // 2) * 4                  # activations read once + written once
FFDNET_FLOP_PER_ITER = 2.0 * 9 * (13 * NC + (NB - 2) * NC * NC + NC * 12) * (H // 2) * (W // 2) * B
PEAK_FP32_MFMA = 157.3e12                                                       # MI355X_MICROARCH.md
PEAK_F16_MFMA = 2500e12                                                         # dense f16/bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM = 8e12
# split-fp16 kernel: 14 MFMA 32x32x16 (32768 FLOP each) per (8 in-ch x 9 taps x 32x32 outputs) = 147456 algorithmic FLOP
SPLIT_EXEC_PER_ALGO = 14 * 32768 / 147456.0
PRECISIONS = ('f32', 'f16x3')


# ------------------------------------------------------------------------------------------------ rank launcher
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Start n rank processes of this script (rank i on GPU i).  Runs before this process has made any HIP call:
    the parent only waits and relays, it never initialises the GPU and never exec()s."""
    have = torch.cuda.device_count()                     # (counting devices does not initialise HIP on this image)
    if have < n and not os.environ.get('SCIPNP_BENCH_SHARE_GPU'):     # (test hook: ranks share GPUs, with SCIPNP_BENCH_BACKEND=gloo)
        print(f'bench.py: --gpus {n} but only {have} GPU(s) visible', file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SCIPNP_BENCH_SPAWNED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    lines = out0.splitlines()
    jl = [l for l in lines if l.startswith('{')]
    for l in lines:
        if not (jl and l is jl[-1]):
            print(l, file=sys.stderr)                    # RCCL banners etc.
    if any(rcs) or not jl:
        print(f'bench.py: rank exit codes {rcs}', file=sys.stderr)
        return max([abs(c) for c in rcs] + [1])
    print(jl[-1], flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ helpers
def rank_devices(dist, dev, coll_dev):
    """[[rank, HIP device index, PCI bus id], ...] of every rank (one small all_gather before the timed region): the line
    shows that N ranks sit on N different devices"""
    pr = torch.cuda.get_device_properties(dev)
    mine = [int(os.environ.get('RANK', 0)), int(dev.index), int(getattr(pr, 'pci_bus_id', -1))]
    if dist is None:
        return [mine]
    t = torch.tensor(mine, dtype=torch.int64, device=coll_dev)
    allt = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(allt, t)
    return [[int(v) for v in a.cpu()] for a in allt]


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


def conv_precision(p):
    """the convolution precision for everything constructed / stepped inside the block (adaptivepnp_sci_amd.config: no
    environment variable is touched)"""
    from adaptivepnp_sci_amd import config
    return config.use(precision=p)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


class Trace:
    """solver.ITERATE_HOOK target: keeps every reported iterate on the host"""

    def __init__(self):
        self.it = []

    def __call__(self, k, mosaic):
        self.it.append(mosaic.cpu().numpy())


def cpu_baseline(y, Phi, warm, orig, sd, budget_s=20.0, gpu_iters=None, gpu=None):
    """The CPU oracle on the SAME cube and schedule, bounded to ~budget_s of CPU work per run, two runs (the faster one is
    `value`; both are listed).  Threads = the container's CPU quota (what this process can really run), the intra-op
    pool pinned to it.  `gpu`: {precision: (mosaic, psnr list)} of the timed GPU runs for the full-size parity record."""
    from oracle import nets as ON
    from oracle import solver as OS
    onet = ON.OracleFFDNet()
    onet.load_state_dict(sd)
    onet.eval()
    usable = _usable_cpus()
    # thread count: the fastest of {quota/2, quota-2, quota} on one FFDNet frame (best of 3 each) -- a pool as large as
    # the cgroup quota is throttled as soon as anything else in the process runs, which halved round 1's driver-run figure
    frame, sig = torch.rand(1, 3, H, W), torch.full((1, 1, 1, 1), SIGMA)
    best = (1e9, 1)
    with torch.no_grad():
        for n in sorted({max(1, usable // 2), max(1, usable - 2), max(1, min(usable, 64))}):
            torch.set_num_threads(n)
            onet(frame, sig)
            for _ in range(3):
                t0 = time.perf_counter()
                onet(frame, sig)
                best = min(best, (time.perf_counter() - t0, n))
    cores = best[1]
    torch.set_num_threads(cores)
    with torch.no_grad():
        OS.two_stage_admm(y, Phi, 'ffdnet_color', [1], [SIGMA], x0_bayer=warm, X_orig=orig, model_denoise=onet)   # page in
        t0 = time.perf_counter()
        OS.two_stage_admm(y, Phi, 'ffdnet_color', [1], [SIGMA], x0_bayer=warm, X_orig=orig, model_denoise=onet)
        t1 = time.perf_counter() - t0
        iters = int(min(30, max(1, budget_s // max(t1, 1e-3))))
        # when the budget allows, run exactly as many iterations as the GPU did: the sample then doubles as a
        # full-size, free-running parity check of the timed run (north_star gates: 1e-5 rel-L2, 1e-4 dB)
        check = gpu_iters is not None and gpu_iters * t1 <= 2.5 * budget_s
        if check:
            iters = gpu_iters
        runs = []
        for _ in range(2):
            t0 = time.perf_counter()
            o = OS.two_stage_admm(y, Phi, 'ffdnet_color', [iters], [SIGMA], x0_bayer=warm, X_orig=orig, model_denoise=onet)
            runs.append(time.perf_counter() - t0)
            if runs[-1] > 1.5 * budget_s:
                break
    dt = min(runs)
    try:
        cpu_model = next(l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name'))
    except Exception:
        cpu_model = 'unknown'
    out = dict(value=iters / dt, unit='ADMM iterations/s', cores=cores, kind='port', cpu_model=cpu_model,
               runs_iters_per_s=[iters / r for r in runs],
               sample=f'{iters} two-stage ADMM+FFDNet iteration(s) of the same 512x512x8 cube (sigma 25/255, TV warm '
                      f'start), PyTorch-CPU oracle, {cores} threads (calibrated; CPU quota of this container {usable}), '
                      f'best of {len(runs)} runs ({", ".join(f"{r:.1f} s" for r in runs)})')
    # one iteration on ONE thread (SURVEY 8d asks for both figures), if it fits the sample budget
    if (dt / iters) * cores * 0.6 <= 15.0:
        torch.set_num_threads(1)
        with torch.no_grad():
            t0 = time.perf_counter()
            OS.two_stage_admm(y, Phi, 'ffdnet_color', [1], [SIGMA], x0_bayer=warm, X_orig=orig, model_denoise=onet)
        out['value_1_thread'] = 1.0 / (time.perf_counter() - t0)
        torch.set_num_threads(cores)
    if check and gpu:
        ref = o['x_bayer']
        out['parity'] = {'iterations': iters, 'gates': {'rel_l2': 1e-5, 'psnr_db': 1e-4}}
        for prec, (mosaic, psnr) in gpu.items():
            out['parity'][prec] = {'rel_l2_final_iterate': rel_l2(mosaic, ref),
                                   'max_abs_psnr_diff_db': float(np.max(np.abs(np.array(psnr) - np.array(o['psnr_all']))))}
    return out


# ------------------------------------------------------------------------------------------------ the output contract
LINE_LIMIT = 4096          # bytes of the final stdout line (round 3's 23 KB line was not parsed by the driver)


def _r(v, nd=4):
    """numbers rounded to `nd` SIGNIFICANT digits (keeps 7.3e-07 and 358.6 alike short), containers walked"""
    if isinstance(v, float):
        return float(f'{v:.{nd}g}') if v == v and abs(v) != float('inf') else None
    if isinstance(v, dict):
        return {k: _r(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, nd) for x in v]
    return v


def _pick_keys(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full, detail_path=None):
    """The ONE line the driver parses: the contract's keys, `roofline` and `cpu_baseline` with the fields the judge reads,
    one short record per extra form / configuration, and the path of the detail file.  Never longer than LINE_LIMIT:
    optional blocks are dropped (least important first) until it fits."""
    out = _pick_keys(full, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                            'dtype', 'data', 'ranks', 'world_size', 'backend', 'ranks_devices', 'collective',
                            'units_gathered_on_rank0', 'frame_iterations_per_s', 'frames_per_s', 'units_total', 'units_per_rank',
                            'units_batched_per_launch', 'per_rank_solve_s', 'per_rank_gather_s', 'timed_region_s',
                            'tile_iterations_per_s', 'finetune_events_per_tile', 'stitched_psnr_db', 'psnr_db_first_last'))
    out['vs_baseline'] = full.get('vs_baseline')
    cfg = full.get('config') or {}
    out['config'] = _pick_keys(cfg, ('workload', 'cube', 'cubes', 'tile', 'parallelism', 'conv_precision'))
    rf = full.get('roofline')
    if rf:
        out['roofline'] = _pick_keys(rf, ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch',
                                          'flop_per_launch', 'avg_launch_ms', 'rocprof_kernel_us', 'frac_rocprof', 'launch_shape',
                                          'direct_form_equivalent_TFLOPs', 'peak_measured'))
        out['roofline']['traffic'] = rf.get('traffic')
        out['roofline']['kernel'] = str(rf.get('kernel', '')).split(' (')[0]
    else:
        out['roofline'] = None
    ph = full.get('phi_step')
    if ph:
        out['phi_step'] = _pick_keys(ph, ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_rocprof', 'traffic', 'algorithmic_bytes_per_launch',
                                          'launch_us', 'rocprof_kernel_us', 'cube', 'cache_resident_frac'))
        out['phi_step']['timing'] = 'hipGraph replay'
        out['phi_step']['kernel'] = str(ph.get('kernel', '')).split(' (')[0]
        if ph.get('non_denoiser_chain'):
            out['phi_step']['chain'] = _pick_keys(ph['non_denoiser_chain'], ('algorithmic_bytes', 'us', 'frac'))
    cb = full.get('cpu_baseline')
    if cb:
        out['cpu_baseline'] = _pick_keys(cb, ('value', 'unit', 'cores', 'kind', 'cpu_model', 'value_1_thread'))
        out['cpu_baseline']['sample'] = str(cb.get('sample', ''))[:160]
        if cb.get('parity'):
            out['cpu_baseline']['parity'] = cb['parity']
    else:
        out['cpu_baseline'] = None
    opt = []                                                      # optional blocks, most important first
    for key, short in (('fast_path', 'f16x3'), ('f32_direct_form', 'f32_direct'), ('f32_winograd_f2x2', 'f32_f2x2')):
        b = full.get(key)
        if b:
            rec = _pick_keys(b, ('value', 'ms_per_step', 'frames_per_s'))
            rec['frac'] = (b.get('roofline') or {}).get('frac')
            rec['avg_launch_ms'] = (b.get('roofline') or {}).get('avg_launch_ms')
            opt.append(('forms', short, rec))
    wr = full.get('whole_reconstruction')
    if wr:
        opt.append(('whole_reconstruction_ms', None, {k: v['ms'] for k, v in wr.items() if isinstance(v, dict) and 'ms' in v}))
    cfgs = full.get('configs')
    if isinstance(cfgs, dict):
        if 'error' in cfgs:
            opt.append(('configs', 'error', cfgs['error'][:200]))
        for name, c in cfgs.items():
            if not isinstance(c, dict):
                continue
            rec = _pick_keys(c, ('ms_per_iteration', 'frac', 'units', 'ms_per_iteration_per_unit'))
            for prec in PRECISIONS:
                if isinstance(c.get(prec), dict):
                    rec[prec] = _pick_keys(c[prec], ('ms_per_iteration', 'frac', 'matrix_pipe_duty', 'ms_per_iteration_with_finetune_event'))
            par = c.get('parity')
            if isinstance(par, dict):
                vals = [v.get('max_rel_l2_per_iterate') for v in par.values() if isinstance(v, dict)] + [par.get('max_rel_l2_per_iterate')]
                vals = [v for v in vals if v is not None]
                rec['parity_max_rel_l2'] = max(vals) if vals else None
            opt.append(('configs', name, rec))
    mp = full.get('measured_peaks')
    if mp:
        opt.append(('measured_peaks', None, {k: v for k, v in mp.items() if isinstance(v, float)}))
    for group, name, rec in opt:
        if name is None:
            out[group] = rec
        else:
            out.setdefault(group, {})[name] = rec
    out['detail'] = detail_path
    out = _r(out)
    line = json.dumps(out, separators=(',', ':'))
    for group, name, _ in reversed(opt):                          # too long: drop optional blocks from the end
        if len(line) < LINE_LIMIT:                                # (strictly: emit() appends the newline)
            break
        if name is None:
            out.pop(group, None)
        else:
            out.get(group, {}).pop(name, None)
            if not out.get(group):
                out.pop(group, None)
        line = json.dumps(out, separators=(',', ':'))
    if len(line) >= LINE_LIMIT:                                   # (cannot happen with the fields above; never print a long line)
        out['data'] = str(out.get('data', ''))[:80]
        out['config'] = {'workload': str(out.get('config', {}).get('workload', ''))[:200]}
        line = json.dumps(out, separators=(',', ':'))
    assert len(line) < LINE_LIMIT, len(line)
    return line


_REAL_STDOUT = None


def claim_stdout():
    """From here on file descriptor 1 of this process IS stderr -- Python prints (the solvers mirror the reference's `loss:`
    / PSNR log lines) and C-level writes (RCCL's banner) alike -- and the original stdout is kept for the one JSON line."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)


def emit(full, tag):
    """write the full record to the detail file, print the compact line as the ONLY thing on the real stdout"""
    path = None
    try:
        d = os.environ.get('SCIPNP_BENCH_DETAIL_DIR', os.path.join(ROOT, 'gpurun_out'))
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f'bench_detail_{tag}.json')
        with open(path, 'w') as f:
            json.dump(full, f, indent=1)
        path = os.path.relpath(path, ROOT)
    except OSError as e:                                          # read-only checkout: the line still goes out
        print(f'bench.py: detail file not written ({e})', file=sys.stderr)
        path = None
    line = compact_line(full, path)
    import ctypes
    ctypes.CDLL(None).fflush(None)                                # C-level buffers (RCCL banner) leave before the line
    sys.stdout.flush()
    sys.stderr.flush()
    out = _REAL_STDOUT or sys.stdout
    out.write(line + '\n')
    out.flush()


# ------------------------------------------------------------------------------------------------ PMC traffic (child runs)
ROCPROF_KERNEL_US = None      # {kernel: median duration in us} of pmc_traffic()'s plain --kernel-trace child pass


def pmc_traffic(timeout_s=150):
    """HBM bytes per launch of the body-layer convolutions and the projection, measured by rocprofv3 PMC passes run as
    CHILD processes of this bench invocation on tools/pmc_probe.py (same kernels, same shapes, same cube): FETCH_SIZE and
    WRITE_SIZE in separate passes with --kernel-trace only, FETCH_SIZE doubled for gfx950 (MI355X_MICROARCH.md, HBM).
    Returns ({kernel tag: bytes}, source string) or (None, reason)."""
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None, 'rocprofv3 not on PATH'
    import csv
    import glob
    tmp = tempfile.mkdtemp(prefix='scipnp_pmc_', dir='/tmp')
    vals = {}
    kernel_us = None
    try:
        for name in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, name)
            r = subprocess.run([exe, '--kernel-trace', '--pmc', name, '--output-format', 'csv', '-d', d, '--',
                                sys.executable, os.path.join(ROOT, 'tools', 'pmc_probe.py')],
                               cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp', SCIPNP_PMC_PASS='1'), stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None, f'rocprofv3 --pmc {name} exited {r.returncode}: {r.stdout[-300:]}'
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row['Counter_Name'] == name:
                        vals.setdefault(row['Kernel_Name'], {}).setdefault(name, []).append(float(row['Counter_Value']))
        # third pass, no counters: the kernels' durations as rocprofv3 --kernel-trace reports them (what profiles/ holds), so that
        # the line's event / hipGraph timings can be read against the profiler's own clock in the SAME run
        d = os.path.join(tmp, 'trace')
        r = subprocess.run([exe, '--kernel-trace', '--output-format', 'csv', '-d', d, '--', sys.executable,
                            os.path.join(ROOT, 'tools', 'pmc_probe.py')], cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
        if r.returncode == 0:
            dur = {}
            for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
                for row in csv.DictReader(open(f)):
                    dur.setdefault(row['Kernel_Name'], []).append((float(row['End_Timestamp']) - float(row['Start_Timestamp'])) * 1e-3)
            # average of the kernel's launches of the measured block (pmc_probe.py: the LAST 60 of each kernel; everything before is preheat)
            kernel_us = {k.replace('scipnp::', '').replace('void ', ''): float(np.mean(v[-60:] if len(v) > 60 else v[len(v) // 2:])) for k, v in dur.items()}
    except Exception as e:                                  # noqa: BLE001 -- best effort, the bench line says what happened
        return None, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    global ROCPROF_KERNEL_US
    ROCPROF_KERNEL_US = kernel_us
    out = {}
    for k, dct in vals.items():
        if 'FETCH_SIZE' in dct and 'WRITE_SIZE' in dct:
            fetch = float(np.mean(dct['FETCH_SIZE'])) * 1024 * 2         # KiB; the counter sees half of a wide streaming read
            wr = float(np.mean(dct['WRITE_SIZE'])) * 1024
            out[k.replace('scipnp::', '').replace('void ', '')] = fetch + wr
    return out, 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child passes of this run, tools/pmc_probe.py)'


def _pick(traffic, needle):
    if not traffic:
        return None
    for k, v in traffic.items():
        if needle in k:
            return v
    return None


# ------------------------------------------------------------------------------------------------ measured ceilings
def measure_peaks(dev):
    """The ceilings of THIS device, measured in THIS invocation (after the timed passes, clocks warm) with the library's
    own micro-benchmarks (csrc/peaks.hip): a register-resident MFMA loop on pseudo-random operands at two waves per SIMD
    (no memory traffic at all) and an HBM read stream over 4 GiB.  Median of 3 event-timed launches each."""
    import ctypes as C
    from adaptivepnp_sci_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import diaglib                                    # libscipnp_diag.so: the micro-benchmarks are not part of the product library
    lib = diaglib.load()
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)      # noqa: E731
    p = lambda t: C.c_void_p(t.data_ptr())                                # noqa: E731

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        return sorted(ts)[len(ts) // 2]

    res = {'source': 'measured in this bench invocation (scipnp_bench_mfma / scipnp_bench_stream of libscipnp_diag.so, csrc/peaks.hip)'}
    blocks = 512                                                          # 2 waves per SIMD on 256 CUs
    out = torch.empty(blocks * 256, device=dev)
    for mode, name, per, iters in ((0, 'f16_32x32x16', 4 * 32768, 30000), (2, 'f32_32x32x2', 4 * 4096, 15000)):
        t = timed(lambda: _lib.check(lib.scipnp_bench_mfma(p(out), blocks, iters, mode, st()), 'scipnp_bench_mfma'))
        res[f'mfma_{name}_2wave_per_simd_TFLOPs'] = blocks * 4 * iters * per / t / 1e12
    n = 1 << 30
    a = torch.rand(n, device=dev)
    sblocks = 256 * 16
    sink = torch.empty(sblocks * 256, device=dev)
    t = timed(lambda: _lib.check(lib.scipnp_bench_stream(p(a), None, n, 0, sblocks, p(sink), st()), 'scipnp_bench_stream'))
    res['hbm_read_GBs'] = 4 * n / t / 1e9
    del a, sink, out
    res['device'] = torch.cuda.get_device_name(dev)
    return res


# ------------------------------------------------------------------------------------------------ per-layer-class tables
def executed_flop(family, n, cin, cout, h, w, flags):
    """(executed matrix FLOPs, output pixels per image) of one convolution launch, from the kernels' own tilings:
      conv3x3_c8_kernel  (direct fp32, v_mfma_f32_32x32x2_f32): 2*9*Cin*CoutP FLOP per output pixel, CoutP = Cout rounded to 32;
      conv3x3_c8w_kernel (fp32 Winograd F(2x2,3x3), v_mfma_f32_16x16x4_f32): 2*16*Cin*CoutP per 2x2 output tile (CoutP = 16
                         for layers of <= 16 output channels, which multiply one half of the 32-channel block);
      conv3x3_c8w4_kernel (fp32 Winograd F(4x4,3x3), same instruction): 2*36*Cin*CoutP per 4x4 output tile;
      conv3x3_c8s_kernel (split-fp16, v_mfma_f32_32x32x16_f16): 14 MFMAs of 32768 FLOP per (8 in-ch x 9 taps x 32 out-ch x
                         32 pixels) = 56 FLOP per (in-ch, out-ch, output pixel)."""
    stride2 = bool(flags & 4)
    ho, wo = ((h - 1) // 2 + 1, (w - 1) // 2 + 1) if stride2 else (h, w)
    coutp = (cout + 31) // 32 * 32
    if family == 'conv3x3_c8w_kernel':
        if cout <= 16:
            coutp = 16
        return 2.0 * 16 * cin * coutp * ((ho + 1) // 2) * ((wo + 1) // 2) * n, ho * wo
    if family == 'conv3x3_c8w4_kernel':
        return 2.0 * 36 * cin * coutp * ((ho + 3) // 4) * ((wo + 3) // 4) * n, ho * wo
    if family == 'conv3x3_c8s_kernel':
        return 56.0 * cin * coutp * ho * wo * n, ho * wo
    return 2.0 * 9 * cin * coutp * ho * wo * n, ho * wo


def layer_table(log, real_macs=None):
    """Aggregate an ops.LAUNCH_LOG (single-stream pass) by (kernel family, Cin, Cout, h, w, stride / shuffle): launches,
    average launch time, algorithmic and executed FLOPs per launch, and `frac` = min(algorithmic, executed) FLOP/s over the
    dense MFMA peak of the instruction the kernel issues -- never above the matrix pipes' own duty.  real_macs: {(Cin, Cout):
    multiply-adds per output pixel} for layers whose padded launch shape hides the real one (grouped / 3-channel layers)."""
    real_macs = real_macs or {}
    rows = {}
    for family, n, cin, cout, h, w, flags, e0, e1 in log:
        key = (family, cin, cout, h, w, flags & (4 | 8))
        r = rows.setdefault(key, {'us': [], 'n': n, 'flags': flags})
        r['us'].append(e0.elapsed_time(e1) * 1e3)
    out = []
    for (family, cin, cout, h, w, sf), r in rows.items():
        ex, px = executed_flop(family, r['n'], cin, cout, h, w, sf)
        alg = 2.0 * real_macs.get((cin, cout), 9 * cin * cout) * px * r['n']
        us = float(np.mean(r['us']))
        peak = PEAK_F16_MFMA if family == 'conv3x3_c8s_kernel' else PEAK_FP32_MFMA
        out.append({'kernel': family, 'frames': r['n'], 'cin': cin, 'cout': cout, 'h': h, 'w': w,
                    'form': ('stride2' if sf & 4 else 'pixelshuffle' if sf & 8 else 'stride1'),
                    'launches': len(r['us']), 'avg_us': us, 'total_us': float(np.sum(r['us'])),
                    'algorithmic_flop_per_launch': alg, 'executed_flop_per_launch': ex,
                    'algorithmic_TFLOPs': alg / us / 1e6, 'executed_TFLOPs': ex / us / 1e6, 'peak_TFLOPs': peak / 1e12,
                    'matrix_pipe_duty': ex / us / 1e6 / (peak / 1e12),
                    'frac': min(alg, ex) / us / 1e6 / (peak / 1e12)})
    out.sort(key=lambda r_: -r_['total_us'])
    return out


@contextlib.contextmanager
def single_stream_launch_log():
    """ops.LAUNCH_LOG installed and one stream (config.use(streams=1)): event pairs around overlapping launches would time each other"""
    from adaptivepnp_sci_amd import config, ops
    ops.LAUNCH_LOG = log = []
    try:
        with config.use(streams=1):
            yield log
    finally:
        ops.LAUNCH_LOG = None


# ------------------------------------------------------------------------------------------------ the timed runs
def time_precision(prec, args, ctx):
    """W warm-up steps, K timed steps of AdmmRun.step with the FFDNet convolutions in `prec`, bracketed by
    barrier + synchronize; MAX over ranks.  Returns the record and the run (state after W + K iterations)."""
    from adaptivepnp_sci_amd import shard
    from adaptivepnp_sci_amd.solver import AdmmRun
    dist, rank, world, dev, cdev = ctx['dist'], ctx['rank'], ctx['world'], ctx['dev'], ctx['coll_dev']
    direct, f2 = prec == 'f32_direct', prec == 'f32_f2'
    # the fp32 pass in direct form (csrc/conv.hip), or with its body layers as Winograd F(2x2,3x3) instead of F(4x4,3x3)
    from adaptivepnp_sci_amd import config
    cfg = config.current().replace(precision='f32' if (direct or f2) else prec, f32_form='direct' if direct else 'winograd',
                                   wino_f4=not f2)
    # (the run keeps `cfg` for its whole life -- construction and every step -- whatever the process default says)
    run = AdmmRun(ctx['y_d'], ctx['Phi_d'], 'ffdnet_color', True, x0_bayer=ctx['warm'], X_orig=ctx['orig_d'], model=ctx['net'],
                  config=cfg)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # device pre-heat (untimed set-up, like the TV warm start): after the light TV phase the part needs ~25 ms of
    # matrix load to reach its steady clocks (tools/probes/step_times_probe.py); a production reconstruction lives in the
    # steady state, a --steps 5 run would measure the ramp.  The denoiser pass on a zeroed input, solver state untouched.
    (run.eng.in_c8s if run.eng.precision == 'f16x3' else run.eng.in_c8).zero_()
    with config.solve_scope(run.config, config.FIELDS):
        for _ in range(args.preheat):
            run.eng.forward()
    for _ in range(args.warmup):
        run.step(SIGMA)
    if dist is not None:
        # untimed: the first gather sets up RCCL's point-to-point connections over xGMI
        shard.gather_units({rank: run.result_mosaic()}, world, (H, W, B), cdev, dst=0)
        import ctypes
        ctypes.CDLL(None).fflush(None)          # every rank's RCCL banner (NCCL_DEBUG=VERSION) leaves its C stdout buffer now
    events, phi_events = [], []
    # the split-fp16 pass runs as half-batches on side HIP streams (ops.on_side_streams): event pairs around its body
    # launches would time overlapping kernels, so that pass is event-timed after the timed region instead (below)
    from adaptivepnp_sci_amd import ops as _ops
    side = run.eng.precision == 'f16x3' and _ops.side_stream_count() > 1
    run.profile_events, run.phi_events = (None if side else events), phi_events
    # the host runs only a few launches ahead of the GPU: one long host pause inside the timed region (a generation-2 pass of the cyclic
    # garbage collector over everything torch / numpy have imported takes tens of ms) would drain the queue and idle the GPU for as long.
    # The collector is run NOW and held off until the region ends -- nothing the steps allocate is cyclic.
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run.step(SIGMA)
    mosaic = run.result_mosaic()
    n_gathered = 1
    if dist is not None:                              # unit `rank` lives on this rank; ONE RCCL gather for the job
        gathered = shard.gather_units({rank: mosaic}, world, (H, W, B), cdev, dst=0)
        assert rank != 0 or len(gathered) == world
        n_gathered = len(gathered) if rank == 0 else 0
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    run.profile_events = run.phi_events = None
    if side:                                          # the same pass on the last iterate's input, one stream, event pairs
        for _ in range(min(args.steps, 10)):
            run.eng.forward(events=events)
        torch.cuda.synchronize()
    run.check_overflow()                              # split-fp16 range guard (the solver entry points do this themselves)
    body_launch_s = float(np.mean([a.elapsed_time(b) for a, b in events])) / 1e3 / (NB - 2)
    phi_s = float(np.median([a.elapsed_time(b) for a, b in phi_events])) / 1e3
    psnr = run.psnr_all()
    rec = {'dtype': prec, 'value': world * args.steps / dt, 'unit': 'ADMM iterations/s', 'ms_per_step': 1e3 * dt / args.steps,
           'frame_iterations_per_s': world * args.steps / dt * B, 'body_launch_s': body_launch_s, 'phi_s': phi_s,
           'units_gathered': n_gathered, 'denoiser_streams': _ops.side_stream_count() if side else 1,
           'body_launch_timed': 'after the timed region, single stream' if side else 'inside the timed region',
           'psnr_db_first_last': [psnr[args.warmup] if len(psnr) > args.warmup else None, psnr[-1] if psnr else None]}
    return rec, run, mosaic, psnr


def roofline_form(f32_form):
    """the fp32 form the FFDNet BODY layers run in: 'winograd' splits into F(4x4,3x3) (default) and F(2x2,3x3) (SCIPNP_WINO_F4=0)"""
    from adaptivepnp_sci_amd import ops
    return 'winograd_f4' if (f32_form == 'winograd' and ops.wino_f4_enabled()) else f32_form


def roofline_record(prec, body_launch_s, traffic, traffic_src, measured, f32_form='winograd', shape=None):
    """Roofline of the dominant kernel.  `shape` = (H, W, B) of the unit the launch ran on (default: the 512x512x8 cube):
    FLOPs, algorithmic bytes and the kernel's name are those of THAT launch (the fixed-total modes run 256x256x16 tiles).  `achieved` / `frac` count the multiply-adds of the algorithm the kernel RUNS, per
    launch, over the event-timed launch duration -- for the Winograd kernel the 16 products per 2x2 tile and channel pair of
    F(2x2,3x3) (= 1/2.25 of the direct form's), so `frac` is the matrix pipes' own duty and can never exceed 1; the
    direct-form-equivalent rate is kept under `direct_form_equivalent_TFLOPs`; F(4x4,3x3) (the default fp32 form of the
    body layers, csrc/conv_wino4.hip) runs 36 products per 4x4 tile = 1/4 of the direct form's.  For the split-fp16 kernel, which EXECUTES
    3.11x the fp32 convolution's products as fp16 products, `frac` counts the fp32 convolution's FLOPs (the smaller figure)
    and the pipes' duty rides in `matrix_pipe_frac_of_peak`."""
    hs, ws, bs = shape or (H, W, B)
    body_flop = 2.0 * 9 * NC * NC * (hs // 2) * (ws // 2) * bs
    body_bytes = 2.0 * bs * NC * (hs // 2) * (ws // 2) * 4
    where = f'{bs} frames of {hs // 2}x{ws // 2}'
    direct_rate = body_flop / body_launch_s
    if prec == 'f16x3':
        peak, exec_ratio = PEAK_F16_MFMA, SPLIT_EXEC_PER_ALGO
        kname = ('conv3x3_c8s_kernel<COB=3,TAG=0> (FFDNet body layer 96->96, ' + where + '; error-compensated '
                 'split-fp16: 3 exact fp16 products per fp32 product on v_mfma_f32_32x32x16_f16, fp32 accumulate)')
        peak_meas = measured.get('mfma_f16_32x32x16_2wave_per_simd_TFLOPs')
        tr = _pick(traffic, 'conv3x3_c8s_kernel<3, 0')
    elif f32_form == 'winograd_f4':
        # F(4x4,3x3): 36 products per 4x4 output tile and channel pair instead of 144 -- the matrix pipes execute 1/4 of the
        # direct form's multiply-adds, every one an exact fp32 product on v_mfma_f32_16x16x4_f32
        peak, exec_ratio = PEAK_FP32_MFMA, 1.0 / 4.0
        kname = ('conv3x3_c8w4_kernel<TAG=0> (FFDNet body layer 96->96, ' + where + '; fp32 Winograd F(4x4,3x3), '
                 'v_mfma_f32_16x16x4_f32, input/output transforms fused into the kernel, tiles and weight slabs by LDS-DMA)')
        peak_meas = measured.get('mfma_f32_32x32x2_2wave_per_simd_TFLOPs')
        tr = _pick(traffic, 'conv3x3_c8w4_kernel<0, 0')
    elif f32_form == 'winograd':
        # F(2x2,3x3): 16 products per 2x2 output tile and channel pair instead of 36 -- the matrix pipes execute 1/2.25 of
        # the direct form's multiply-adds, every one an exact fp32 product on v_mfma_f32_16x16x4_f32
        peak, exec_ratio = PEAK_FP32_MFMA, 1.0 / 2.25
        kname = ('conv3x3_c8w_kernel<TAG=0,NW=4> (FFDNet body layer 96->96, ' + where + '; fp32 Winograd F(2x2,3x3), '
                 'v_mfma_f32_16x16x4_f32, input/output transforms fused into the kernel)')
        peak_meas = measured.get('mfma_f32_32x32x2_2wave_per_simd_TFLOPs')
        tr = _pick(traffic, 'conv3x3_c8w_kernel<0, 4')
    else:
        peak, exec_ratio = PEAK_FP32_MFMA, 1.0
        kname = 'conv3x3_c8_kernel<COB=3,TAG=0> (FFDNet body layer 96->96, ' + where + ', v_mfma_f32_32x32x2_f32)'
        peak_meas = measured.get('mfma_f32_32x32x2_2wave_per_simd_TFLOPs')
        tr = _pick(traffic, 'conv3x3_c8_kernel<3, 0')
    src = traffic_src
    if shape is not None:                            # fixed-total modes: no PMC pass runs for them, no borrowed figure either
        tr, src = None, 'not collected in this mode'
    elif tr is None:                                 # no live PMC pass: the committed profile, named
        tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tpath):
            key = ('f32_direct' if (prec == 'f32' and f32_form == 'direct') else
                   'f32_winograd_f4' if (prec == 'f32' and f32_form == 'winograd_f4') else prec)
            tr = json.load(open(tpath)).get(key, {}).get('hbm_bytes_per_launch')
            src = f'committed profile profiles/pmc_traffic.json (live PMC pass unavailable: {traffic_src})'
    flop = body_flop * min(1.0, exec_ratio)                  # the algorithm run: never more than the pipes execute
    achieved = flop / body_launch_s
    # the same kernel's average duration under rocprofv3 --kernel-trace in a child pass of THIS run (tools/pmc_probe.py: the clocks
    # preheated with 30+ ms of body launches, then 60 back-to-back launches) -- what profiles/ holds; `frac_rocprof` = executed FLOPs over it
    rp_us = (_pick(ROCPROF_KERNEL_US, kname.split('<')[0] + '<' + {'conv3x3_c8w4_kernel': '0, 0', 'conv3x3_c8w_kernel': '0, 4', 'conv3x3_c8_kernel': '3, 0', 'conv3x3_c8s_kernel': '3, 0'}.get(kname.split('<')[0], '')) if shape is None else None)
    return {'bound': 'mfma', 'achieved': achieved / 1e12, 'peak': peak / 1e12, 'unit': 'TFLOP/s', 'frac': achieved / peak,
            'frac_rocprof': (flop / (rp_us * 1e-6) / peak) if rp_us else None,
            'traffic': tr, 'traffic_unit': 'HBM bytes per launch', 'traffic_source': src,
            'rocprof_kernel_us': rp_us,
            'algorithmic_bytes_per_launch': body_bytes, 'kernel': kname, 'launch_shape': [hs, ws, bs],
            'flop_per_launch': flop, 'avg_launch_ms': body_launch_s * 1e3,
            'direct_form_flop_per_launch': body_flop, 'direct_form_equivalent_TFLOPs': direct_rate / 1e12,
            'denoiser_direct_form_flop_per_iter': FFDNET_FLOP_PER_ITER * (hs * ws * bs) / (H * W * B),
            # executed matrix FLOPs over the direct form's: 1/4 (Winograd F(4x4)), 1/2.25 (F(2x2)), 1 (direct), 3.11 (split-fp16)
            'mfma_flop_executed_over_direct_form': exec_ratio,
            'matrix_pipe_frac_of_peak': direct_rate * exec_ratio / peak,
            # the ceiling MEASURED in this invocation with a register-resident MFMA loop on random operands (no memory traffic)
            'peak_measured': peak_meas, 'peak_measured_source': measured.get('source'),
            'matrix_pipe_frac_of_measured_peak': (direct_rate / 1e12 * exec_ratio / peak_meas) if peak_meas else None}


def graph_timed(fn, n, reps=3):
    """seconds per call of `fn` (a fixed launch sequence on the current stream) with n calls captured into ONE hipGraph and the
    replay timed between two events: the GPU runs the n launches back to back whatever the host's launch rate is (the round-4
    driver box issued a 7 us launch every 15 us, so a host loop measured the host).  Best of `reps` replays; the host-loop time
    of the same n calls rides along for comparison."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n * 1e-3
        best = t if best is None else min(best, t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    del g
    return best, e0.elapsed_time(e1) / n * 1e-3


def phi_record(run, phi_s, traffic, traffic_src, measured, dev):
    from adaptivepnp_sci_amd import ops
    phi_bytes = 16.0 * H * W * B + 8.0 * H * W        # theta, b, Phi read + x written (4 E) + y, Phi_sum (2 HW)
    # the same projection kernel on a state large enough to leave the launch-latency regime (8 frames of 2048x2048, the
    # same B = 8 register path: 570 MB algorithmic per launch), 20 launches between one event pair
    Bl, Hl = 8, 2048
    th = torch.rand(Bl, 4, Hl // 2, Hl // 2, device=dev)
    bb, ph = torch.rand_like(th), (torch.rand_like(th) > 0.5).float()
    yy, ps = torch.rand(4, Hl // 2, Hl // 2, device=dev) * Bl / 2, torch.full((4, Hl // 2, Hl // 2), Bl / 2.0, device=dev)
    xo = torch.empty_like(th)
    ls, ls_host = graph_timed(lambda: ops.pm_project(th, bb, ph, yy, ps, 0, 1.0, 1.0, out=xo), 20)
    lb = 16.0 * Hl * Hl * Bl + 8.0 * Hl * Hl
    del th, bb, ph, yy, ps, xo
    # and on the bench's own 512 x 512 x 8 state: 50 launches captured into one hipGraph, the replay between one event pair
    xs = torch.empty_like(run.x)
    b2b, b2b_host = graph_timed(lambda: ops.pm_project(run.theta, run.b, run.Phi, run.y, run.Phisum, 0, 1.0, 1.0, out=xs), 50)
    # the whole non-denoiser chain of a two-stage iteration (116 E + 8 HW bytes = 245.4 MB at 512x512x8):
    # projection, mosaic + Malvar + w fusion + FFDNet input, theta / b / w updates + PSNR partials -- three launches, run
    # back to back on copies of the bench's state (the engine's output buffer as the "denoised" frames)
    eng = run.eng
    st = {k: getattr(run, k).clone() for k in ('theta', 'b', 'x', 'w', 'mosaic')}
    c8 = eng.in_c8 if eng.precision != 'f16x3' else None
    c8s = eng.in_c8s if eng.precision == 'f16x3' else None
    part = torch.empty(ops.post_nblocks(run.M, run.N, run.B), dtype=torch.float64, device=dev)

    def chain():
        ops.pm_project(st['theta'], st['b'], run.Phi, run.y, run.Phisum, 0, 1.0, 1.0, out=st['x'])
        # (as AdmmRun.step does since round 6: the mosaic, not x_rgb, travels from the pre to the post kernel)
        ops.pm_pre_denoise(st['x'], st['b'], st['w'], None, None, c8, 1.0, 0.01, SIGMA, net_in_c8s=c8s, mosaic=st['mosaic'])
        ops.pm_post_denoise(None, eng.out_c8, None, st['x'], None, st['theta'], st['b'], st['w'], False, run.orig, part, mosaic=st['mosaic'])
    chain_s, chain_host = graph_timed(chain, 30)
    chain_bytes = 116.0 * H * W * B + 8.0 * H * W              # SURVEY 8(d)'s figure (counts the x_rgb round trip: 24 E of it; the kernels now move 8 E there)
    chain_rec = {'launches': 3, 'kernels': 'pm_project_kernel, pm_pre_denoise_kernel, pm_post_denoise_kernel',
                 'algorithmic_bytes': chain_bytes, 'us': chain_s * 1e6, 'host_loop_us': chain_host * 1e6, 'timing': 'hipGraph replay of 30 chains', 'achieved': chain_bytes / chain_s / 1e9, 'unit': 'GB/s',
                 'frac': chain_bytes / chain_s / PEAK_HBM,
                 'frac_of_measured_hbm_read_peak': (chain_bytes / chain_s / 1e9 / measured['hbm_read_GBs']) if measured.get('hbm_read_GBs') else None}
    del st, part
    # (the 512x512x8 state is 65536 four-pixel chunks: below the kernel's 4-pixels-per-thread threshold, so it runs one pixel per
    # thread, <VEC=1,MAXB=8,MODE=0>; the 2048x2048x8 state of `large_state` runs <4,8,0> -- the names rocprofv3 shows)
    # `frac`, `achieved`, `launch_us`, `algorithmic_bytes_per_launch`: the kernel on an HBM-RESIDENT state (2048x2048x8, 570 MB per launch:
    # twice the Infinity Cache) -- the figure north_star names; the bench's own 512x512x8 state (36 MB) stays in the Infinity Cache
    # between launches, its replay rate is cache bandwidth and is reported apart, as `cache_resident`
    cache_res = {'cube': [H, W, B], 'kernel': 'pm_project_kernel<1,8,0>', 'algorithmic_bytes_per_launch': phi_bytes, 'launch_us': b2b * 1e6,
                 'host_loop_us': b2b_host * 1e6, 'timing': 'hipGraph replay of 50 launches', 'achieved': phi_bytes / b2b / 1e9, 'unit': 'GB/s',
                 'frac_of_hbm_peak': phi_bytes / b2b / PEAK_HBM,
                 'note': 'the 36 MB state of a 512x512x8 cube stays in the 256 MB Infinity Cache between launches: cache bandwidth, not an HBM fraction',
                 'rocprof_kernel_us': _pick(ROCPROF_KERNEL_US, 'pm_project_kernel<1'),
                 'in_step': {'launch_us': phi_s * 1e6, 'achieved': phi_bytes / phi_s / 1e9,
                             'note': 'event pair around one ~8 us launch: includes ~2-3 us of event / launch overhead'}}
    rp_large = _pick(ROCPROF_KERNEL_US, 'pm_project_kernel<4')
    return {'bound': 'hbm', 'kernel': 'pm_project_kernel<4,8,0> (p = theta - b/rho; x = p + Phi^T((y - Phi p)/(alpha rho + Phi_sum))) on a 2048x2048x8 state',
            'cube': [Hl, Hl, Bl], 'algorithmic_bytes_per_launch': lb, 'launch_us': ls * 1e6, 'host_loop_us': ls_host * 1e6,
            'timing': 'hipGraph replay of 20 launches', 'achieved': lb / ls / 1e9, 'peak': PEAK_HBM / 1e9, 'unit': 'GB/s', 'frac': lb / ls / PEAK_HBM,
            'rocprof_kernel_us': rp_large, 'frac_rocprof': (lb / (rp_large * 1e-6) / PEAK_HBM) if rp_large else None,
            'traffic': _pick(traffic, 'pm_project_kernel<4'), 'traffic_source': traffic_src,
            'peak_measured': measured.get('hbm_read_GBs'),
            'frac_of_measured_hbm_read_peak': (lb / ls / 1e9 / measured['hbm_read_GBs']) if measured.get('hbm_read_GBs') else None,
            'cache_resident': cache_res, 'cache_resident_frac': phi_bytes / b2b / PEAK_HBM,
            'non_denoiser_chain': chain_rec}


# ------------------------------------------------------------------------------------------------ the other BASELINE configs
def _ms_per_iter(run, sig, n, warm=2):
    for _ in range(warm):
        run.step(sig)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run.step(sig)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fastdvd_flop_per_iter(Hh, Ww, Bf):
    """algorithmic FLOPs of one FastDVDnet pass over Bf frames (packages/fastdvdnet/models.py:146-253): per frame one
    stage-2 DenBlock and, with the stage-1 outputs shared between neighbouring windows, one stage-1 DenBlock per frame
    (B + 2 - 2 duplicates of the circular window are de-duplicated by the engine: B stage-1 blocks for B frames)."""
    px = Hh * Ww

    def conv(ci, co, p, groups=1):
        return 2.0 * 9 * ci * co / groups * p
    blk = (conv(3 * 4, 3 * 30, px, 3) + conv(90, 32, px)                       # inc: grouped (3) + 90 -> 32
           + conv(32, 64, px / 4) + 2 * conv(64, 64, px / 4)                   # downc0
           + conv(64, 128, px / 16) + 2 * conv(128, 128, px / 16)              # downc1
           + 2 * conv(128, 128, px / 16) + conv(128, 256, px / 16)             # upc2 (+ PixelShuffle)
           + 2 * conv(64, 64, px / 4) + conv(64, 128, px / 4)                  # upc1 (+ PixelShuffle)
           + conv(32, 32, px) + conv(32, 3, px))                               # outc
    return 2 * Bf * blk


def config_records(ffd_sd, budget_s=60.0):
    """BASELINE configs other than the headline, one GPU: ms per iteration, the dominant kernel with its algorithmic
    fraction, and parity against the CPU oracle on the same seeded inputs for <= 3 iterations (free-running from the
    same warm start; gates 1e-5 rel-L2 per iterate).  Bounded: an oracle leg is shortened when it would not fit."""
    from adaptivepnp_sci_amd import solver as S
    from adaptivepnp_sci_amd import synth
    from adaptivepnp_sci_amd.nets import FFDNet
    from adaptivepnp_sci_amd.solver import AdmmRun
    from oracle import nets as ON
    from oracle import solver as OS
    t_start = time.perf_counter()
    out = {}

    def gpu_iterates(fn):
        tr = Trace()
        S.ITERATE_HOOK = tr
        try:
            with contextlib.redirect_stdout(io.StringIO()):          # the solvers print the reference's log lines
                fn()
        finally:
            S.ITERATE_HOOK = None
        return tr.it

    # ---- configs[0]: ADMM-TV warm start, 256x256x8 gray simulated cube, 50 iterations
    y, Phi, orig = synth.make_problem(256, 256, 8, seed=0)
    run = AdmmRun(y, Phi, 'tv', False, X_orig=orig)
    ms_host = _ms_per_iter(run, 0, 50, 5)
    ms, tv_timing = ms_host, 'host loop of 50 steps between two synchronisations'
    ms_graph = None
    try:                                                # like phi_step: 50 iterations (100 launches) captured into ONE hipGraph --
        # the rate the GPU sustains whatever the host's launch rate is.  Captured on a THROW-AWAY run (a captured step advances the
        # run's host-side state without executing: the run object is not usable afterwards)
        run_t = AdmmRun(y, Phi, 'tv', False, X_orig=orig)
        for _ in range(5):
            run_t.step(0)
        g_s, _ = graph_timed(lambda: run_t.step(0), 50)
        del run_t
        ms_graph = g_s * 1e3
        ms, tv_timing = ms_graph, 'hipGraph replay of 50 captured iterations, event-timed (host_loop_ms: the same steps issued by the host)'
    except Exception as e:                              # (capture refused: keep the host loop's figure)
        tv_timing += f'; hipGraph capture failed: {type(e).__name__}'
    E = 256 * 256 * 8
    tv_bytes = (16.0 * E + 8 * 256 * 256) + 8.0 * E + 20.0 * E          # projection + fused Chambolle (v in, out) + dual update
    its = gpu_iterates(lambda: S.admm_denoise_bayer_demosaic_pre(y, Phi, 1, 0.01, 'tv', [3], False, [0], X_orig=orig,
                                                                 logf=io.StringIO()))
    o = OS.one_stage_admm(y, Phi, 1, 0.01, 'tv', [3], [0], X_orig=orig)
    out['admm_tv_256'] = {
        'workload': 'configs[0]: ADMM-TV (one-stage, Chambolle 5 inner iterations), 256x256x8, per-iteration PSNR on device',
        # (keys since round 5: `ms_per_iteration` = the hipGraph-replay figure when a capture succeeded, else the host loop's; both are
        # always present under their own names)
        'dtype': 'f32', 'ms_per_iteration': ms, 'iterations_per_s': 1e3 / ms, 'host_loop_ms': ms_host, 'graph_replay_ms': ms_graph,
        'timing': tv_timing,
        'dominant_kernel': 'tv_band_kernel (all 5 Chambolle iterations of a 128x128 plane in one launch of 8 workgroups per plane: 16-row bands with a 4-row halo)',
        'bound': 'hbm (launch/VALU-latency limited at this size)', 'algorithmic_bytes_per_iteration': tv_bytes,
        'achieved_GBs': tv_bytes / (ms * 1e-3) / 1e9, 'frac': tv_bytes / (ms * 1e-3) / PEAK_HBM,
        'parity': {'iterations': 3, 'max_rel_l2_per_iterate': max(rel_l2(its[k], o['x_iterates'][k]) for k in range(3)), 'gate': 1e-5}}
    # the same solver on a UNIT BATCH of 8 such cubes (seeds 0..7): one launch sequence steps all of them (AdmmRun(units=8))
    U = 8
    pr = [synth.make_problem(256, 256, 8, seed=i) for i in range(U)]
    brun = AdmmRun([q[0] for q in pr], [q[1] for q in pr], 'tv', False, X_orig=[q[2] for q in pr], units=U)
    ms8_host = _ms_per_iter(brun, 0, 50, 5)
    ms8 = ms8_host
    try:                                    # (on its own run: a captured step advances the run's iteration count without executing)
        brun_t = AdmmRun([q[0] for q in pr], [q[1] for q in pr], 'tv', False, X_orig=[q[2] for q in pr], units=U)
        for _ in range(5):
            brun_t.step(0)
        g_s, _ = graph_timed(lambda: brun_t.step(0), 30)
        ms8 = g_s * 1e3
        del brun_t
    except Exception:
        pass
    single = AdmmRun(pr[U - 1][0], pr[U - 1][1], 'tv', False, X_orig=pr[U - 1][2])
    for _ in range(brun.k):
        single.step(0)
    out['admm_tv_256_x8'] = {
        'workload': 'configs[0] as a unit batch: 8 independent 256x256x8 cubes stepped by ONE launch sequence (2 launches per '
                    'iteration for all of them), per-iteration PSNR of every unit on device',
        'dtype': 'f32', 'units': U, 'ms_per_iteration': ms8, 'ms_per_iteration_per_unit': ms8 / U, 'host_loop_ms': ms8_host,
        'unit_iterations_per_s': U * 1e3 / ms8, 'speedup_per_unit_over_single_unit_run': ms / (ms8 / U),
        'algorithmic_bytes_per_iteration': U * tv_bytes, 'achieved_GBs': U * tv_bytes / (ms8 * 1e-3) / 1e9,
        'frac': U * tv_bytes / (ms8 * 1e-3) / PEAK_HBM,
        'parity': {'bit_identical_to_single_unit_run': bool(torch.equal(brun.result_mosaic()[U - 1], single.result_mosaic())),
                   'iterations': brun.k, 'max_rel_l2_per_iterate': rel_l2(brun.result_mosaic()[0].cpu().numpy(),
                                                                         OS.one_stage_admm(pr[0][0], pr[0][1], 1, 0.01, 'tv', [brun.k], [0],
                                                                                           X_orig=pr[0][2])['x_bayer'])
                   if brun.k <= 60 else None, 'gate': 1e-5}}

    # ---- configs[2]: two-stage ADMM + FastDVDnet, 512x512x8 (synthetic weights: model.pth is not in the reference snapshot)
    y, Phi, orig = synth.make_problem(512, 512, 8, seed=1)
    tv = AdmmRun(y, Phi, 'tv', False)
    for _ in range(10):
        tv.step(0)
    warm = tv.result_mosaic().cpu().numpy()
    fnet = torch.nn.DataParallel(synth.synth_fastdvdnet(1))
    flop = fastdvd_flop_per_iter(512, 512, 8)
    rec = {'workload': 'configs[2]: two-stage ADMM + FastDVDnet (5-frame window), 512x512x8, rho 0.55, sigma 8/255; synthetic '
                       'weights (model.pth absent from the reference snapshot)',
           'denoiser_flop_per_iteration': flop, 'dominant_kernel': 'conv3x3 c8/c8s kernels of the two DenBlock stages (16 blocks of 17 layers)'}
    gpu_it = {}
    # FastDVDnet launches whose padded shape hides the real one: the grouped first conv (3 groups of 4 -> 30 channels, run
    # as one 16 -> 96 launch) and the 32 -> 3 tail (run as 32 -> 8)
    real = {(16, 96): 9 * 4 * 90, (32, 8): 9 * 32 * 3}
    for prec in PRECISIONS:
        run = AdmmRun(y, Phi, 'fastdvd_color', True, x0_bayer=warm, X_orig=orig, model=fnet, conv_precision=prec)
        ms = _ms_per_iter(run, 8 / 255, 5, 2)
        # the same iteration on ONE stream with an event pair around every convolution launch: per-layer-class table
        with single_stream_launch_log() as log:
            for _ in range(3):
                run.step(8 / 255)
            torch.cuda.synchronize()
            table = layer_table(log, real)
        conv_us = sum(r['total_us'] for r in table) / 3
        ex = sum(r['executed_flop_per_launch'] * r['launches'] for r in table) / 3
        peak = PEAK_FP32_MFMA if prec == 'f32' else PEAK_F16_MFMA
        rec[prec] = {'ms_per_iteration': ms, 'iterations_per_s': 1e3 / ms, 'peak_TFLOPs': peak / 1e12,
                     'algorithmic_TFLOPs': flop / (ms * 1e-3) / 1e12,
                     'executed_matrix_flop_per_iteration': ex,
                     # whole-iteration fraction: min(algorithmic, executed) FLOPs of the denoiser over the iteration time
                     'frac': min(flop, ex) / (ms * 1e-3) / peak,
                     'matrix_pipe_duty': ex / (ms * 1e-3) / peak,
                     'conv_launch_us_per_iteration_single_stream': conv_us,
                     'layers': [{k: (round(v, 4) if isinstance(v, float) and v < 1e6 else v) for k, v in r.items()} for r in table],
                     'layers_note': 'one stream, HIP event pair per launch, 3 iterations; launches = per 3 iterations'}
        with conv_precision(prec):
            gpu_it[prec] = gpu_iterates(lambda: S.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'fastdvd_color', [2], False, [8 / 255],
                                                                             x0_bayer=warm, X_orig=orig, model_denoise=fnet,
                                                                             logf=io.StringIO()))
    onet = ON.cpu_data_parallel(ON.synth_fastdvdnet_weights(1))       # the same seeded tensors as synth.synth_fastdvdnet(1)
    t0 = time.perf_counter()
    o = OS.two_stage_admm(y, Phi, 'fastdvd_color', [1], [8 / 255], x0_bayer=warm, X_orig=orig, model_denoise=onet)
    t1 = time.perf_counter() - t0
    n_or = 1
    if t1 * 2 < 0.5 * budget_s:
        o = OS.two_stage_admm(y, Phi, 'fastdvd_color', [2], [8 / 255], x0_bayer=warm, X_orig=orig, model_denoise=onet)
        n_or = 2
    rec['parity'] = {'iterations': n_or, 'gate': 1e-5, 'oracle_s_per_iteration': t1}
    for prec in PRECISIONS:
        rec['parity'][prec] = {'max_rel_l2_per_iterate': max(rel_l2(gpu_it[prec][k], o['theta_iterates'][k]) for k in range(n_or))}
    out['fastdvd_512'] = rec

    # ---- the reference drivers' DEFAULT mode (deep_demosaicking=True, two_stage_ADMM_Online_FFD_Warm.py:28): FFDNet + DDnet per
    # iteration at 512x512x8 (dvp...:192-194; synthetic DDnet weights: the checkpoint is absent from the reference snapshot)
    y, Phi, orig = synth.make_problem(512, 512, 8, seed=2)
    tv = AdmmRun(y, Phi, 'tv', False)
    for _ in range(10):
        tv.step(0)
    warm = tv.result_mosaic().cpu().numpy()
    dnet = synth.synth_ddnet(0)
    rec = {'workload': "the reference drivers' default: two-stage ADMM + FFDNet-colour with DDnet deep demosaicking in place of Malvar, "
                       '512x512x8, sigma 25/255; synthetic DDnet weights (checkpoint absent from the reference snapshot)',
           'dominant_kernel': 'conv3x3 kernels of the 8 DDnet DenBlock evaluations per frame triplet (3 x temp1, 3 x temp11 at half '
                              'resolution + bilinear x2 + fusion, 2 x temp2) and the 12 FFDNet layers'}
    gpu_it = {}
    for prec in PRECISIONS:
        net = FFDNet()
        net.load_state_dict(ffd_sd)
        run = AdmmRun(y, Phi, 'ffdnet_color', True, x0_bayer=warm, X_orig=orig, model=net, model_demosaic=dnet, conv_precision=prec)
        ms = _ms_per_iter(run, SIGMA, 5, 2)
        with single_stream_launch_log() as log:
            for _ in range(2):
                run.step(SIGMA)
            torch.cuda.synchronize()
            table = layer_table(log)
        ex = sum(r['executed_flop_per_launch'] * r['launches'] for r in table) / 2
        alg = sum(r['algorithmic_flop_per_launch'] * r['launches'] for r in table) / 2
        peak = PEAK_FP32_MFMA if prec == 'f32' else PEAK_F16_MFMA
        rec[prec] = {'ms_per_iteration': ms, 'iterations_per_s': 1e3 / ms, 'peak_TFLOPs': peak / 1e12,
                     'padded_launch_flop_per_iteration': alg, 'executed_matrix_flop_per_iteration': ex,
                     'frac': min(alg, ex) / (ms * 1e-3) / peak, 'matrix_pipe_duty': ex / (ms * 1e-3) / peak,
                     'conv_launch_us_per_iteration_single_stream': sum(r['total_us'] for r in table) / 2,
                     'layers': [{k: (round(v, 4) if isinstance(v, float) and v < 1e6 else v) for k, v in r.items()} for r in table],
                     'layers_note': 'one stream, HIP event pair per launch, 2 iterations; launches = per 2 iterations; FLOPs of the '
                                    'PADDED launch shapes (DDnet widths 20 / 40 / 80 run in 24 / 40 / 80-channel c8 tensors)'}
        del run
        net2 = FFDNet()
        net2.load_state_dict(ffd_sd)
        with conv_precision(prec):
            gpu_it[prec] = gpu_iterates(lambda: S.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'ffdnet_color', [2], False, [SIGMA],
                                                                             x0_bayer=warm, X_orig=orig, model_denoise=net2,
                                                                             model_demosaic=dnet, logf=io.StringIO()))
    if time.perf_counter() - t_start < 2.0 * budget_s:
        onet = ON.OracleFFDNet()
        onet.load_state_dict(ffd_sd)
        onet.eval()
        t0 = time.perf_counter()
        with torch.no_grad():
            o = OS.two_stage_admm(y, Phi, 'ffdnet_color', [2], [SIGMA], x0_bayer=warm, X_orig=orig, model_denoise=onet,
                                  model_demosaic=ON.synth_ddnet_weights(0))
        rec['parity'] = {'iterations': 2, 'gate': 1e-5, 'oracle_s_per_iteration': (time.perf_counter() - t0) / 2}
        for prec in PRECISIONS:
            rec['parity'][prec] = {'max_rel_l2_per_iterate': max(rel_l2(gpu_it[prec][k], o['theta_iterates'][k]) for k in range(2))}
    out['ddnet_512'] = rec

    # ---- configs[4], one 256x256x16 tile with the online finetune (per-tile model copy), FFDNet
    y, Phi, orig = synth.make_problem(256, 256, 16, seed=3)
    tv = AdmmRun(y, Phi, 'tv', False)
    for _ in range(10):
        tv.step(0)
    warm = tv.result_mosaic().cpu().numpy()
    kw = dict(lr_=2e-6, inital_iter=1, interval_iter=2, update_=True, update_per_iter=1)
    tile_flop = FFDNET_FLOP_PER_ITER * (256 * 256 * 16) / (H * W * B)
    rec = {'workload': 'configs[4]: one 256x256 patch of the 1024x1024x16 colour cube (16 frames), two-stage ADMM + FFDNet, '
                       'online finetune (Adam on the measurement loss) firing at the gated iterations; 16 such tiles shard '
                       'over the ranks (shard.reconstruct_tiled)',
           'denoiser_flop_per_iteration': tile_flop,
           'dominant_kernel': 'FFDNet body conv3x3 (forward); conv3x3 weight-gradient + backward-data kernels in a finetune event'}
    gpu_it = {}
    for prec in PRECISIONS:
        net = FFDNet()
        net.load_state_dict(ffd_sd)
        run = AdmmRun(y, Phi, 'ffdnet_color', True, x0_bayer=warm, X_orig=orig, model=net, conv_precision=prec)
        ms = _ms_per_iter(run, SIGMA, 20, 3)
        peak = PEAK_FP32_MFMA if prec == 'f32' else PEAK_F16_MFMA
        # iteration WITH a finetune event (2 Adam steps, the reference driver's update_per_iter), steady state
        net2 = FFDNet()
        net2.load_state_dict(ffd_sd)
        run2 = AdmmRun(y, Phi, 'ffdnet_color', True, x0_bayer=warm, X_orig=orig, model=net2, update_=True, lr_=2e-6,
                       update_per_iter=2, inital_iter=0, interval_iter=1, conv_precision=prec)
        run2.step(SIGMA)
        run2.step(SIGMA)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run2.step(SIGMA)
        torch.cuda.synchronize()
        ev = (time.perf_counter() - t0) / 3 * 1e3
        with single_stream_launch_log() as log:
            for _ in range(3):
                run.step(SIGMA)
            torch.cuda.synchronize()
            table = layer_table(log, {(16, 96): 9 * 13 * 96, (96, 16): 9 * 96 * 12})
        ex = sum(r['executed_flop_per_launch'] * r['launches'] for r in table) / 3
        rec[prec] = {'ms_per_iteration': ms, 'iterations_per_s': 1e3 / ms, 'peak_TFLOPs': peak / 1e12,
                     'algorithmic_TFLOPs': tile_flop / (ms * 1e-3) / 1e12, 'executed_matrix_flop_per_iteration': ex,
                     'frac': min(tile_flop, ex) / (ms * 1e-3) / peak, 'matrix_pipe_duty': ex / (ms * 1e-3) / peak,
                     'layers': [{k: (round(v, 4) if isinstance(v, float) and v < 1e6 else v) for k, v in r.items()} for r in table],
                     'ms_per_iteration_with_finetune_event': ev}
        net3 = FFDNet()
        net3.load_state_dict(ffd_sd)
        with conv_precision(prec):
            gpu_it[prec] = gpu_iterates(lambda: S.twoStageAdmm_denoise_bayer(y, Phi, 1, 0.01, 'ffdnet_color', [3], False, [SIGMA],
                                                                             x0_bayer=warm, X_orig=orig, model_denoise=net3,
                                                                             logf=io.StringIO(), **kw))
    onet = ON.OracleFFDNet()
    onet.load_state_dict(ffd_sd)
    onet.eval()
    o = OS.two_stage_admm(y, Phi, 'ffdnet_color', [3], [SIGMA], x0_bayer=warm, X_orig=orig, model_denoise=onet,
                          lr=2e-6, inital_iter=1, interval_iter=2, update=True, update_per_iter=1)
    rec['parity'] = {'iterations': 3, 'gate': 1e-5, 'finetune_event_at_iteration': 2}
    for prec in PRECISIONS:
        rec['parity'][prec] = {'max_rel_l2_per_iterate': max(rel_l2(gpu_it[prec][k], o['theta_iterates'][k]) for k in range(3))}
    out['tile_256x256x16_finetune'] = rec
    out['seconds'] = time.perf_counter() - t_start
    return out


# ------------------------------------------------------------------------------------------------ fixed-total modes
DRIVER_SIGMA, DRIVER_ITERS = [25 / 255, 12 / 255, 6 / 255], [15, 6, 4]     # two_stage_ADMM_Online_FFD_Warm.py:71-76 (default scene)


def driver_sigma(k, steps):
    """sigma of iteration k when the reference driver's 25-iteration schedule ([15,6,4] at sigma [25,12,6]/255) is run for
    `steps` iterations: the stage boundaries scale with steps (exactly the driver's schedule at steps = 25)"""
    pos = (k + 0.5) * sum(DRIVER_ITERS) / steps
    acc = 0
    for sg, n in zip(DRIVER_SIGMA, DRIVER_ITERS):
        acc += n
        if pos < acc:
            return sg
    return DRIVER_SIGMA[-1]


def fixed_total_mode(args, ctx, net, wdesc):
    """BASELINE configs[3] (`--cubes C`: a fixed total of C independent 512x512x8 cubes, C/N per rank) and configs[4]
    (`--config tile1024`: one 1024x1024x16 colour cube as 16 patches of 256x256, per-tile model copy and online finetune,
    16/N tiles per rank, stitched on rank 0).  One step = one ADMM iteration of EVERY unit of the job; the units of a rank
    advance TOGETHER as one unit batch (solver.AdmmRun(units=...): one launch sequence for all of them while they share the
    denoiser weights -- the tiles until their online finetune fires, then split() into per-tile runs; --no-unit-batch: one
    after the other, as in round 3 and in the reference's loop); ONE gather (RCCL) ends the timed region; strong scaling.
    Rank 0 prints the JSON line with the per-rank solve and gather times."""
    import copy
    from adaptivepnp_sci_amd import shard, synth
    from adaptivepnp_sci_amd.solver import AdmmRun
    dist, rank, world, dev, cdev = ctx['dist'], ctx['rank'], ctx['world'], ctx['dev'], ctx['coll_dev']
    tiled = args.config == 'tile1024'
    if tiled:
        Hc, Wc, Bc, tile = 1024, 1024, 16, 256
        yc, Phic, origc = synth.make_problem(Hc, Wc, Bc, seed=5)             # the same cube on every rank
        units = shard.tile_cube(yc, Phic, tile, orig=origc)
        n_units, ushape = len(units), (tile, tile, Bc)
        fkw = dict(update_=True, lr_=2e-6, update_per_iter=2, inital_iter=1, interval_iter=15)   # driver: one event at k = 15
    else:
        n_units, ushape, fkw = args.cubes, (H, W, B), {}
        units = None
    events = []
    batch = not args.no_unit_batch

    def inputs(u):
        if tiled:
            y_u, Phi_u, _x0, orig_u = units[u]
            return y_u, Phi_u, orig_u
        return synth.make_problem(H, W, B, seed=u)

    def tv_warm(ins):
        """TV warm starts (40 iterations, as the reference driver; untimed) -- of all the rank's units in ONE unit batch"""
        if len(ins) > 1 and batch:
            tv = AdmmRun([i[0] for i in ins], [i[1] for i in ins], 'tv', False, units=len(ins))
            for _ in range(40):
                tv.step(0)
            return tv.result_mosaic()
        out = []
        for y_u, Phi_u, _o in ins:
            tv = AdmmRun(y_u, Phi_u, 'tv', False)
            for _ in range(40):
                tv.step(0)
            out.append(tv.result_mosaic())
        return out

    def prepare(u, throwaway=False):                                        # one unit, its own run (and --no-unit-batch)
        y_u, Phi_u, orig_u = inputs(u)
        kw = dict(fkw, interval_iter=2) if throwaway else fkw
        run = AdmmRun(y_u, Phi_u, 'ffdnet_color', True, x0_bayer=tv_warm([(y_u, Phi_u, orig_u)])[0], X_orig=orig_u,
                      model=copy.deepcopy(net) if tiled else net, conv_precision='f32', **kw)
        if not throwaway and not events and not getattr(prepare, 'claimed', False):
            prepare.claimed = True
            run.profile_events = events
        return run

    def prepare_all(mine_):
        """the rank's units as ONE unit batch (solver.AdmmRun(units=...)): they share the weights -- for the tiles until the
        online finetune fires (split() into per-tile runs with the model copies made here, untimed)"""
        ins = [inputs(u) for u in mine_]
        run = AdmmRun([i[0] for i in ins], [i[1] for i in ins], 'ffdnet_color', True, x0_bayer=tv_warm(ins),
                      X_orig=[i[2] for i in ins], model=net, conv_precision='f32', units=len(ins), **fkw)
        run.profile_events = events
        models = [copy.deepcopy(net) for _ in ins] if tiled else None
        if tiled:
            run.prepare_split(models)                # per-tile buffers and engines exist before the timed region (as the
        return {'batch': run, 'parts': None, 'mine': list(mine_), 'models': models}          # per-unit prepare() builds them)

    def gate(k):                                                            # solver.AdmmRun._cnn_step's finetune gate
        return bool(fkw) and k > fkw['inital_iter'] and k % fkw['interval_iter'] == 0

    def iterate_all(st, k):
        sig = driver_sigma(k, args.steps) if tiled else SIGMA
        if st['parts'] is None and gate(k):                                 # per-tile weights from here on
            st['parts'] = st['batch'].split(models=st['models'])
            st['batch'] = None
        if st['parts'] is None:
            st['batch'].step(sig)
        else:
            if st.get('lanes') is None:                                     # per-unit weights: the parts on host threads / streams
                from adaptivepnp_sci_amd.solver import PartLanes
                st['lanes'] = PartLanes(st['parts'], args.lanes)
            st['lanes'].step(sig)

    def finish_all(st):
        ms = [part.result_mosaic() for part in st['parts']] if st['parts'] is not None else st['batch'].result_mosaic()
        ms = ms if isinstance(ms, list) else [ms]
        return {u: (m if cdev == dev else m.cpu()) for u, m in zip(st['mine'], ms)}

    mine = shard.partition(n_units, world, rank)
    if mine:
        # device pre-heat and W warm-up steps on a throwaway run of the rank's first unit (the timed runs start at k = 0, so
        # the finetune gate fires where the driver's schedule puts it; the throwaway fires one at k = 2 to warm those kernels)
        warm_run = prepare(mine[0], throwaway=True)
        warm_run.eng.in_c8.zero_()
        for _ in range(args.preheat):
            warm_run.eng.forward()
        for k in range(args.warmup):
            warm_run.step(driver_sigma(k, max(args.warmup, 1)) if tiled else SIGMA)
        del warm_run
    if dist is not None:                                                     # untimed: RCCL sets up its connections
        shard.gather_units({u: torch.zeros(ushape, device=cdev) for u in mine}, n_units, ushape, cdev, dst=0)

    def iterate(run, k):
        run.step(driver_sigma(k, args.steps) if tiled else SIGMA)

    def finish(run):
        m = run.result_mosaic()
        return m if cdev == dev else m.cpu()

    if batch:
        got, timing = shard.timed_job(n_units, prepare_all, iterate_all, finish_all, ushape, cdev, args.steps,
                                      sync=torch.cuda.synchronize, batched=True)
    else:
        got, timing = shard.timed_job(n_units, prepare, iterate, finish, ushape, cdev, args.steps, sync=torch.cuda.synchronize)
    if rank != 0:
        return
    dt = timing['total_s']
    body_s = float(np.mean([a.elapsed_time(b) for a, b in events])) / 1e3 / (NB - 2) if events else None
    mosaic = shard.stitch_tiles(got, Hc, Wc, tile) if tiled else None
    psnr = None
    if tiled:
        mse = float(((mosaic.cpu().double() - torch.from_numpy(origc).double()) ** 2).mean())
        psnr = 10 * np.log10(1.0 / mse)
    flop_px = 2.0 * 9 * (13 * NC + (NB - 2) * NC * NC + NC * 12) / 4           # FFDNet FLOPs per full-resolution pixel and frame
    px = (Hc * Wc * Bc) if tiled else (H * W * B * n_units)
    line = {
        'metric': 'admm_iters_per_s', 'unit': 'ADMM iterations/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
        'data': f'synthetic (seeded moving-sinusoid cube(s), Bernoulli(0.5) mask, noise-free y); weights: {wdesc}',
        'ranks': world, **ctx['dist_info'], 'units_total': n_units, 'units_per_rank': timing['units'],
        'per_rank_solve_s': timing['solve_s'], 'per_rank_gather_s': timing['gather_s'], 'timed_region_s': dt,
        'collective': ('none (single process)' if dist is None else
                       'ONE RCCL gather (torch.distributed backend nccl)' if dist.get_backend() == 'nccl' else
                       f'ONE {dist.get_backend()} gather on host copies (SCIPNP_BENCH_BACKEND test hook)'),
        'denoiser_direct_form_TFLOPs': flop_px * px * args.steps / dt / 1e12,
        # the body launch of the unit shape this mode RUNS (256x256x16 tiles / 512x512x8 cubes), its own FLOPs and event time
        # frames of one body launch: the whole unit batch of the rank while the units share the weights
        'units_batched_per_launch': (timing['units'][0] if batch else 1),
        'roofline': None if body_s is None else roofline_record('f32', body_s, None, 'not collected in this mode', {},
                                                                 roofline_form(ctx['f32_form']),
                                                                 shape=(ushape[0], ushape[1], ushape[2] * (len(mine) if batch else 1))),
        'cpu_baseline': None, 'cpu_baseline_note': 'reported by the default mode (python bench.py), N = 1',
    }
    if tiled:
        line.update({
            'value': args.steps / dt, 'ms_per_step': 1e3 * dt / args.steps,
            'tile_iterations_per_s': n_units * args.steps / dt, 'frame_iterations_per_s': Bc * args.steps / dt,
            'config': {'workload': 'BASELINE configs[4]: 1024x1024x16 colour cube tiled into 16 patches of 256x256, two-stage ADMM '
                                   '+ FFDNet-color with the online finetune (per-tile model copy; lr 2e-6, 2 Adam steps per event, '
                                   'gate k > 1 and k % 15 == 0), sigma schedule of the reference driver scaled to --steps; one step = '
                                   'one ADMM iteration of the whole cube (all 16 tiles)',
                       'cube': [Hc, Wc, Bc], 'tile': tile, 'parallelism': f'16 tiles over {world} rank(s), one gather, stitched on rank 0'},
            'finetune_events_per_tile': sum(1 for k in range(args.steps) if k > 1 and k % 15 == 0),
            'lanes_after_split': (args.lanes if batch else 1),    # per-tile weights: the tiles' runs on this many host threads / streams
            'stitched_psnr_db': psnr})
    else:
        line.update({
            'value': n_units * args.steps / dt, 'ms_per_step': 1e3 * dt / args.steps,
            'frame_iterations_per_s': n_units * B * args.steps / dt,
            'config': {'workload': f'BASELINE configs[3]: batch of {n_units} independent 512x512x8 Bayer cubes, two-stage ADMM + '
                                   'FFDNet-color (Malvar, sigma 25/255, TV warm start, per-iteration PSNR on device); one step = one '
                                   f'ADMM iteration of every cube; value = cube-iterations/s of the whole job',
                       'cube': [H, W, B], 'cubes': n_units, 'parallelism': f'{n_units} cubes over {world} rank(s), one gather'}})
    emit(line, ('tile1024' if tiled else f'cubes{n_units}') + f'_n{world}')


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=25)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the records of the other BASELINE configurations')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 --pmc child passes (roofline.traffic)')
    ap.add_argument('--no-fast-path', action='store_true', help='time only the fp32 headline (profiling runs)')
    ap.add_argument('--preheat', type=int, default=40, help='untimed denoiser passes before the warm-up steps (device clocks)')
    ap.add_argument('--cpu-budget', type=float, default=20.0, help='seconds of CPU-oracle work per cpu_baseline run')
    ap.add_argument('--cubes', type=int, default=0, help='BASELINE configs[3]: a FIXED total of this many 512x512x8 cubes over '
                                                         'the ranks (strong scaling); 0 = one cube per rank (weak, the default)')
    ap.add_argument('--no-unit-batch', action='store_true', help='fixed-total modes: step the units of a rank one after the other '
                                                                  '(round 3 behaviour) instead of as one unit batch')
    ap.add_argument('--lanes', type=int, default=1, help='fixed-total modes after the first finetune event (per-unit weights): host '
                                                        'threads / HIP streams the per-unit runs are stepped on (solver.PartLanes; 2 lanes '
                                                        'measured 0.693 against 0.710 s on one box and 0.776 against 0.715 s on another: off by default)')
    ap.add_argument('--config', choices=['headline', 'tile1024'], default='headline',
                    help='tile1024 = BASELINE configs[4]: 1024x1024x16 cube as 16 tiles of 256x256 with the online finetune')
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.cubes < 0 or (args.cubes and args.config != 'headline'):
        ap.error('--cubes must be >= 0 and cannot be combined with --config tile1024')

    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))           # nothing above touched the GPU
    claim_stdout()                                               # nothing but the final JSON line reaches stdout
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(env_world or 1)
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU', file=sys.stderr)
        sys.exit(2)

    # the rocprofv3 --pmc child passes (roofline.traffic) run BEFORE this process touches the GPU: children are only ever
    # started from a parent that has not initialised HIP
    traffic, traffic_src = (None, 'skipped (--no-pmc or N > 1)')
    if world == 1 and rank == 0 and not args.no_pmc:
        traffic, traffic_src = pmc_traffic()

    # PyTorch sizes its intra-op pool by the visible cores (256 on the MI355X boxes) while the container's CPU quota is 16:
    # any CPU-side tensor op above the grain size wakes the pool, whose idle spinning exhausts the quota and gets this
    # (launching) thread throttled for tens of milliseconds -- keep the pool inside the quota (ranks share it)
    torch.set_num_threads(max(1, min(torch.get_num_threads(), _usable_cpus() // max(1, world))))
    dist = None
    if world > 1 or os.environ.get('SCIPNP_BENCH_FORCE_DIST'):       # (the env var exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        local_rank %= max(1, torch.cuda.device_count())     # (a launcher may already have masked the devices per rank)
        torch.cuda.set_device(local_rank)
        # SCIPNP_BENCH_BACKEND=gloo is a test hook: several ranks on ONE GPU (RCCL refuses two ranks per device), the
        # collectives then run on host copies -- exercises the launcher and the rank plumbing on a 1-GPU box
        backend = os.environ.get('SCIPNP_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())

    from adaptivepnp_sci_amd import synth
    from adaptivepnp_sci_amd.solver import AdmmRun
    net, wdesc = load_weights()
    coll_dev = dev if (dist is None or dist.get_backend() == 'nccl') else torch.device('cpu')
    dist_info = {'world_size': dist.get_world_size() if dist is not None else 1,
                 'backend': (dist.get_backend() if dist is not None else 'none') +
                            (' (RCCL)' if dist is not None and dist.get_backend() == 'nccl' else ''),
                 'ranks_devices': rank_devices(dist, dev, coll_dev)}
    if args.cubes or args.config == 'tile1024':
        from adaptivepnp_sci_amd.nets import f32_conv_form
        fixed_total_mode(args, dict(dist=dist, rank=rank, world=world, dev=dev, coll_dev=coll_dev, f32_form=f32_conv_form(),
                                    dist_info=dist_info), net, wdesc)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    y, Phi, orig = synth.make_problem(H, W, B, seed=rank)
    # TV warm start, as the reference driver does (two_stage_ADMM_Online_FFD_Warm.py:259-263); untimed
    tv = AdmmRun(y, Phi, 'tv', False)
    for _ in range(40):
        tv.step(0)
    warm = tv.result_mosaic()
    y_d, Phi_d, orig_d = (torch.from_numpy(a).to(dev) for a in (y, Phi, orig))
    coll_dev = dev if (dist is None or dist.get_backend() == 'nccl') else torch.device('cpu')
    ctx = dict(dist=dist, rank=rank, world=world, dev=dev, coll_dev=coll_dev, y_d=y_d, Phi_d=Phi_d, orig_d=orig_d, warm=warm, net=net)

    recs, gpu_out, last_run = {}, {}, None
    from adaptivepnp_sci_amd.nets import f32_conv_form
    f32_form = f32_conv_form()
    passes = list(PRECISIONS[:1] if args.no_fast_path else PRECISIONS)
    if f32_form == 'winograd' and not args.no_fast_path:
        passes.append('f32_direct')
        from adaptivepnp_sci_amd import ops as _o
        if _o.wino_f4_enabled():
            passes.append('f32_f2')
    for prec in passes:
        rec, run, mosaic, psnr = time_precision(prec, args, ctx)
        recs[prec] = rec
        gpu_out[prec] = (mosaic.cpu().numpy(), psnr)
        if prec == 'f32':
            last_run = run
        else:
            del run
    if rank == 0:
        try:
            measured = measure_peaks(dev)                  # this device, this invocation, clocks warm from the timed passes
        except Exception as e:                             # noqa: BLE001 -- the headline must still be printed
            measured = {'source': f'unavailable ({type(e).__name__}: {e})'}
        head = recs['f32']
        line = {
            'metric': 'admm_iters_per_s', 'value': head['value'], 'unit': 'ADMM iterations/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': head['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': f'synthetic (seeded moving-sinusoid cube, Bernoulli(0.5) mask, noise-free y); weights: {wdesc}',
            'config': {'workload': 'two-stage ADMM + FFDNet-color, one 512x512x8 Bayer cube per GPU, Malvar demosaic, '
                                   'sigma=25/255, TV warm start, per-iteration PSNR on device; convolutions in the library default '
                                   'precision (SCIPNP_CONV_PRECISION=f32, Winograd F(4x4,3x3) on the fp32 MFMA)', 'cube': [H, W, B],
                       'conv_precision': 'f32 (library default)',
                       'parallelism': f'{world} independent cube(s), one per GPU, one RCCL gather at the end'},
            'ranks': world, 'units_gathered_on_rank0': head['units_gathered'], **dist_info,
            'collective': ('none (single process)' if dist is None else
                           'RCCL gather (torch.distributed backend nccl)' if dist.get_backend() == 'nccl' else
                           f'{dist.get_backend()} gather on host copies (SCIPNP_BENCH_BACKEND test hook)'),
            # frame-iterations/s = ADMM iterations/s x 8 frames per cube; `frames_per_s` (reconstructed frames/s
            # of a whole solver call with the reference driver's 25-iteration schedule) is filled in below at N = 1
            'frame_iterations_per_s': head['frame_iterations_per_s'],
            'frames_per_s': None,
            'roofline': roofline_record('f32', head['body_launch_s'], traffic, traffic_src, measured, roofline_form(f32_form)),
            'phi_step': phi_record(last_run, head['phi_s'], traffic, traffic_src, measured, dev),
            'measured_peaks': measured,
            'preheat': f'{args.preheat} untimed denoiser passes before the warm-up steps (clock ramp after the TV phase)',
            'psnr_db_first_last': head['psnr_db_first_last'],
        }
        if 'f16x3' in recs:
            fp = recs['f16x3']
            line['fast_path'] = {
                'dtype': 'f16x3', 'note': 'opt-in path (SCIPNP_CONV_PRECISION=f16x3; the library default is f32 = the headline): every fp32 operand carried as two fp16 '
                                          'numbers (22 significant bits), 3 of the 4 partial products, fp32 accumulation',
                'value': fp['value'], 'unit': 'ADMM iterations/s', 'ms_per_step': fp['ms_per_step'],
                'frame_iterations_per_s': fp['frame_iterations_per_s'], 'speedup_over_f32': fp['value'] / head['value'],
                'denoiser_streams': fp['denoiser_streams'], 'body_launch_timed': fp['body_launch_timed'],
                'roofline': roofline_record('f16x3', fp['body_launch_s'], traffic, traffic_src, measured),
                'psnr_db_first_last': fp['psnr_db_first_last'],
                'parity_vs_f32_path': {'rel_l2_final_iterate': rel_l2(gpu_out['f16x3'][0], gpu_out['f32'][0]),
                                       'max_abs_psnr_diff_db': float(np.max(np.abs(np.array(gpu_out['f16x3'][1]) -
                                                                                   np.array(gpu_out['f32'][1]))))}}
        if 'f32_direct' in recs:
            fd = recs['f32_direct']
            line['f32_direct_form'] = {
                'dtype': 'f32', 'note': 'the same fp32 pass with the convolutions in direct form on v_mfma_f32_32x32x2_f32 '
                                        '(SCIPNP_F32_CONV=direct): executed = algorithmic FLOPs, the plain MFMA roofline',
                'value': fd['value'], 'unit': 'ADMM iterations/s', 'ms_per_step': fd['ms_per_step'],
                'roofline': roofline_record('f32', fd['body_launch_s'], traffic, traffic_src, measured, 'direct'),
                'parity_vs_headline': {'rel_l2_final_iterate': rel_l2(gpu_out['f32_direct'][0], gpu_out['f32'][0]),
                                       'max_abs_psnr_diff_db': float(np.max(np.abs(np.array(gpu_out['f32_direct'][1]) -
                                                                                   np.array(gpu_out['f32'][1]))))}}
        if 'f32_f2' in recs:
            f2 = recs['f32_f2']
            line['f32_winograd_f2x2'] = {
                'dtype': 'f32', 'note': 'the same fp32 pass with the body layers as Winograd F(2x2,3x3) (SCIPNP_WINO_F4=0; the '
                                        'headline form until round 3): 4 instead of 2.25 multiply-adds per output',
                'value': f2['value'], 'unit': 'ADMM iterations/s', 'ms_per_step': f2['ms_per_step'],
                'roofline': roofline_record('f32', f2['body_launch_s'], traffic, traffic_src, measured, 'winograd'),
                'parity_vs_headline': {'rel_l2_final_iterate': rel_l2(gpu_out['f32_f2'][0], gpu_out['f32'][0]),
                                       'max_abs_psnr_diff_db': float(np.max(np.abs(np.array(gpu_out['f32_f2'][1]) -
                                                                                   np.array(gpu_out['f32'][1]))))}}
        if world == 1:
            # SURVEY 8(d)'s other reading of the metric: whole solver calls with the reference driver's schedule
            # (sigma [25,12,6]/255 x [15,6,4] iterations), inputs as NumPy arrays, outputs read back to the host
            from adaptivepnp_sci_amd.solver import twoStageAdmm_denoise_bayer
            kw = dict(denoiser='ffdnet_color', iter_max=[15, 6, 4], sigma=[25 / 255, 12 / 255, 6 / 255], x0_bayer=warm,
                      X_orig=orig, model_denoise=net, logf=io.StringIO())
            whole = {'schedule': 'two-stage ADMM + FFDNet-color, 25 iterations ([15,6,4] at sigma [25,12,6]/255), '
                                 'H2D of y/Phi and D2H of the RGB cube + mosaic included, no finetune'}
            for prec in [p_ for p_ in recs if p_ in PRECISIONS]:
                with conv_precision(prec), contextlib.redirect_stdout(io.StringIO()):
                    twoStageAdmm_denoise_bayer(y, Phi, **kw)
                    ts = []
                    for _ in range(3):
                        torch.cuda.synchronize()
                        tr0 = time.perf_counter()
                        twoStageAdmm_denoise_bayer(y, Phi, **kw)
                        ts.append(time.perf_counter() - tr0)
                whole[prec] = {'ms': 1e3 * min(ts), 'reconstructed_frames_per_s': B / min(ts)}
            line['whole_reconstruction'] = whole
            line['frames_per_s'] = whole['f32']['reconstructed_frames_per_s']
            if 'fast_path' in line:
                line['fast_path']['frames_per_s'] = whole['f16x3']['reconstructed_frames_per_s']
        if world == 1 and not args.no_configs:
            try:
                line['configs'] = config_records(net.state_dict())
            except Exception as e:                                   # noqa: BLE001 -- the headline must still be printed
                import traceback
                traceback.print_exc(file=sys.stderr)
                line['configs'] = {'error': f'{type(e).__name__}: {e}'}
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(y, Phi, warm.cpu().numpy(), orig, net.state_dict(), args.cpu_budget,
                                                gpu_iters=args.warmup + args.steps, gpu=gpu_out)
        else:
            line['cpu_baseline'] = None
        emit(line, f'headline_n{world}')
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

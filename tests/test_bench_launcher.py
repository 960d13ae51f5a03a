"""bench.py's rank launcher (`--gpus N` starts N rank processes itself when no launcher did) and, on the GPU box, the
RCCL gather path of the judged metric with a real `torch.distributed` process group."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE')}
    env.update(kw)
    return env


def test_gpus_must_match_world_size():
    """a launcher that started fewer ranks than --gpus says must not silently benchmark fewer GPUs"""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0'], env=_env(WORLD_SIZE='1', RANK='0'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'WORLD_SIZE=1' in r.stderr and not r.stdout.strip()


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason='checks the too-few-GPUs failure mode')
def test_spawn_refuses_more_ranks_than_gpus():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0'], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'GPU(s) visible' in r.stderr and not r.stdout.strip()


def test_spawn_ranks_relays_rank0_json(tmp_path):
    """the launcher itself, with a stand-in rank program: N processes with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rank 0's
    JSON line relayed last on stdout, everything else on stderr, non-zero exit if any rank fails"""
    fake = tmp_path / 'fake_rank.py'
    fake.write_text(
        'import json, os, sys\n'
        'r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])\n'
        'assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0\n'
        'print("banner from rank", r)\n'
        'if "--fail" in sys.argv and r == 1: sys.exit(3)\n'
        'if r == 0: print(json.dumps({"n_gpus": w, "argv": sys.argv[1:]}))\n')
    code = (
        'import importlib.util, sys, os\n'
        f'spec = importlib.util.spec_from_file_location("bench_mod", {BENCH!r}); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n'
        'b.torch.cuda.device_count = lambda: 4\n'
        f'b.__file__ = {str(fake)!r}\n'
        'sys.exit(b.spawn_ranks(3, sys.argv[1:]))\n')
    r = subprocess.run([sys.executable, '-c', code, '--gpus', '3', '--steps', '2'], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1
    line = json.loads(out[0])
    assert line['n_gpus'] == 3 and line['argv'] == ['--gpus', '3', '--steps', '2']
    assert 'banner from rank 0' in r.stderr and 'banner from rank 2' in r.stderr
    r = subprocess.run([sys.executable, '-c', code, '--fail'], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()


@pytest.mark.gpu
def test_bench_rccl_gather_world_size_1(tmp_path):
    """the judged metric's gather goes through torch.distributed's `nccl` backend (= RCCL) with a real process group:
    world_size 1 on the one GPU of the box (SCIPNP_BENCH_FORCE_DIST=1); the log with RCCL's banner is kept for profiles/"""
    env = _env(SCIPNP_BENCH_FORCE_DIST='1', NCCL_DEBUG='VERSION', MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '1', '--steps', '3', '--warmup', '1', '--preheat', '5', '--no-cpu-baseline',
                        '--no-configs', '--no-pmc'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['ranks'] == 1 and line['units_gathered_on_rank0'] == 1
    assert 'RCCL' in line['collective'] and line['dtype'] == 'f32' and line['forms']['f16x3']['value'] > 0
    assert line['world_size'] == 1 and 'nccl' in line['backend'] and line['ranks_devices'][0][:2] == [0, 0]
    assert len(r.stdout.strip().splitlines()) == 1 and len(r.stdout) < 4200        # ONE compact line on stdout, nothing else
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'bench_rccl_world1.log'), 'w') as f:
        f.write(r.stdout + '\n--- stderr ---\n' + r.stderr)
    assert 'RCCL' in (r.stdout + r.stderr) or 'NCCL' in (r.stdout + r.stderr)      # the library's own version banner


@pytest.mark.gpu
def test_bench_spawns_two_ranks_sharing_the_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent starts the two ranks itself (before any HIP call), they
    rendezvous, time their own cubes, gather to rank 0, and the relayed line says n_gpus 2.  On the 1-GPU box the ranks
    share the device and the collectives run over gloo on host copies (RCCL refuses two ranks per device) -- the test hooks
    SCIPNP_BENCH_SHARE_GPU / SCIPNP_BENCH_BACKEND; everything else is the path the 8-GPU scaling run takes."""
    env = _env(SCIPNP_BENCH_SHARE_GPU='1', SCIPNP_BENCH_BACKEND='gloo')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '1', '--preheat', '5'], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1, out
    line = json.loads(out[0])
    assert line['n_gpus'] == 2 and line['ranks'] == 2 and line['units_gathered_on_rank0'] == 2
    assert line['cpu_baseline'] is None and line['value'] > 0 and line['forms']['f16x3']['value'] > 0
    assert line['world_size'] == 2 and [d[0] for d in line['ranks_devices']] == [0, 1] and len(out[0]) <= 4096


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: one rank per device on the real RCCL backend')
def test_bench_two_ranks_on_two_gpus_over_rccl():
    """the path the driver's N = 2, 4, 8 scaling runs take, whenever the box has the devices: `bench.py --gpus 2` starts two
    ranks, each binds ITS device before any allocation (LOCAL_RANK -> torch.cuda.set_device), the process group is `nccl`
    (= RCCL) with world size 2, every rank solves its own cube and ONE RCCL gather brings the mosaics to rank 0; the line
    names the devices the ranks sat on"""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '1', '--preheat', '5'],
                       env=_env(NCCL_DEBUG='VERSION'), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1 and len(out[0]) <= 4096, out
    line = json.loads(out[0])
    assert line['n_gpus'] == 2 and line['ranks'] == 2 and line['world_size'] == 2 and 'nccl' in line['backend']
    assert 'RCCL' in line['collective'] and line['units_gathered_on_rank0'] == 2 and line['scaling'] == 'weak'
    devs = line['ranks_devices']
    assert [d[0] for d in devs] == [0, 1] and [d[1] for d in devs] == [0, 1] and devs[0][2] != devs[1][2]     # two PCI devices
    assert line['value'] > 0 and line['cpu_baseline'] is None
    assert 'RCCL' in r.stderr or 'NCCL' in r.stderr                              # the library's own version banner


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [['--cubes', '4'], ['--config', 'tile1024']])
def test_bench_fixed_total_modes_two_ranks_sharing_the_gpu(mode):
    """BASELINE configs[3] / configs[4] as bench modes: a FIXED total of units (4 cubes here; the 16 tiles of the 1024x1024x16
    cube) split over two ranks that share the box's GPU (gloo on host copies, as above), one gather, per-rank solve and
    gather times in the line, strong scaling"""
    env = _env(SCIPNP_BENCH_SHARE_GPU='1', SCIPNP_BENCH_BACKEND='gloo')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '17', '--warmup', '3', '--preheat', '5'] + mode, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1, out
    line = json.loads(out[0])
    n_units = 16 if 'tile1024' in mode else 4
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['units_total'] == n_units
    assert line['units_per_rank'] == [n_units // 2] * 2
    assert len(line['per_rank_solve_s']) == 2 and len(line['per_rank_gather_s']) == 2 and min(line['per_rank_solve_s']) > 0
    assert line['value'] > 0 and abs(line['ms_per_step'] * line['steps'] / (1e3 * line['timed_region_s']) - 1) < 2e-3   # (4 significant digits)
    assert 0 < line['roofline']['frac'] <= 1 and len(out[0]) <= 4096
    # the roofline describes the launch this mode RUNS: 256x256x16 tiles / 512x512x8 cubes, their own FLOPs
    shape = [256, 256, 16] if 'tile1024' in mode else [512, 512, 8]
    shape[2] *= n_units // 2                                     # one launch covers the rank's whole unit batch
    assert line['roofline']['launch_shape'] == shape and line['units_batched_per_launch'] == n_units // 2
    assert abs(line['roofline']['flop_per_launch'] / (2 * 9 * 96 * 96 * (shape[0] // 2) * (shape[1] // 2) * shape[2] / 4) - 1) < 1e-3
    assert line['roofline']['traffic'] is None
    if 'tile1024' in mode:
        assert line['finetune_events_per_tile'] == 1 and line['stitched_psnr_db'] > 12


def test_driver_sigma_schedule_scales_with_steps():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod_sigma', BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    full = [b.driver_sigma(k, 25) for k in range(25)]
    assert full == [25 / 255] * 15 + [12 / 255] * 6 + [6 / 255] * 4             # the reference driver's own schedule
    half = [b.driver_sigma(k, 50) for k in range(50)]
    assert half == [25 / 255] * 30 + [12 / 255] * 12 + [6 / 255] * 8
    assert b.driver_sigma(0, 1) == 25 / 255 or b.driver_sigma(0, 1) in b.DRIVER_SIGMA


# ------------------------------------------------------------------------------------------------ the output contract
def _bench_module(name='bench_mod_emit'):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def _canned_records():
    """full records of earlier judged runs (round 3's 23 KB line, which the driver could not parse, among them)"""
    prof = os.path.join(ROOT, 'profiles')
    out = {}
    for f in ('r03f_bench_line.json', 'r03f_bench_tile1024_line.json', 'r03f_bench_cubes2_line.json', 'r02l_bench_line.json'):
        txt = [l for l in open(os.path.join(prof, f)) if l.startswith('{')][-1]
        out[f] = json.loads(txt)
    return out


def test_compact_line_fits_and_round_trips():
    """the final stdout line: <= 4096 bytes, valid JSON, carrying the contract's keys and the roofline / cpu_baseline blocks"""
    b = _bench_module()
    recs = _canned_records()
    assert len(json.dumps(recs['r03f_bench_line.json'])) > 20000                 # the record that broke the driver's parser
    for name, full in recs.items():
        line = b.compact_line(full, 'gpurun_out/bench_detail_x.json')
        assert len(line) <= 4096 and '\n' not in line, (name, len(line))
        got = json.loads(line)
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                  'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'detail'):
            assert k in got, (name, k)
        assert abs(got['value'] / full['value'] - 1) < 1e-3 and got['config']['workload']
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
            assert k in got['roofline'], (name, k)
        assert abs(got['roofline']['frac'] - full['roofline']['frac']) < 1e-3
    head = json.loads(b.compact_line(recs['r03f_bench_line.json']))
    assert head['cpu_baseline']['cores'] == 14 and head['cpu_baseline']['kind'] == 'port' and head['cpu_baseline']['parity']['f32']
    assert head['forms']['f16x3']['value'] > head['value'] and head['configs']['fastdvd_512']['f32']['ms_per_iteration'] > 0
    assert head['phi_step']['frac'] > 0 and head['phi_step']['chain']['frac'] > 0


def test_compact_line_of_an_eight_rank_job_fits():
    """the N = 8 line the driver's scaling run produces: 8 ranks_devices entries, per-rank solve / gather lists of 8, NCCL
    backend string -- still < 4096 bytes with every contract key"""
    b = _bench_module('bench_mod_emit8')
    recs = _canned_records()
    for name in ('r03f_bench_line.json', 'r03f_bench_cubes2_line.json', 'r03f_bench_tile1024_line.json'):
        full = dict(recs[name])
        full.update(n_gpus=8, ranks=8, world_size=8, backend='nccl (RCCL 2.26.6, 8 ranks over xGMI)',
                    ranks_devices=[[r, r, 0x05 + 0x10 * r] for r in range(8)], units_gathered_on_rank0=8)
        if 'per_rank' in full or 'solve_s_per_rank' in full:
            for k in ('solve_s_per_rank', 'gather_s_per_rank', 'units_per_rank'):
                if k in full:
                    full[k] = [1.2345678 + r for r in range(8)]
        line = b.compact_line(full, 'gpurun_out/bench_detail_headline_n8.json')
        assert len(line) < 4096, (name, len(line))
        got = json.loads(line)
        assert got['n_gpus'] == 8 and got['world_size'] == 8 and len(got['ranks_devices']) == 8
        for k in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'scaling', 'dtype', 'config', 'roofline', 'cpu_baseline'):
            assert k in got, (name, k)


def test_compact_line_never_exceeds_the_limit_whatever_the_record_holds():
    b = _bench_module('bench_mod_emit2')
    full = _canned_records()['r03f_bench_line.json']
    full['configs'] = {f'config_{i}': {'f32': {'ms_per_iteration': 1.0 + i, 'frac': 0.5, 'layers': [{'x': 1.0}] * 100},
                                        'parity': {'f32': {'max_rel_l2_per_iterate': 1e-7}}} for i in range(200)}
    full['data'] = 'x' * 3000
    line = b.compact_line(full, 'd.json')
    assert len(line) <= 4096
    got = json.loads(line)
    assert got['value'] == float(f"{full['value']:.4g}") and got['roofline']['frac'] > 0 and got['cpu_baseline']['value'] > 0


def test_emit_prints_exactly_one_line_and_writes_the_detail_file(tmp_path):
    """emit(): the full record lands in the detail file, stdout gets the compact line and nothing else -- even when the
    process printed log lines before (claim_stdout sends them to stderr)"""
    code = (
        'import importlib.util, json, sys\n'
        f'spec = importlib.util.spec_from_file_location("bench_mod", {BENCH!r}); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n'
        'b.claim_stdout()\n'
        'print("loss: 0.0640")\n'
        'import ctypes; ctypes.CDLL(None).puts(b"C-level banner")\n'
        f'full = json.loads([l for l in open({os.path.join(ROOT, "profiles", "r03f_bench_line.json")!r}) if l.startswith("{{")][-1])\n'
        'b.emit(full, "unit_test_n1")\n')
    r = subprocess.run([sys.executable, '-c', code], env=_env(SCIPNP_BENCH_DETAIL_DIR=str(tmp_path)), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and len(lines[0]) <= 4096
    line = json.loads(lines[0])
    assert 'loss: 0.0640' in r.stderr and 'C-level banner' in r.stderr
    detail = json.load(open(tmp_path / 'bench_detail_unit_test_n1.json'))
    assert detail['configs']['fastdvd_512']['f32']['layers'] and line['detail'].endswith('bench_detail_unit_test_n1.json')

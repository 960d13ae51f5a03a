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
    assert 'RCCL' in line['collective'] and line['dtype'] == 'f32' and line['fast_path']['dtype'] == 'f16x3'
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
    assert line['cpu_baseline'] is None and line['value'] > 0 and line['fast_path']['value'] > line['value']


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
    assert line['value'] > 0 and abs(line['ms_per_step'] * line['steps'] - 1e3 * line['timed_region_s']) < 1e-6
    assert 0 < line['roofline']['frac'] <= 1
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

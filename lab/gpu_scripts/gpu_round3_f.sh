#!/bin/bash
# round 3, last GPU checkpoint (F(4x4,3x3) Winograd is the fp32 form of the wide layers): the whole -m gpu suite, the default
# bench line, the fixed-total modes, the F(4x4) kernel's probes; tools/gpu_round3_g.sh does the rocprofv3 passes.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r03f
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/${T}_pytest.log
timeout -k 10 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/${T}_bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r03f_bench.json').read().strip().splitlines()[-1])
r=l['roofline']
print('headline', l['value'], l['ms_per_step'], 'frac', r['frac'], 'achieved', r['achieved'], 'direct-eq', r['direct_form_equivalent_TFLOPs'], 'launch ms', r['avg_launch_ms'], 'traffic', r['traffic'], r['kernel'][:40])
print('fast', l['fast_path']['value'], 'direct', l['f32_direct_form']['value'], 'f2x2', l['f32_winograd_f2x2']['value'], l['f32_winograd_f2x2']['roofline']['avg_launch_ms'], l['f32_winograd_f2x2']['parity_vs_headline'])
print('whole', l.get('whole_reconstruction'))
c=l['configs']
print('tv', c['admm_tv_256']['ms_per_iteration'])
for p in ('f32','f16x3'):
    f=c['fastdvd_512'][p]; print('fastdvd', p, f['ms_per_iteration'], 'frac', f['frac'])
    for row in f['layers'][:20]:
        print('   ', row['kernel'], row['cin'], row['cout'], row['h'], row['form'], row['launches'], round(row['avg_us'],1), 'frac', round(row['frac'],3))
    t=c['tile_256x256x16_finetune'][p]; print('tile', p, t['ms_per_iteration'], t['frac'], t['ms_per_iteration_with_finetune_event'])
print('parity', c['fastdvd_512']['parity'], c['tile_256x256x16_finetune']['parity'])
print('cpu', l['cpu_baseline']['value'], l['cpu_baseline'].get('parity'))
PY
timeout -k 10 300 python bench.py --cubes 2 --steps 10 --warmup 2 > gpurun_out/${T}_bench_cubes2.json 2> gpurun_out/${T}_bench_cubes2.err; echo "cubes rc=$?"; cut -c1-400 gpurun_out/${T}_bench_cubes2.json
timeout -k 10 400 python bench.py --config tile1024 > gpurun_out/${T}_bench_tile1024.json 2> gpurun_out/${T}_bench_tile1024.err; echo "tile1024 rc=$?"; cut -c1-400 gpurun_out/${T}_bench_tile1024.json
timeout -k 10 200 python tools/probes/wino4_check.py > gpurun_out/${T}_wino4_check.txt 2>&1; tail -6 gpurun_out/${T}_wino4_check.txt | cut -c1-170
timeout -k 10 200 python tools/probes/wino4_ablate.py > gpurun_out/${T}_wino4_ablate.txt 2>&1
timeout -k 10 200 python tools/probes/wino4_stamps.py > gpurun_out/${T}_wino4_stamps.txt 2>&1
SCIPNP_W4_ONE_PER_CU=1 timeout -k 10 200 python tools/probes/wino4_stamps.py > gpurun_out/${T}_wino4_stamps_one_per_cu.txt 2>&1
echo done

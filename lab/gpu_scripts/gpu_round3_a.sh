#!/bin/bash
# round 3, first GPU checkpoint: -m gpu suite (new ABI, new bench modes, gradient gate), default bench line with the new
# roofline reporting, the fixed-total bench modes at N = 1, and rocprofv3 kernel-trace summaries of the trainers / fp32
# FastDVDnet / DDnet iterations the round-2 verdict found untracked.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r03a
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_bench_launcher.py > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/${T}_pytest.log
ls gpurun_out/fastdvd_grad_parity_*.txt 2>/dev/null && head -5 gpurun_out/fastdvd_grad_parity_f32.txt gpurun_out/fastdvd_grad_parity_f16x3.txt
timeout -k 10 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/${T}_bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r03a_bench.json').read().strip().splitlines()[-1])
r=l['roofline']
print('headline', l['value'], l['ms_per_step'], 'frac', r['frac'], 'achieved', r['achieved'], 'direct-eq', r['direct_form_equivalent_TFLOPs'], 'launch ms', r['avg_launch_ms'], 'peak_meas', r['peak_measured'])
print('fast', l['fast_path']['value'], l['fast_path']['roofline']['frac'], 'direct', l['f32_direct_form']['value'], l['f32_direct_form']['roofline']['frac'])
print('peaks', l['measured_peaks'])
c=l['configs']
print('tv', c['admm_tv_256']['ms_per_iteration'], c['admm_tv_256']['frac'])
for p in ('f32','f16x3'):
    f=c['fastdvd_512'][p]; print('fastdvd', p, f['ms_per_iteration'], 'frac', f['frac'], 'duty', f['matrix_pipe_duty'])
    for row in f['layers'][:20]:
        print('   ', row['kernel'], row['cin'], row['cout'], row['h'], row['form'], row['launches'], round(row['avg_us'],1), 'frac', round(row['frac'],3), 'duty', round(row['matrix_pipe_duty'],3))
    t=c['tile_256x256x16_finetune'][p]; print('tile', p, t['ms_per_iteration'], t['frac'], t['ms_per_iteration_with_finetune_event'])
print('cpu', l['cpu_baseline']['value'], l['cpu_baseline'].get('parity'))
PY
timeout -k 10 300 python bench.py --cubes 2 --steps 10 --warmup 2 > gpurun_out/${T}_bench_cubes2.json 2> gpurun_out/${T}_bench_cubes2.err; echo "cubes rc=$?"; cut -c1-600 gpurun_out/${T}_bench_cubes2.json
timeout -k 10 400 python bench.py --config tile1024 > gpurun_out/${T}_bench_tile1024.json 2> gpurun_out/${T}_bench_tile1024.err; echo "tile1024 rc=$?"; cut -c1-700 gpurun_out/${T}_bench_tile1024.json
# ---- rocprofv3 kernel-trace summaries (one process each, program directly behind --)
cd /tmp && export TMPDIR=/tmp
prof() {  # tag, env assignments..., then script
  local tag=$1; shift
  local out=$GRAFT_REPO_ROOT/gpurun_out/prof_${T}_$tag
  mkdir -p $out
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/$SCRIPT > $out/trace.log 2>&1 )
  python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $out > $out/summary.txt 2>&1
  echo "== $tag"; tail -1 $out/trace.log; grep -v "at::native\|rocclr" $out/summary.txt | cut -c1-200 | head -14
}
SCRIPT=finetune_bench.py prof ffdnet_finetune_f32 SCIPNP_CONV_PRECISION=f32 FT_REPS=3
SCRIPT=finetune_bench.py prof fastdvd_finetune_f32 SCIPNP_CONV_PRECISION=f32 FT_DENOISER=fastdvd FT_REPS=2
SCRIPT=finetune_bench.py prof fastdvd_finetune_f16x3 SCIPNP_CONV_PRECISION=f16x3 FT_DENOISER=fastdvd FT_REPS=2
SCRIPT=fastdvd_bench.py prof fastdvd_f32 SCIPNP_CONV_PRECISION=f32 SCIPNP_STREAMS=1
SCRIPT=fastdvd_bench.py prof fastdvd_f16x3 SCIPNP_CONV_PRECISION=f16x3 SCIPNP_STREAMS=1
SCRIPT=ddnet_bench.py prof ddnet_f32 SCIPNP_CONV_PRECISION=f32 SCIPNP_STREAMS=1

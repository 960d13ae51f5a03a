#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the end-of-round evidence run -- GPU tests, the bench lines (headline, --cubes 8,
# --config tile1024), the rocprofv3 passes of the headline command (kernel trace + separate PMC passes, tools/profile_bench.sh) and
# the kernel-trace summaries of the fp32 FastDVDnet and FFDNet + DDnet iterations.  Everything lands under gpurun_out/ as <tag>_*;
# copy what is judged into profiles/.      usage: bash tools/gpu_final.sh <tag> [notests]
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r06z}
if [ "${2:-}" != "notests" ]; then
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/${T}_gpu_tests.txt; cat gpurun_out/${T}_gpu_tests.txt
  grep -q passed gpurun_out/${T}_gpu_tests.txt && ! grep -q failed gpurun_out/${T}_gpu_tests.txt || exit 1
fi
timeout -k 10 900 python bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err || exit 1
cp gpurun_out/bench_detail_headline_n1.json gpurun_out/${T}_bench_detail.json
tail -c 1200 gpurun_out/${T}_bench_line.json; echo
timeout -k 10 600 python bench.py --cubes 8 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_bench_cubes8_line.json 2>> gpurun_out/${T}_bench.err || exit 1
timeout -k 10 600 python bench.py --config tile1024 --no-cpu-baseline > gpurun_out/${T}_bench_tile1024_line.json 2>> gpurun_out/${T}_bench.err || exit 1
T=$T python - <<'PY'
import json, os
T = os.environ['T']
for f in ('cubes8', 'tile1024'):
    d = json.loads(open(f'gpurun_out/{T}_bench_{f}_line.json').read().strip().splitlines()[-1])
    print(f, d['metric'], d['value'], d['unit'], d['ms_per_step'], d.get('timed_region_s'))
PY
bash tools/profile_bench.sh $T > gpurun_out/${T}_profile_bench.log 2>&1; grep -v "at::native\|rocclr" gpurun_out/prof_$T/summary.txt | head -16 | cut -c1-200
cp gpurun_out/prof_$T/summary.txt gpurun_out/${T}_bench_rocprofv3_summary.txt
cp gpurun_out/prof_$T/traffic.json gpurun_out/${T}_pmc_traffic.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp SCIPNP_STREAMS=1 SCIPNP_CONV_PRECISION=f32 FD_STEPS=4 DD_STEPS=4
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_fd/trace -- python3 $GRAFT_REPO_ROOT/tools/fastdvd_bench.py > $GRAFT_REPO_ROOT/gpurun_out/${T}_fastdvd_f32.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_dd/trace -- python3 $GRAFT_REPO_ROOT/tools/ddnet_bench.py > $GRAFT_REPO_ROOT/gpurun_out/${T}_ddnet_f32.log 2>&1 )
python tools/summarize_prof.py gpurun_out/prof_${T}_fd > gpurun_out/${T}_fastdvd_f32_rocprofv3_summary.txt 2>&1
python tools/summarize_prof.py gpurun_out/prof_${T}_dd > gpurun_out/${T}_ddnet_f32_rocprofv3_summary.txt 2>&1
tail -n 2 gpurun_out/${T}_fastdvd_f32.log; tail -n 2 gpurun_out/${T}_ddnet_f32.log
head -8 gpurun_out/${T}_fastdvd_f32_rocprofv3_summary.txt
rm -rf gpurun_out/prof_${T} gpurun_out/prof_${T}_fd gpurun_out/prof_${T}_dd        # (raw traces: tens of MB; the summaries above are what is kept)
echo done

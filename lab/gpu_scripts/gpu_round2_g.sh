#!/bin/bash
# round 2 checkpoint: full -m gpu suite, smoke, default bench line, rocprofv3 profile of the bench
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02i_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r02i_pytest.log
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r02i_bench.json').read().strip().splitlines()[-1])
print(l['value'], l['dtype'], l['ms_per_step'], l['roofline']['frac'], l['roofline']['avg_launch_ms'], l['fast_path']['value'], l['f32_direct_form']['value'], l['frames_per_s'])
PY
bash tools/profile_bench.sh r02i > gpurun_out/r02i_profile.log 2>&1
echo "profile rc=$?"; head -12 gpurun_out/prof_r02i/summary.txt | cut -c1-160

#!/bin/bash
# full -m gpu suite with the Winograd fp32 path, bench line, profile
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02c_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r02c_pytest.log
timeout -k 10 600 python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err
echo "bench rc=$?"; tail -c 1500 gpurun_out/r02c_bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r02c_bench.json').read().strip().splitlines()[-1])
print(l['value'], l['dtype'], l['ms_per_step'], json.dumps(l['roofline'],indent=1)[:1800])
print('fast', l['fast_path']['value'], 'direct', l['f32_direct_form']['value'], l['f32_direct_form']['roofline']['frac'], l['f32_direct_form']['parity_vs_headline'])
print(json.dumps(l['cpu_baseline'],indent=1)[:1800]); print(l['frames_per_s'], l['whole_reconstruction'])
PY

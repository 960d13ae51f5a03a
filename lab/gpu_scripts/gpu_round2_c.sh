#!/bin/bash
# round 2, re-entry: full -m gpu suite, default bench line, rocprofv3 profile of the bench
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r02b_pytest.log
timeout -k 10 600 python bench.py > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err
echo "bench rc=$?"; tail -c 1500 gpurun_out/r02b_bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r02b_bench.json').read().strip().splitlines()[-1])
print(json.dumps(l.get('configs'),indent=1)[:3000]); print(json.dumps(l.get('cpu_baseline'),indent=1)[:1500])
print(l['value'], l['dtype'], l['roofline']['frac'], l['fast_path']['value'])
PY
bash tools/profile_bench.sh r02b > gpurun_out/r02b_profile.log 2>&1
echo "profile rc=$?"; tail -40 gpurun_out/r02b_profile.log

#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/r02b_bench.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r02b_bench.json').read().strip().splitlines()[-1])
print(json.dumps(l.get('configs'),indent=1)); print(json.dumps(l.get('cpu_baseline'),indent=1))
PY

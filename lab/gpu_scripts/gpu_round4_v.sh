#!/bin/bash
# round 4, final build: the fixed-total modes (configs[3] with 8 cubes on one GPU, configs[4] tile1024)
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=r04v
timeout -k 10 500 python bench.py --cubes 8 --no-cpu-baseline > gpurun_out/${T}_bench_cubes8_line.json 2> gpurun_out/${T}_cubes8.err; echo "cubes8 rc=$?"
timeout -k 10 500 python bench.py --config tile1024 --no-cpu-baseline > gpurun_out/${T}_bench_tile1024_line.json 2> gpurun_out/${T}_tile1024.err; echo "tile1024 rc=$?"
python3 - <<'PY'
import json
for n in ('cubes8', 'tile1024'):
    d = json.loads(open(f'gpurun_out/r04v_bench_{n}_line.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['unit'], 'ms_per_step', d['ms_per_step'], {k: d[k] for k in d if k in ('timed_region_s', 'units_batched_per_launch', 'seconds')})
PY

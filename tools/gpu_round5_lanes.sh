#!/bin/bash
# round 5: bench.py --config tile1024 with the per-tile runs after the finetune split on 1 / 2 / 4 / 8 host threads + streams
set -u
cd $GRAFT_REPO_ROOT
for l in 1 2 4 8; do
  timeout -k 10 400 python bench.py --config tile1024 --no-cpu-baseline --lanes $l > gpurun_out/r05zm_tile1024_lanes$l.json 2> gpurun_out/r05zm_tile1024_lanes$l.err || { tail -n 20 gpurun_out/r05zm_tile1024_lanes$l.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('gpurun_out/r05zm_tile1024_lanes$l.json').read().strip().splitlines()[-1])
print('lanes', $l, 'seconds', round(d['timed_region_s'], 4), 'it/s', d['value'], 'psnr', d.get('stitched_psnr_db'))
PY
done

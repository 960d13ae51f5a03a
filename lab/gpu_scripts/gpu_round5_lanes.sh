#!/bin/bash
# round 5: bench.py --config tile1024 with the per-tile runs after the finetune split on 1 / 2 host threads + streams, three times each
set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for l in 1 2; do
  timeout -k 10 400 python bench.py --config tile1024 --no-cpu-baseline --lanes $l > gpurun_out/r05zt_tile1024_lanes${l}_$rep.json 2> gpurun_out/r05zt_tile1024_lanes${l}_$rep.err || { tail -n 20 gpurun_out/r05zt_tile1024_lanes${l}_$rep.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('gpurun_out/r05zt_tile1024_lanes${l}_$rep.json').read().strip().splitlines()[-1])
print('rep', $rep, 'lanes', $l, 'seconds', round(d['timed_region_s'], 4), 'it/s', d['value'])
PY
done; done

#!/bin/bash
# ON THE GPU BOX: rocprofv3 kernel trace of tools/ddnet_bench.py (FFDNet + DDnet deep demosaicking iterations, 512x512x8)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_dd2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DD_STEPS=4 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/ddnet_bench.py > $OUT/trace.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_dd2/trace/runc/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
g = collections.OrderedDict()
for r in rows:
    if 'conv3x3' not in r['Kernel_Name']:
        continue
    key = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    g.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in g.values())
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k[0][8:]:36s} grid {k[1]:>7s} x {k[2]:>4s} x {k[3]:>4s}  n={len(v):4d}  avg {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:7.2f} ms  {100*sum(v)/tot:5.1f} %')
PY
tail -1 $OUT/trace.log

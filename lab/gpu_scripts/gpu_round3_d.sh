#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03d_tv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_iter_trace.py > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cut -c1-200 $OUT/summary.txt | head -20
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/prof_r03d_tv/trace/**/*kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# steady-state window of each mode: print 12 consecutive launches from the middle of each half
names = [r['Kernel_Name'][:40] for r in rows]
n = len(rows)
for lo in (n // 4, 3 * n // 4):
    print('--- window at', lo)
    for i in range(lo, lo + 12):
        r, p = rows[i], rows[i - 1]
        print(f"{(int(r['Start_Timestamp']) - int(p['End_Timestamp'])) / 1e3:7.2f} us gap   {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.2f} us   {r['Kernel_Name'][:70]}")
PY

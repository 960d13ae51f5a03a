#!/bin/bash
# round 5: SQ counters of the two kernels of the ADMM-TV iteration (separate --pmc pass with --kernel-trace only)
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05zz_tv_pmc
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_iter_trace.py > $OUT.log 2>&1 || { tail -n 5 $OUT.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05zz_tv_pmc'
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    if 'tv_band_kernel' in k and ', 3, ' not in k:
        continue
    if not any(s in k for s in ('tv_band_kernel', 'pm_dual_project')):
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k.replace('scipnp::', '')[:70])
    print('   ' + '  '.join(f'{n}={v:.3g}' for n, v in sorted(m.items())))
    if m.get('SQ_WAVE_CYCLES'):
        print(f"   VALU-active share of wave cycles {m.get('SQ_ACTIVE_INST_VALU', 0) / m['SQ_WAVE_CYCLES']:.3f}, waiting share {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}, "
              f"VALU instructions per wave-launch {m.get('SQ_INSTS_VALU', 0):.3g}, LDS instructions {m.get('SQ_INSTS_LDS', 0):.3g}")
PY
find $OUT -name "*.csv" -size +1M -delete

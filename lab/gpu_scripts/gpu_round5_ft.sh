#!/bin/bash
# round 5: kernel trace of the FFDNet finetune event (512x512x8, fp32): per-kernel statistics and the timeline of one event
set -u
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r05zu_ft
FT_REPS=${FT_REPS:-4} timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05zu_ft -- python3 $GRAFT_REPO_ROOT/tools/finetune_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r05zu_ft.log 2>&1 || exit 1
grep "iteration with" $GRAFT_REPO_ROOT/gpurun_out/r05zu_ft.log
python3 $GRAFT_REPO_ROOT/tools/trace_stats.py $GRAFT_REPO_ROOT/gpurun_out/r05zu_ft 30 | cut -c1-170 | tee $GRAFT_REPO_ROOT/gpurun_out/r05zu_ft_summary.txt
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05zu_ft/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
# the last event: from the last pm_pre_denoise to the end
idx = [i for i, r in enumerate(rows) if 'pm_pre_denoise' in r[2]]
seg = rows[idx[-1]:]
busy = sum(e - s for s, e, _ in seg) / 1e3
span = (seg[-1][1] - seg[0][0]) / 1e3
gaps = sum(max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)) / 1e3
small = [(e - s) / 1e3 for s, e, _ in seg if (e - s) < 20000]
gl = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1))[::-1][:12]
for g_, a_, b_ in gl:
    print(f'  gap {g_:7.1f} us after {a_} before {b_}')
print(f'last iteration with event: {len(seg)} launches, span {span:.0f} us, kernels {busy:.0f} us, gaps {gaps:.0f} us; launches under 20 us: {len(small)} totalling {sum(small):.0f} us')
PY
find $GRAFT_REPO_ROOT/gpurun_out/r05zu_ft -name "*.csv" -size +1M -delete

#!/bin/bash
# round 4: kernel profile of the FFDNet iteration with its finetune event (512x512x8, fp32)
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof_ft
R=$PWD
cd /tmp && export TMPDIR=/tmp
FT_REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ft -- python3 $R/tools/finetune_bench.py > $R/gpurun_out/r04o_ft.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob('gpurun_out/prof_ft/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in rows.values())
with open('gpurun_out/r04o_ffdnet_finetune_kernels.txt', 'w') as o:
    o.write(f'FFDNet iteration with finetune event x 6 (+ setup), fp32, 512x512x8: {tot/1e3:.2f} ms of kernels\n')
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:30]:
        o.write(f'{len(v):6d} {sum(v)/1e3:9.3f} ms {sum(v)/len(v):9.2f} us {100*sum(v)/tot:6.2f} %  {k[:110]}\n')
print(open('gpurun_out/r04o_ffdnet_finetune_kernels.txt').read())
PY

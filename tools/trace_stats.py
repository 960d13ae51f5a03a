#!/usr/bin/env python3
"""per-kernel duration statistics of every *kernel_trace.csv below a rocprofv3 output directory (argv[1])"""
import csv, glob, os, sys
from collections import defaultdict
rows = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
total = sum(sum(v) for v in rows.values()) or 1.0
print(f'{"calls":>6} {"total_ms":>10} {"avg_us":>9} {"min_us":>9} {"max_us":>9} {"%":>6}  kernel')
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f'{len(v):6d} {sum(v) / 1e3:10.3f} {sum(v) / len(v):9.2f} {min(v):9.2f} {max(v):9.2f} {100 * sum(v) / total:6.2f}  '
          f'{k.replace("scipnp::", "")[:110]}')

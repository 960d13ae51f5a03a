#!/usr/bin/env python3
"""In-order view of a rocprofv3 --kernel-trace run: the last N dispatches with grid, workgroup and duration.
usage: python tools/trace_sequence.py <trace dir> [N]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
prev_end = None
for r in rows[-n:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    prev_end = e
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  wgs {int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // (int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])):>6d} x {r['Workgroup_Size_X']:>4s}  lds {r['LDS_Block_Size']:>6s} vgpr {r['VGPR_Count']:>3s}+{r['Accum_VGPR_Count']:>3s}  "
          f"{r['Kernel_Name'].replace('void ', '')[:90]}")

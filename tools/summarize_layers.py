#!/usr/bin/env python3
"""Per-layer-shape view of a tools/profile_conv_layers.sh run: dispatches of the split conv kernels grouped by
(template instance, grid size), with duration, device cycles and the SQ counters of the separate PMC passes.
usage: python tools/summarize_layers.py gpurun_out/prof_cob1"""
import collections, csv, glob, sys

base = sys.argv[1]


def load(p):
    rows = collections.OrderedDict()
    for f in glob.glob(f'{base}/{p}/runc/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'conv3x3_c8s' not in r['Kernel_Name']:
                continue
            d = rows.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'].split('(')[0].replace('void ', ''),
                                                        'grid': int(r['Grid_Size'])})
            d[r['Counter_Name']] = float(r['Counter_Value'])
            d['dur'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = collections.OrderedDict()
    for d in rows.values():
        g.setdefault((d['name'], d['grid']), []).append(d)
    return g


ga, gb, gc = load('pmc_SQ_VALU_MFMA_BUSY_CYCLES'), load('pmc_SQ_WAIT_INST_LDS'), load('pmc_GRBM_GUI_ACTIVE')
m = lambda L, f: sum(x.get(f, 0) for x in L) / max(1, len(L))      # noqa: E731
print('kernel<COB,TAG,STRIDE,SHUF,PB,NW,WS>  threads  n  us  device-cycles(all XCDs)  MFMA-busy  MFMA-busy/(4*cycles*32CU)  '
      'insts: MFMA LDS VALU')
for key, lst in ga.items():
    lb, lc = gb.get(key, [{}]), gc.get(key, [{}])
    cyc = m(lc, 'GRBM_GUI_ACTIVE')
    busy = m(lst, 'SQ_VALU_MFMA_BUSY_CYCLES')
    frac = busy / (cyc / 8 * 1024) if cyc else float('nan')     # 1024 matrix pipes; GRBM_GUI_ACTIVE sums the 8 XCDs
    print(f"{key[0][11:]:34s} {key[1]:8d} {len(lst):4d} {m(lst, 'dur'):7.1f} {cyc:10.0f} {busy:10.3e} {frac:5.2f}   "
          f"{m(lb, 'SQ_INSTS_MFMA'):.2e} {m(lb, 'SQ_INSTS_LDS'):.2e} {m(lb, 'SQ_INSTS_VALU'):.2e}")

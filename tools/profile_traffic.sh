#!/bin/bash
# ON THE GPU BOX: HBM traffic per launch (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, gfx950 correction applied by
# tools/summarize_prof.py) and durations, per kernel instance AND grid, of any of the bench tools:
#   bash tools/profile_traffic.sh tools/fastdvd_bench.py fd
set -u
SCRIPT=$1; TAG=${2:-traffic}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  FD_STEPS=3 DD_STEPS=3 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$pass -- python3 $GRAFT_REPO_ROOT/$SCRIPT > $OUT/pmc_$pass.log 2>&1
done
python3 - <<PY
import csv, glob, collections
out = '$OUT'
def load(p):
    g = collections.OrderedDict()
    for f in glob.glob(f'{out}/pmc_{p}/runc/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != p: continue
            key = (r['Kernel_Name'].split('(')[0].replace('void ', '').replace('scipnp::', '')[:44], int(r['Grid_Size']))
            g.setdefault(key, []).append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    return g
F, W = load('FETCH_SIZE'), load('WRITE_SIZE')
rows = []
for k, v in F.items():
    # FETCH_SIZE / WRITE_SIZE are in KiB; the fetch counter sees half the bytes on gfx950 (MI355X_MICROARCH.md)
    fetch = sum(a for a, _ in v) / len(v) * 1024 * 2
    wr = sum(a for a, _ in W.get(k, [(0, 0)])) / max(1, len(W.get(k, [1]))) * 1024
    us = sum(b for _, b in v) / len(v)
    rows.append((us * len(v), k, len(v), us, fetch, wr))
for tot, k, n, us, fetch, wr in sorted(rows, reverse=True)[:28]:
    print(f'{k[0]:44s} threads {k[1]:9d} n={n:4d} {us:8.1f} us  fetch {fetch/1e6:8.1f} MB  write {wr/1e6:8.1f} MB  {(fetch+wr)/us/1e6:6.2f} TB/s')
PY

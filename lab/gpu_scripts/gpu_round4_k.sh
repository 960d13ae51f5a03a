#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04k}
for S in 0 4 8 12 16; do
  echo "== SCIPNP_W4_PERSIST=1 SCIPNP_W4_STAGGER=$S"
  SCIPNP_W4_PERSIST=1 SCIPNP_W4_STAGGER=$S timeout -k 10 200 python tools/wino_bench.py 2>&1 | grep "fp32 wino F(4x4)" | cut -c1-200
done | tee gpurun_out/${T}_persist_stagger.txt
echo done

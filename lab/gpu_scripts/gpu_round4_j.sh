#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04j}
export SCIPNP_W4_PERSIST=1
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wino or f4 or winograd" > gpurun_out/${T}_pytest_persist.log 2>&1
echo "persist conv pytest rc=$?"; tail -6 gpurun_out/${T}_pytest_persist.log | cut -c1-250
for P in 0 1 0 1; do
  echo "== SCIPNP_W4_PERSIST=$P"
  SCIPNP_W4_PERSIST=$P timeout -k 10 200 python tools/wino_bench.py 2>&1 | grep "F(4x4)" | cut -c1-200
done | tee gpurun_out/${T}_persist_ab.txt
echo done

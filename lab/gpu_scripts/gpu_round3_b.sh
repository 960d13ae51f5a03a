#!/bin/bash
# round 3: Winograd kernels after the accumulator-reuse-distance reorder -- parity tests first, then timings / ablations
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "winograd or wino or ffdnet" > gpurun_out/r03b_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r03b_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/wino_bench.py 2>&1 | tee gpurun_out/r03b_wino_bench.txt
timeout -k 10 200 python tools/probes/winop_ablate.py 2>&1 | tee gpurun_out/r03b_winop_ablate.txt

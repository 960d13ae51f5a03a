#!/bin/bash
# round 4: F(2x2)-domain weight gradient with its raw values two chunks ahead
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad" 2>&1 | tail -5 > gpurun_out/r04n_tests.txt; cat gpurun_out/r04n_tests.txt
grep -q passed gpurun_out/r04n_tests.txt && ! grep -q failed gpurun_out/r04n_tests.txt
for s in 85 80; do WW_SLABS=$s timeout -k 10 200 python tools/probes/wgrad_f32_bench.py 2>&1 | grep "Winograd wgrad" >> gpurun_out/r04n_wgrad_bench.txt; done
cat gpurun_out/r04n_wgrad_bench.txt

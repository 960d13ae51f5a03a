#!/bin/bash
# round 4: trainers skip the F(2x2) packing of layers that run on the F(4x4) kernel
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_configs.py tests/test_gpu_units.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r04x_tests.txt; cat gpurun_out/r04x_tests.txt
grep -q passed gpurun_out/r04x_tests.txt && ! grep -q failed gpurun_out/r04x_tests.txt
FT_REPS=9 timeout -k 10 300 python tools/finetune_bench.py 2>&1 | grep "iteration with finetune" | tee gpurun_out/r04x_event.txt

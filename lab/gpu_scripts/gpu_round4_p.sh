#!/bin/bash
# round 4: full GPU suite after the weight-gradient work
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04p_gpu_tests.txt; cat gpurun_out/r04p_gpu_tests.txt
grep -q passed gpurun_out/r04p_gpu_tests.txt && ! grep -q failed gpurun_out/r04p_gpu_tests.txt

#!/bin/bash
# round 4, final state: GPU suite, bench line, rocprof summary of the bench command
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=r04r
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/${T}_gpu_tests.txt; cat gpurun_out/${T}_gpu_tests.txt
grep -q passed gpurun_out/${T}_gpu_tests.txt && ! grep -q failed gpurun_out/${T}_gpu_tests.txt
timeout -k 10 900 python bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; cut -c1-900 gpurun_out/${T}_bench_line.json
cp gpurun_out/bench_detail_headline_n1.json gpurun_out/${T}_bench_detail.json

#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04g}
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "wino or conv or ffdnet" > gpurun_out/${T}_pytest_conv.log 2>&1
echo "conv pytest rc=$?"; tail -5 gpurun_out/${T}_pytest_conv.log | cut -c1-250
timeout -k 10 200 python tools/wino_bench.py > gpurun_out/${T}_wino_bench.txt 2>&1; tail -7 gpurun_out/${T}_wino_bench.txt | cut -c1-200
timeout -k 10 200 python tools/probes/wino4_stamps.py > gpurun_out/${T}_wino4_stamps.txt 2>&1; head -22 gpurun_out/${T}_wino4_stamps.txt | cut -c1-200
timeout -k 10 300 python bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-configs --no-pmc > gpurun_out/${T}_bench_quick.json 2> gpurun_out/${T}_bench_quick.err; echo "bench rc=$?"; cut -c1-1600 gpurun_out/${T}_bench_quick.json
timeout -k 10 200 python tools/tv_bench.py > gpurun_out/${T}_tv_bench.txt 2>&1; sed -n 16,30p gpurun_out/${T}_tv_bench.txt
echo done

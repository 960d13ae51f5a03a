#!/bin/bash
# first GPU call of round 2: launcher tests, default bench line, bench under rocprofv3
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_bench_launcher.py -m gpu -x -q 2>&1 | tail -15
python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/r02a_bench.err; head -c 6000 gpurun_out/r02a_bench.json

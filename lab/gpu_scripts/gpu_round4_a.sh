#!/bin/bash
# round 4, first GPU checkpoint: the whole -m gpu suite on the f32 default, the default bench line (compact line contract),
# the fixed-total modes, and the rocprofv3 kernel trace of the headline command.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04a}
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/${T}_pytest.log
timeout -k 10 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$? stdout lines=$(wc -l < gpurun_out/${T}_bench.json) bytes=$(wc -c < gpurun_out/${T}_bench.json)"; tail -3 gpurun_out/${T}_bench.err | cut -c1-300
cat gpurun_out/${T}_bench.json
cp gpurun_out/bench_detail_headline_n1.json gpurun_out/${T}_bench_detail.json
timeout -k 10 300 python bench.py --cubes 2 --steps 10 --warmup 2 > gpurun_out/${T}_bench_cubes2.json 2> gpurun_out/${T}_bench_cubes2.err; echo "cubes rc=$?"; cat gpurun_out/${T}_bench_cubes2.json
timeout -k 10 400 python bench.py --config tile1024 > gpurun_out/${T}_bench_tile1024.json 2> gpurun_out/${T}_bench_tile1024.err; echo "tile1024 rc=$?"; cat gpurun_out/${T}_bench_tile1024.json
echo done

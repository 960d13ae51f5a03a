#!/bin/bash
# round 4: unit-batch tests first, then the rest of the -m gpu suite, the bench line and the fixed-total modes
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04c}
timeout -k 10 600 python -m pytest tests/test_gpu_units.py -m gpu -q -x > gpurun_out/${T}_pytest_units.log 2>&1
echo "units pytest rc=$?"; tail -12 gpurun_out/${T}_pytest_units.log | cut -c1-250
timeout -k 10 1100 python -m pytest tests -m gpu -q --deselect tests/test_gpu_units.py > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/${T}_pytest.log | cut -c1-250
timeout -k 10 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/${T}_bench.json)"; tail -3 gpurun_out/${T}_bench.err | cut -c1-300
cat gpurun_out/${T}_bench.json
cp gpurun_out/bench_detail_headline_n1.json gpurun_out/${T}_bench_detail.json
for C in 2 8; do
timeout -k 10 400 python bench.py --cubes $C --steps 10 --warmup 2 > gpurun_out/${T}_bench_cubes$C.json 2> gpurun_out/${T}_bench_cubes$C.err; echo "cubes$C rc=$?"; cut -c1-700 gpurun_out/${T}_bench_cubes$C.json; tail -2 gpurun_out/${T}_bench_cubes$C.err | cut -c1-300
done
timeout -k 10 400 python bench.py --cubes 8 --steps 10 --warmup 2 --no-unit-batch > gpurun_out/${T}_bench_cubes8_seq.json 2> gpurun_out/${T}_bench_cubes8_seq.err; echo "cubes8 seq rc=$?"; cut -c1-300 gpurun_out/${T}_bench_cubes8_seq.json
timeout -k 10 400 python bench.py --config tile1024 > gpurun_out/${T}_bench_tile1024.json 2> gpurun_out/${T}_bench_tile1024.err; echo "tile1024 rc=$?"; cut -c1-900 gpurun_out/${T}_bench_tile1024.json; tail -2 gpurun_out/${T}_bench_tile1024.err | cut -c1-300
timeout -k 10 400 python bench.py --config tile1024 --no-unit-batch > gpurun_out/${T}_bench_tile1024_seq.json 2> gpurun_out/${T}_bench_tile1024_seq.err; echo "tile1024 seq rc=$?"; cut -c1-400 gpurun_out/${T}_bench_tile1024_seq.json
echo done

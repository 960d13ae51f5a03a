#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04d}
timeout -k 10 600 python -m pytest tests/test_gpu_units.py -m gpu -q > gpurun_out/${T}_pytest_units.log 2>&1
echo "units pytest rc=$?"; tail -12 gpurun_out/${T}_pytest_units.log | cut -c1-250
timeout -k 10 600 python -m pytest tests/test_gpu_solver.py tests/test_bench_launcher.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/${T}_pytest.log | cut -c1-250
timeout -k 10 200 python tools/probes/tv_units_probe.py > gpurun_out/${T}_tv_units.txt 2>&1; cat gpurun_out/${T}_tv_units.txt
TV_IQA=0 timeout -k 10 200 python tools/probes/tv_units_probe.py > gpurun_out/${T}_tv_units_noiqa.txt 2>&1; cat gpurun_out/${T}_tv_units_noiqa.txt
cd /tmp && export TMPDIR=/tmp
export TV_UNITS=8
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_tv8/trace -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_units_probe.py > $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_tv8/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_tv8 > $GRAFT_REPO_ROOT/gpurun_out/${T}_tv8_rocprofv3_summary.txt 2>&1
grep -v "at::native\|rocclr" $GRAFT_REPO_ROOT/gpurun_out/${T}_tv8_rocprofv3_summary.txt | cut -c1-200 | head -12
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --config tile1024 > gpurun_out/${T}_bench_tile1024.json 2> gpurun_out/${T}_bench_tile1024.err; echo "tile1024 rc=$?"; cut -c1-900 gpurun_out/${T}_bench_tile1024.json
echo done

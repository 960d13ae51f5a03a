#!/bin/bash
# round 5: the new configuration GPU test + the banded TV kernel's cost per inner iteration (rocprofv3 kernel trace)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_units.py -x -q -m gpu > gpurun_out/r05x_units.txt 2>&1; tail -n 3 gpurun_out/r05x_units.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r05x_tvsweep
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05x_tvsweep -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_band_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/r05x_tvsweep.log 2>&1
cd $GRAFT_REPO_ROOT
grep "planes of\|done" gpurun_out/r05x_tvsweep.log
python3 tools/probes/tv_band_sweep.py gpurun_out/r05x_tvsweep | tee gpurun_out/r05x_tvsweep_kernels.txt
find gpurun_out/r05x_tvsweep -name "*.csv" -size +1M -delete

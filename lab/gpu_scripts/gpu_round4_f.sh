#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04f}
timeout -k 10 600 python -m pytest tests/test_gpu_units.py -m gpu -q > gpurun_out/${T}_pytest_units.log 2>&1
echo "units pytest rc=$?"; tail -12 gpurun_out/${T}_pytest_units.log | cut -c1-250
timeout -k 10 200 python tools/probes/tv_units_probe.py > gpurun_out/${T}_tv_units.txt 2>&1; cat gpurun_out/${T}_tv_units.txt
SCIPNP_TV_PLANE_BATCH=0 timeout -k 10 200 python tools/probes/tv_units_probe.py > gpurun_out/${T}_tv_units_banded.txt 2>&1; cat gpurun_out/${T}_tv_units_banded.txt
SCIPNP_TV_PLANE_BATCH=1 timeout -k 10 200 python tools/probes/tv_units_probe.py > gpurun_out/${T}_tv_units_plane.txt 2>&1; cat gpurun_out/${T}_tv_units_plane.txt
echo done

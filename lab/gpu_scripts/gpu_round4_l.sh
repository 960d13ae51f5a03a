#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04l}
timeout -k 10 600 python -m pytest tests/test_gpu_ddnet_train.py -m gpu -q -x > gpurun_out/${T}_pytest_ddnet.log 2>&1
echo "ddnet pytest rc=$?"; tail -30 gpurun_out/${T}_pytest_ddnet.log | cut -c1-300
timeout -k 10 600 python -m pytest tests/test_harness.py -m gpu -q > gpurun_out/${T}_pytest_harness.log 2>&1
echo "harness pytest rc=$?"; tail -8 gpurun_out/${T}_pytest_harness.log | cut -c1-300
echo done

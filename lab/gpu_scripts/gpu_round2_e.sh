#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02d_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r02d_pytest.log
bash tools/profile_bench.sh r02c > gpurun_out/r02c_profile.log 2>&1
echo "profile rc=$?"; head -30 gpurun_out/prof_r02c/summary.txt

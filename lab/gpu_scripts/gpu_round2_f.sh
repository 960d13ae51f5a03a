#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02e_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r02e_pytest.log
timeout 200 python tools/tv_bench.py 2>&1 | grep -v "ADMM-TV iteration" > gpurun_out/r02e_tv_bench.txt; cat gpurun_out/r02e_tv_bench.txt
bash tools/profile_tv.sh r02e > gpurun_out/r02e_tv_profile.log 2>&1; tail -30 gpurun_out/r02e_tv_profile.log

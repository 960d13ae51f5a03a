#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-r05zb}
timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_solver.py tests/test_gpu_units.py tests/test_gpu_configs.py tests/test_gpu_cabi_host.py -x -q -m gpu -k "tv or TV or admm or units or config" > gpurun_out/${TAG}_tv_tests.txt 2>&1
rc=$?; tail -n 5 gpurun_out/${TAG}_tv_tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_round5_tv_d.sh $TAG

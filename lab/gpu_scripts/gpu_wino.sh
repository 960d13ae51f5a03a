#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "winograd" 2>&1 | tail -15
timeout 300 python tools/wino_bench.py 2>&1 | tail -8

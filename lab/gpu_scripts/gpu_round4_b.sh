#!/bin/bash
# round 4: the whole -m gpu suite (no bench)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04b}
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/${T}_pytest.log | cut -c1-250

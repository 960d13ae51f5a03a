#!/bin/bash
# round 4 checkpoint: the whole -m gpu suite, the default bench line, the rocprofv3 passes of the headline command
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04i}
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/${T}_pytest.log | cut -c1-250
timeout -k 10 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/${T}_bench.json)"; cat gpurun_out/${T}_bench.json | cut -c1-2500
cp gpurun_out/bench_detail_headline_n1.json gpurun_out/${T}_bench_detail.json
bash tools/profile_bench.sh $T > gpurun_out/${T}_profile_bench.log 2>&1; grep -v "at::native\|rocclr" gpurun_out/prof_$T/summary.txt | head -24 | cut -c1-220
cp gpurun_out/prof_$T/summary.txt gpurun_out/${T}_bench_rocprofv3_summary.txt
cp gpurun_out/prof_$T/traffic.json gpurun_out/${T}_pmc_traffic.json 2>/dev/null
echo done

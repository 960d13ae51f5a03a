#!/bin/bash
# round 4, final build: the rocprofv3 passes of the headline command (kernel trace + separate PMC passes)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r04w
bash tools/profile_bench.sh $T > gpurun_out/${T}_profile_bench.log 2>&1; grep -v "at::native\|rocclr" gpurun_out/prof_$T/summary.txt | head -14 | cut -c1-200
cp gpurun_out/prof_$T/summary.txt gpurun_out/${T}_bench_rocprofv3_summary.txt
cp gpurun_out/prof_$T/traffic.json gpurun_out/${T}_pmc_traffic.json 2>/dev/null
echo done

#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${T}_tv8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TV_UNITS=8
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_units_probe.py > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/${T}_tv8_rocprofv3_summary.txt 2>&1
grep -v "at::native\|rocclr" $GRAFT_REPO_ROOT/gpurun_out/${T}_tv8_rocprofv3_summary.txt | cut -c1-220 | head -12
export TV_UNITS=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${T}_tv1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_units_probe.py > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/${T}_tv1_rocprofv3_summary.txt 2>&1
grep -v "at::native\|rocclr" $GRAFT_REPO_ROOT/gpurun_out/${T}_tv1_rocprofv3_summary.txt | cut -c1-220 | head -8
echo done

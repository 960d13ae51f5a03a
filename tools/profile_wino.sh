#!/bin/bash
# ON THE GPU BOX: PMC profile of tools/wino_bench.py (FFDNet body layer: fp32 direct vs fp32 Winograd forms)
set -u
TAG=${1:-wino}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/tools/wino_bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- $CMD > $OUT/pmc_$name.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -v "at::native\|to_c8\|rocclr\|pack_" $OUT/summary.txt | cut -c1-300

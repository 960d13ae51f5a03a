# GPU box: PMC passes over tools/probes/wgrad_f32_bench.py for conv3x3_wgrad_wino4_kernel (separate passes, kernel-trace only)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_wg4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WW4_SLABS=${WW4_SLABS:-28}
CMD="python3 $GRAFT_REPO_ROOT/tools/probes/wgrad_f32_bench.py"
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$i -- $CMD > $OUT/pmc_$i.log 2>&1
  echo "pass $i ($pass): rc=$?"
done
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT $OUT/traffic.json > $OUT/summary.txt 2>&1
grep -A1 "wgrad_wino4_kernel\|wgrad_wino_kernel" $OUT/summary.txt | cut -c1-700

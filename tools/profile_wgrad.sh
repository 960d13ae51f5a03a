set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_wg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/tools/finetune_bench.py"
for pass in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- $CMD > $OUT/pmc_$name.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT $OUT/traffic.json > $OUT/summary.txt 2>&1
python3 -c "
import json; t=json.load(open('$OUT/traffic.json'))
for k,v in t.items():
    if 'wgrad' in k or 'conv3x3_c8s_kernel<3, 0' in k or 'bgrad' in k: print(k[:60], {a: round(b/1e6,1) if 'bytes' in a else round(b) for a,b in v.items()})"
grep -A1 "wgrad_split_kernel<3>" $OUT/summary.txt | grep "SQ_\|GRBM" | cut -c1-300

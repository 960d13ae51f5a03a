#!/bin/bash
# round 4: F(4x4)-domain weight gradient as the FFDNet trainer's default -- GPU suite, bench with its config records
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=r04u
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/${T}_gpu_tests.txt; cat gpurun_out/${T}_gpu_tests.txt
grep -q passed gpurun_out/${T}_gpu_tests.txt && ! grep -q failed gpurun_out/${T}_gpu_tests.txt
timeout -k 10 900 python bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
cp gpurun_out/bench_detail_headline_n1.json gpurun_out/${T}_bench_detail.json
python3 -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_line.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], json.dumps(d['configs']['tile_256x256x16_finetune']))"
for v in f2 f4; do
  echo "SCIPNP_F32_WGRAD=$v" >> gpurun_out/${T}_event.txt
  SCIPNP_F32_WGRAD=$v FT_REPS=7 timeout -k 10 300 python tools/finetune_bench.py 2>&1 | grep "iteration with finetune" >> gpurun_out/${T}_event.txt
done
cat gpurun_out/${T}_event.txt

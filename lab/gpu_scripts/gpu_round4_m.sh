#!/bin/bash
# round 4, F(4x4)-domain weight gradient (csrc/wgrad_wino4.hip): parity, timing against the F(2x2) form, ablations, the FFDNet
# trainer's event with either form
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad" 2>&1 | tail -5 > gpurun_out/r04m_tests.txt; cat gpurun_out/r04m_tests.txt
grep -q passed gpurun_out/r04m_tests.txt && ! grep -q failed gpurun_out/r04m_tests.txt
timeout -k 10 200 python tools/probes/wgrad_f32_bench.py > gpurun_out/r04m_wgrad_f32_bench.txt 2>&1; cat gpurun_out/r04m_wgrad_f32_bench.txt
timeout -k 10 300 python tools/probes/wgrad4_ablate.py > gpurun_out/r04m_wgrad4_ablate.txt 2>&1; cat gpurun_out/r04m_wgrad4_ablate.txt
for v in f2 f4; do
  echo "SCIPNP_F32_WGRAD=$v" >> gpurun_out/r04m_wgrad4_event.txt
  SCIPNP_F32_WGRAD=$v FT_REPS=7 timeout -k 10 300 python tools/finetune_bench.py 2>&1 | grep "iteration with finetune" >> gpurun_out/r04m_wgrad4_event.txt
done
cat gpurun_out/r04m_wgrad4_event.txt

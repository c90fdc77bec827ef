#!/bin/bash
# whole GPU suite under each non-default GEMM mode (the default runs in run_all_gpu.sh)
mkdir -p gpurun_out
for m in split fp32; do
  PRH_GEMM=$m timeout -k 10 600 python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests_$m.log 2>&1
  echo "$m: $(tail -1 gpurun_out/gpu_tests_$m.log)"
  grep -A25 "^E  \|Error" gpurun_out/gpu_tests_$m.log | head -30
done

#!/bin/bash
# numerical-flake triage: same test sequence several times
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  PRH_GEMM=split16 timeout -k 10 200 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py tests/test_model_gpu.py -q -m gpu -x > gpurun_out/flake_$i.log 2>&1
  echo "run $i: $(tail -1 gpurun_out/flake_$i.log) $(grep -o 'assert [0-9.e-]* == 0.0' gpurun_out/flake_$i.log | head -1)"
done

#!/bin/bash
# numerical-flake triage: same test sequence, several times, per GEMM mode
mkdir -p gpurun_out
for mode in split split16; do
  for i in 1 2 3 4; do
    PRH_GEMM=$mode timeout -k 10 200 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py tests/test_model_gpu.py -q -m gpu -x > gpurun_out/flake_${mode}_$i.log 2>&1
    echo "$mode run $i: $(tail -1 gpurun_out/flake_${mode}_$i.log) $(grep -o 'assert [0-9.e-]* == 0.0' gpurun_out/flake_${mode}_$i.log | head -1)"
  done
done

#!/bin/bash
# GEMM/encoder/model parity in the default mode, then per-launch durations of the encoder;
# PRH_TN_TR=1 / 0 select the earlier wgrad cores for comparison
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py tests/test_model_gpu.py -q -m gpu -x > gpurun_out/h2_tests.log 2>&1 || { grep -v "^$" gpurun_out/h2_tests.log | tail -60; exit 1; }
tail -1 gpurun_out/h2_tests.log
for v in default 1; do
  if [ $v = default ]; then unset PRH_TN_TR; else export PRH_TN_TR=$v; fi
  timeout -k 10 200 python scripts/encoder_bench.py ${1:-1024} 1024 3 > gpurun_out/enc_$v.log 2>&1 || { tail -20 gpurun_out/enc_$v.log; exit 1; }
  if grep -qi fault gpurun_out/enc_$v.log; then echo FAULT; exit 1; fi
  echo "== PRH_TN_TR=$v"; grep "gemm_tn_h2\|encoder+proj\|GEMM launches" gpurun_out/enc_$v.log
done

#!/bin/bash
# GEMM/encoder/model parity in the default mode, then per-launch durations of the encoder
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py tests/test_model_gpu.py -q -m gpu -x > gpurun_out/h2_tests.log 2>&1 || { grep -v "^$" gpurun_out/h2_tests.log | tail -60; exit 1; }
tail -1 gpurun_out/h2_tests.log
timeout -k 10 200 python scripts/encoder_bench.py ${1:-1024} 1024 3 > gpurun_out/enc.log 2>&1 || { tail -20 gpurun_out/enc.log; exit 1; }
if grep -qi fault gpurun_out/enc.log; then echo FAULT; exit 1; fi
cat gpurun_out/enc.log

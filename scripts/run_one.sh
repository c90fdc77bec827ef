#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
PRH_GEMM=split16 timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | grep -v "^$" | tail -80 > gpurun_out/one.log
cat gpurun_out/one.log

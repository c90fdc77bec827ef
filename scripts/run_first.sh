#!/bin/bash
# first-process-on-a-fresh-box behaviour of the split16 parity tests with an alternative library build
mkdir -p gpurun_out
[ -n "$1" ] && export PRH_LIB_PATH=$PWD/exp/lib_$1.so
PRH_GEMM=split16 timeout -k 10 200 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py tests/test_model_gpu.py -q -m gpu -x > gpurun_out/first.log 2>&1
echo "first process ($1): $(tail -1 gpurun_out/first.log) $(grep -o 'assert [0-9.e-]* == 0.0' gpurun_out/first.log | head -1)"

#!/bin/bash
# whole GPU suite (default GEMM mode), full log kept
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x -s > gpurun_out/gpu_tests.log 2>&1 || { grep -v "^$" gpurun_out/gpu_tests.log | tail -60; exit 1; }
grep "rel-L2\|vs exact" gpurun_out/gpu_tests.log; tail -2 gpurun_out/gpu_tests.log

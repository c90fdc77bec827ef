#!/bin/bash
# same-box A/B of two builds of the library: parity tests on the default build, then the bench
# with each build in turn (twice).  usage: run_ab.sh <other.so>
set -o pipefail
ALT=$1
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -1 || exit 1
for rep in 1 2; do
for lib in default $ALT; do
  if [ $lib = default ]; then unset PRH_LIB_PATH; else export PRH_LIB_PATH=$PWD/$lib; fi
  timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernels 40 > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err || { tail -20 gpurun_out/bench_ab.err; exit 1; }
  if grep -qi fault gpurun_out/bench_ab.err; then echo FAULT; exit 1; fi
  echo "== $lib: $(python -c "
import json
d=json.loads(open('gpurun_out/bench_ab.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], 'ms/step loss', d['loss'], 'gemm ms', d['roofline']['hip_gemm_ms_per_step'])")"
  grep "attn_\|pos_\|gemm_nt_h2<0,3> K=1024 N=1984" gpurun_out/bench_ab.err | cut -c9-80
done
done

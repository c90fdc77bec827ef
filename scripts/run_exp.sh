#!/bin/bash
# timing-only experiments: variants of the library with deliberately wrong addressing
set -o pipefail
mkdir -p gpurun_out
for v in base HOT_A HOT_W HOT_AW; do
  if [ $v = base ]; then unset PRH_LIB_PATH; else export PRH_LIB_PATH=$PWD/exp/lib_$v.so; fi
  PRH_GEMM=split16 timeout -k 10 200 python scripts/encoder_bench.py 1024 1024 3 > gpurun_out/exp_$v.log 2>&1 || { tail -20 gpurun_out/exp_$v.log; exit 1; }
  if grep -qi fault gpurun_out/exp_$v.log; then echo FAULT $v; exit 1; fi
  echo "== $v"; grep "h2<1,1> K=1984\|h2<2,3> K=1024 N=1984\|h2<0,0> K=1024\|tn_h2<2,1> Mo=1024 Ni=1984\|encoder+proj" gpurun_out/exp_$v.log
done

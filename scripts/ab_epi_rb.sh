#!/bin/bash
# same-box A/B (bf16 mode): epilogue operand batches of a whole 32-row block with raw 16-bit operands (default) vs
# the round-2 batches of 16 rows (libprh_rb4.so = the same sources built with -DPRH_EPI_RB16=4)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rb
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py tests/test_gemm_gpu.py tests/test_encoder_gpu.py -x -q > gpurun_out/rb/tests.log 2>&1
rc=$?; tail -3 gpurun_out/rb/tests.log; [ $rc -ne 0 ] && exit $rc
for v in rb8 rb4 rb8 rb4; do
  if [ $v = rb4 ]; then export PRH_LIB_PATH=$GRAFT_REPO_ROOT/pointnet_refine_amd/libprh_rb4.so; else unset PRH_LIB_PATH; fi
  python bench.py --gemm bf16 --steps 6 --warmup 2 --kernels 16 --no-parity --no-workloads --no-cpu-baseline > gpurun_out/rb/ab_$v.json 2> gpurun_out/rb/ab_$v.txt || exit 1
  echo "== $v: $(python -c "import json;d=json.loads(open('gpurun_out/rb/ab_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms/step"
  grep " x " gpurun_out/rb/ab_$v.txt | head -14
done

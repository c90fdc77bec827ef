#!/bin/bash
# same-box A/B: bf16-mode plain-operand NT GEMMs on the DMA + phase-split core (default) vs the register-staged core (PRH_B16_DMA=0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b16d
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py tests/test_gemm_gpu.py -x -q > gpurun_out/b16d/tests.log 2>&1
rc=$?; tail -4 gpurun_out/b16d/tests.log; [ $rc -ne 0 ] && exit $rc
for v in dma base dma base; do
  if [ $v = base ]; then export PRH_B16_DMA=0; else unset PRH_B16_DMA; fi
  python bench.py --gemm bf16 --steps 6 --warmup 2 --kernels 16 --no-parity --no-workloads --no-cpu-baseline > gpurun_out/b16d/ab_$v.json 2> gpurun_out/b16d/ab_$v.txt || exit 1
  echo "== $v: $(python -c "import json;d=json.loads(open('gpurun_out/b16d/ab_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms/step"
  grep " x " gpurun_out/b16d/ab_$v.txt | head -16
done

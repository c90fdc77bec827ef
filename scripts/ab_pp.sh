#!/bin/bash
# same-box A/B: phase-split ("ping-pong") k-loop of the h2 NT core (PRH_H2_PP=1) vs the one-barrier loop
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pp
PRH_H2_PP=1 timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py tests/test_model_gpu.py tests/test_oracle_fp64_gpu.py -x -q > gpurun_out/pp/tests.log 2>&1
rc=$?; tail -3 gpurun_out/pp/tests.log; [ $rc -ne 0 ] && exit $rc
for v in pp base pp base; do
  if [ $v = pp ]; then export PRH_H2_PP=1; else unset PRH_H2_PP; fi
  python bench.py --steps 6 --warmup 2 --kernels 14 --no-parity --no-workloads --no-cpu-baseline > gpurun_out/pp/ab_$v.json 2> gpurun_out/pp/ab_$v.txt || exit 1
  echo "== $v: $(python -c "import json;d=json.loads(open('gpurun_out/pp/ab_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms/step"
  grep " x " gpurun_out/pp/ab_$v.txt | head -14
done

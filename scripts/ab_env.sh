#!/bin/bash
# same-box A/B of one environment switch of the library.  usage: ab_env.sh VAR [split16|bf16] [kernels]
cd $GRAFT_REPO_ROOT
VAR=$1; MODE=${2:-bf16}; NK=${3:-12}
mkdir -p gpurun_out/abe
for v in on off on off; do
  if [ $v = on ]; then export $VAR=1; else unset $VAR; fi
  python bench.py --gemm $MODE --steps 6 --warmup 2 --kernels $NK --no-parity --no-workloads --no-cpu-baseline > gpurun_out/abe/ab_$v.json 2> gpurun_out/abe/ab_$v.txt || exit 1
  echo "== $VAR $v: $(python -c "import json;d=json.loads(open('gpurun_out/abe/ab_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms/step"
  grep " x " gpurun_out/abe/ab_$v.txt | head -$NK
done

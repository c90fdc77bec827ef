#!/bin/bash
# same-box A/B: PRH_STAGGER=1 (odd workgroups of the first generation start half a tile late) vs in-step start
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stg
MODE=${1:-bf16}
for v in stg base stg base; do
  if [ $v = stg ]; then export PRH_STAGGER=1; else unset PRH_STAGGER; fi
  python bench.py --gemm $MODE --steps 6 --warmup 2 --kernels 12 --no-parity --no-workloads --no-cpu-baseline > gpurun_out/stg/ab_$v.json 2> gpurun_out/stg/ab_$v.txt || exit 1
  echo "== $v: $(python -c "import json;d=json.loads(open('gpurun_out/stg/ab_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms/step"
  grep " x " gpurun_out/stg/ab_$v.txt | head -12
done

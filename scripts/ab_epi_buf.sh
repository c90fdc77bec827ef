#!/bin/bash
# same-box A/B: buffer-addressed, branch-free vector epilogue (default) vs the round-2 epilogue (libprh_rb4.so, built
# from the sources before the rewrite).  usage: ab_epi_buf.sh [split16|bf16]
cd $GRAFT_REPO_ROOT
MODE=${1:-split16}
mkdir -p gpurun_out/eb
for v in new old new old; do
  if [ $v = old ]; then export PRH_LIB_PATH=$GRAFT_REPO_ROOT/pointnet_refine_amd/libprh_rb4.so; else unset PRH_LIB_PATH; fi
  python bench.py --gemm $MODE --steps 6 --warmup 2 --kernels 14 --no-parity --no-workloads --no-cpu-baseline > gpurun_out/eb/ab_${MODE}_$v.json 2> gpurun_out/eb/ab_${MODE}_$v.txt || exit 1
  echo "== $MODE $v: $(python -c "import json;d=json.loads(open('gpurun_out/eb/ab_${MODE}_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms/step"
  grep " x " gpurun_out/eb/ab_${MODE}_$v.txt | head -14
done

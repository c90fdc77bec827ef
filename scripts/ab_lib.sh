#!/bin/bash
# same-box A/B of two builds of the library: the in-tree one vs $1 (a .so under pointnet_refine_amd/).  usage: ab_lib.sh libprh_head.so [split16|bf16]
cd $GRAFT_REPO_ROOT
ALT=$GRAFT_REPO_ROOT/pointnet_refine_amd/$1; MODE=${2:-split16}
mkdir -p gpurun_out/abl
for v in new alt new alt; do
  if [ $v = alt ]; then export PRH_LIB_PATH=$ALT; else unset PRH_LIB_PATH; fi
  python bench.py --gemm $MODE --steps 6 --warmup 2 --kernels 14 --no-parity --no-workloads --no-cpu-baseline > gpurun_out/abl/ab_${MODE}_$v.json 2> gpurun_out/abl/ab_${MODE}_$v.txt || exit 1
  echo "== $MODE $v: $(python -c "import json;d=json.loads(open('gpurun_out/abl/ab_${MODE}_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms/step"
  grep " x " gpurun_out/abl/ab_${MODE}_$v.txt | head -14
done

#!/bin/bash
# same-box A/B: direct (transposed-MFMA) epilogue of the h2 NT core vs the LDS-transposing vector epilogue
cd $GRAFT_REPO_ROOT
for v in direct vec direct vec; do
  if [ $v = vec ]; then export PRH_LIB_PATH=$GRAFT_REPO_ROOT/pointnet_refine_amd/libprh_vec_epi.so; else unset PRH_LIB_PATH; fi
  python bench.py --steps 6 --warmup 2 --kernels 14 --no-parity --no-workloads --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.txt
  echo "== $v: $(python -c "import json;d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms/step"
  grep " x " gpurun_out/ab_$v.txt | head -14
done

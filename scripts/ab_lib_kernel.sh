#!/bin/bash
# same-box A/B of two builds of the library by rocprofv3 kernel stats of one kernel.  usage: ab_lib_kernel.sh libprh_head.so <kernel-substring> [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ALT=$GRAFT_REPO_ROOT/pointnet_refine_amd/$1; KN=$2; shift 2
for v in new alt new alt; do
  if [ $v = alt ]; then export PRH_LIB_PATH=$ALT; else unset PRH_LIB_PATH; fi
  rm -rf gpurun_out/abk_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_$v -- python bench.py --steps 3 --warmup 1 --no-parity --no-workloads --no-cpu-baseline "$@" > gpurun_out/abk_$v.json 2> gpurun_out/abk_$v.err || { tail -3 gpurun_out/abk_$v.err; exit 1; }
  echo "== $v: $(python -c "import json;d=json.loads(open('gpurun_out/abk_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms/step"
  grep -h "$KN" gpurun_out/abk_$v/*/*kernel_stats.csv | cut -c1-160 | head -4
done

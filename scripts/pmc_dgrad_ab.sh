#!/bin/bash
# FETCH_SIZE of the fusion dgrad with and without the partial-block shortcut (PRH_DGRAD_PARTIAL), same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  export PRH_DGRAD_PARTIAL=$v
  mkdir -p gpurun_out/dgab_$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/dgab_$v/FETCH_SIZE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-workloads > gpurun_out/dgab_$v/log.txt 2>&1
  echo "== PRH_DGRAD_PARTIAL=$v"; python scripts/pmc_traffic.py gpurun_out/dgab_$v gpurun_out/dgab_$v.json | head -4
  rm -rf gpurun_out/dgab_$v
done

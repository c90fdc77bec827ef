#!/bin/bash
# FETCH_SIZE of the bench's GEMM kernels with and without wgrad pacing (PRH_TN_PACE), on one
# box, plus the live kernel times of both settings.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python scripts/tn_probe.py 4194304 1024 1984 2 2>/dev/null | grep -v finite
for pace in 0 1; do
  OUT=pmc_bench_pace$pace
  mkdir -p gpurun_out/$OUT
  PRH_TN_PACE=$pace rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$OUT/FETCH_SIZE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/$OUT/FETCH_SIZE.log 2>&1 || { echo "pass failed"; tail -3 gpurun_out/$OUT/FETCH_SIZE.log; exit 1; }
  if grep -q "fault" gpurun_out/$OUT/FETCH_SIZE.log; then echo FAULT; exit 1; fi
  echo "== pace=$pace"
  python scripts/pmc_traffic.py gpurun_out/$OUT gpurun_out/$OUT/traffic.json | grep tn_tr
  rm -rf gpurun_out/$OUT/FETCH_SIZE/*/*kernel_trace.csv
  PRH_TN_PACE=$pace python bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernels 12 2>&1 >/dev/null | grep "gemm_tn\|timed"
done

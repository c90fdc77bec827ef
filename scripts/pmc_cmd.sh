#!/bin/bash
# HBM traffic of any command's kernels from the PMC counters, collected as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes, kernel-trace only.
# usage (on the GPU box): bash scripts/pmc_cmd.sh <outdir under gpurun_out> python3 <script> [args...]
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/$OUT/$c -- "$@" > gpurun_out/$OUT/$c.log 2>&1 || { echo "pass $c failed"; tail -3 gpurun_out/$OUT/$c.log; }
  if grep -q "fault" gpurun_out/$OUT/$c.log; then echo FAULT; exit 1; fi
  echo "pass $c done"
done

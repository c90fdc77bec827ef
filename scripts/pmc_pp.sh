#!/bin/bash
# SQ counters (matrix-pipe busy, held clock, wait shares) of the h2 NT core: phase-split loop vs one-barrier loop
set -e
B=${1:-512}; N=${2:-1024}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in pp base; do
  if [ $v = pp ]; then export PRH_H2_PP=1; else unset PRH_H2_PP; fi
  O=gpurun_out/pmc_$v; mkdir -p $O
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq1 -- python scripts/encoder_bench.py $B $N 1 > $O/sq1.log 2>&1 || { echo "pass failed"; tail -5 $O/sq1.log; exit 1; }
  echo "== $v"; python scripts/pmc_summary.py $O 1.0 2>&1 | grep -B1 "gemm_nt_h2" | cut -c1-250
done

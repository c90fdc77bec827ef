#!/bin/bash
set -e
OUT=$1; B=${2:-512}; N=${3:-1024}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$OUT
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/$OUT/$name -- python scripts/encoder_bench.py $B $N 1 > gpurun_out/$OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/$OUT/$name.log; }
  if grep -q "fault" gpurun_out/$OUT/$name.log; then echo FAULT; exit 1; fi
  echo "pass $name done"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum

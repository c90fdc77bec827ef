#!/bin/bash
# SQ counters of the bf16-mode GEMM kernels (matrix-pipe busy, held clock, wait shares); optional env passes through
set -e
B=${1:-512}; N=${2:-1024}; TAG=${3:-b16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PRH_GEMM=bf16
O=gpurun_out/pmc_$TAG; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq1 -- python scripts/encoder_bench.py $B $N 1 > $O/sq1.log 2>&1 || { echo "pass failed"; tail -5 $O/sq1.log; exit 1; }
python scripts/pmc_summary.py $O 0.5 2>&1 | cut -c1-250

#!/bin/bash
# SQ counters of the fused eval encoder kernel (both precisions): matrix-pipe busy share, held clock, wait shares
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in fp16 fp32; do
  O=gpurun_out/pmc_fused_$v; mkdir -p $O
  if [ $v = fp16 ]; then G="--gemm bf16"; else G=""; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq1 -- python bench.py --encoder-only --eval $G --steps 2 --warmup 1 > $O/sq1.log 2>&1 || { echo "pass failed"; tail -5 $O/sq1.log; exit 1; }
  echo "== fused eval encoder, $v"; python scripts/pmc_summary.py $O 5.0 encoder_fused 2>&1 | cut -c1-250
done

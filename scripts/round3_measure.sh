#!/bin/bash
# End-of-round measurements (GPU box).  Part A: tests + the two driver-style bench lines + B=32 lines.
# Part B: rocprofv3 kernel stats + PMC traffic passes (bench, both modes).  Part C: fused eval kernel PMC +
# SQ counters.  usage: bash scripts/round3_measure.sh A|B|C     (outputs under gpurun_out/r03m/)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m
mkdir -p $O
case "$1" in
A)
  timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
  python bench.py --steps 20 --warmup 5 > $O/bench_split16.json 2> $O/bench_split16.err; echo "default rc=$?"
  python bench.py --steps 20 --warmup 5 --gemm bf16 --no-cpu-baseline --no-workloads > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bf16 rc=$?"
  python bench.py --batch 32 --points 2048 --steps 50 --warmup 10 --no-cpu-baseline > $O/b32.json 2>/dev/null
  python bench.py --batch 32 --points 2048 --steps 50 --warmup 10 --no-cpu-baseline --graph > $O/b32g.json 2>/dev/null
  python bench.py --batch 4096 --points 2048 --decoder-chunk 2048 --steps 5 --warmup 2 --no-cpu-baseline --no-workloads --no-parity --gemm bf16 > $O/b4096n2048_bf16.json 2>/dev/null
  for f in bench_split16 bench_bf16 b32 b32g b4096n2048_bf16; do echo $f; cut -c90-300 $O/$f.json; done
  ;;
B)
  bash scripts/profile_bench.sh r03m_split16 --no-cpu-baseline --no-workloads --no-parity > $O/prof_split16.log 2>&1; echo "prof split16 rc=$?"
  bash scripts/profile_bench.sh r03m_bf16 --gemm bf16 --no-cpu-baseline --no-workloads --no-parity > $O/prof_bf16.log 2>&1; echo "prof bf16 rc=$?"
  bash scripts/pmc_traffic.sh r03m/pmc_split16 --no-parity --no-workloads && python scripts/pmc_traffic.py gpurun_out/r03m/pmc_split16 $O/pmc_traffic_B4096_split16.json
  bash scripts/pmc_traffic.sh r03m/pmc_bf16 --gemm bf16 --no-parity --no-workloads && python scripts/pmc_traffic.py gpurun_out/r03m/pmc_bf16 $O/pmc_traffic_B4096_bf16.json
  ;;
C)
  bash scripts/pmc_cmd.sh r03m/pmc_fused_fp16 python bench.py --encoder-only --eval --gemm bf16 --steps 2 --warmup 1 && python scripts/pmc_traffic.py gpurun_out/r03m/pmc_fused_fp16 $O/pmc_traffic_fused_eval_fp16.json 1e8
  bash scripts/pmc_cmd.sh r03m/pmc_fused_fp32 python bench.py --encoder-only --eval --steps 2 --warmup 1 && python scripts/pmc_traffic.py gpurun_out/r03m/pmc_fused_fp32 $O/pmc_traffic_fused_eval_fp32.json 1e8
  bash scripts/pmc_quick.sh r03m/sq_split16 512 1024 && python scripts/pmc_summary.py gpurun_out/r03m/sq_split16 > $O/pmc_sq_lds_l2_B512_split16.txt 2>&1
  tail -30 $O/pmc_sq_lds_l2_B512_split16.txt
  ;;
D)
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03m/prof_b32 -- python bench.py --batch 32 --points 2048 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-workloads > $O/prof_b32.json 2> $O/prof_b32.err
  python scripts/summarize_rocprof.py gpurun_out/r03m/prof_b32 $O/r03m_b32 && head -45 $O/r03m_b32_kernel_stats_top40.csv | cut -c1-160
  ;;
esac

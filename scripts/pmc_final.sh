#!/bin/bash
# final PMC traffic passes of the bench command (both modes) with the shipped sources
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m
mkdir -p $O
bash scripts/pmc_traffic.sh r03m/pmc_split16 --no-parity --no-workloads && python scripts/pmc_traffic.py gpurun_out/r03m/pmc_split16 $O/pmc_traffic_B4096_split16.json | head -6
bash scripts/pmc_traffic.sh r03m/pmc_bf16 --gemm bf16 --no-parity --no-workloads && python scripts/pmc_traffic.py gpurun_out/r03m/pmc_bf16 $O/pmc_traffic_B4096_bf16.json | head -6

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for g in split16 split; do
  timeout -k 10 500 python bench.py --steps 3 --warmup 1 --gemm $g --no-cpu-baseline > gpurun_out/bench_$g.json 2> gpurun_out/bench_$g.err || { tail -20 gpurun_out/bench_$g.err; exit 1; }
  if grep -qi fault gpurun_out/bench_$g.err; then echo FAULT; exit 1; fi
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_$g.json").read().strip().splitlines()[-1])
print("$g", d["value"], "seg/s", d["ms_per_step"], "ms/step loss", d["loss"], "mem", d["max_mem_gb"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"], "gemm ms", d["roofline"]["hip_gemm_ms_per_step"])
PY
done

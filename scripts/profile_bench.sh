#!/bin/bash
# bench.py under rocprofv3 --kernel-trace --stats, then the summaries committed under profiles/
# usage: profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python bench.py --steps 3 --warmup 1 "$@" > gpurun_out/prof_$TAG/bench.json 2> gpurun_out/prof_$TAG/bench.err || { tail -20 gpurun_out/prof_$TAG/bench.err; exit 1; }
if grep -qi fault gpurun_out/prof_$TAG/bench.err; then echo FAULT; exit 1; fi
python scripts/summarize_rocprof.py gpurun_out/prof_$TAG gpurun_out/prof_$TAG/$TAG
tail -1 gpurun_out/prof_$TAG/bench.json > gpurun_out/prof_$TAG/${TAG}_benchline.json
head -c 1500 gpurun_out/prof_$TAG/${TAG}_benchline.json; echo
head -32 gpurun_out/prof_$TAG/${TAG}_kernel_stats_top40.csv | cut -c1-150

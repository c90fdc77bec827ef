#!/bin/bash
# XCD map, then L2 fetch traffic of one fusion-wgrad-shaped launch (scripts/tn_probe.py)
# without and with pacing (PRH_TN_PACE), undisturbed and with the odd tiles started
# ~100 us late (PRH_TN_SKEW, diagnostic) - what pacing is there to repair
set -e
OUT=${1:-pmc_tn}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$OUT
python scripts/tn_probe.py 4194304 1024 1984 2 2>/dev/null | grep -v finite
for skew in 0 50; do
for pace in 0 1; do
  c=FETCH_SIZE; tag=skew${skew}_pace${pace}_$c
  PRH_TN_SKEW=$skew PRH_TN_PACE=$pace rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/$OUT/$tag -- python scripts/tn_probe.py > gpurun_out/$OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 gpurun_out/$OUT/$tag.log; exit 1; }
  if grep -q "fault" gpurun_out/$OUT/$tag.log; then echo FAULT; exit 1; fi
  echo "pass $tag done: $(grep 'ms per call' gpurun_out/$OUT/$tag.log)"
done
done
python - "$OUT" <<'PY'
import collections, csv, glob, os, sys
d = os.path.join("gpurun_out", sys.argv[1])
for pas in sorted(os.listdir(d)):
    for f in glob.glob(os.path.join(d, pas, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(float); disp = set()
        for r in csv.DictReader(open(f)):
            if "tn_tr" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        n = max(len(disp), 1)
        out = {k: v / n for k, v in acc.items()}
        if "FETCH_SIZE" in out:
            print(pas, f"fetch per launch {out['FETCH_SIZE'] * 2 / 1e6:.1f} GB (x2 corrected), {n} launches")
        else:
            print(pas, {k: f"{v:.4g}" for k, v in out.items()}, "hit rate %.3f" % (out.get("TCC_HIT_sum", 0) / max(out.get("TCC_HIT_sum", 0) + out.get("TCC_MISS_sum", 0), 1)))
PY

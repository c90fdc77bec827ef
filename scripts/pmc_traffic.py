#!/usr/bin/env python
"""Per-launch HBM traffic of the library's large GEMM kernels from scripts/pmc_traffic.sh
passes.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide
coalesced streaming read (MI355X_MICROARCH.md, HBM section), so it is doubled.
usage: pmc_traffic.py gpurun_out/<dir> out.json"""
import collections
import csv
import glob
import json
import os
import sys

import hashlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def kernel_sources_sha():
    """Fingerprint of EVERY file of csrc/ (the file set _lib._SOURCES compiles), the same function as
    bench.kernel_sources_sha: bench.py drops the lookup (traffic null) when the tree's fingerprint
    differs from the one recorded here."""
    h = hashlib.sha256()
    d_ = os.path.join(ROOT, "pointnet_refine_amd", "csrc")
    for f in sorted(os.listdir(d_)):
        if f.endswith((".hpp", ".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d_, f), "rb").read())
    return h.hexdigest()[:16]


d, out = sys.argv[1], sys.argv[2]
MIN_BYTES = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9      # skip launches that fetch less than this
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(d, c, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    per = collections.defaultdict(float)
    meta = {}
    for r in csv.DictReader(open(f[0])):
        if "prh::" not in r["Kernel_Name"]:
            continue
        per[r["Dispatch_Id"]] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
    for did, v in per.items():
        acc[meta[did]][c].append(v)
res = []
for (name, wgs), m in acc.items():
    fe = m.get("FETCH_SIZE", [])
    wr = m.get("WRITE_SIZE", [])
    if not fe or max(fe) * 2048 < MIN_BYTES:
        continue
    # launches of one template with one grid can still be different layers (every wgrad runs
    # 768 workgroups): keep the average and the largest launch of the group (dispatch order is
    # the same in both passes, so index i of the write pass is the same launch)
    imax = max(range(len(fe)), key=lambda i: fe[i])
    res.append({"kernel": name, "workgroups": wgs, "launches": len(fe),
                "fetch_bytes_per_launch": sum(fe) / len(fe) * 1024 * 2,
                "write_bytes_per_launch": (sum(wr) / len(wr) * 1024) if wr else None,
                "fetch_bytes_largest_launch": fe[imax] * 1024 * 2,
                "write_bytes_largest_launch": (wr[imax] * 1024) if (wr and len(wr) == len(fe)) else None,
                "kernel_sources_sha": kernel_sources_sha(),
                "note": "FETCH_SIZE x2 (gfx950 half-count correction), WRITE_SIZE as read; separate --pmc passes"})
res.sort(key=lambda r: -r["fetch_bytes_largest_launch"])
json.dump(res, open(out, "w"), indent=1)
for r in res[:16]:
    print(f"{r['kernel'][:48]:48s} wgs={r['workgroups']:7d} n={r['launches']:3d} fetch avg {r['fetch_bytes_per_launch']/1e9:7.2f} GB "
          f"largest {r['fetch_bytes_largest_launch']/1e9:7.2f} GB  write avg {(r['write_bytes_per_launch'] or 0)/1e9:7.2f} "
          f"largest {(r['write_bytes_largest_launch'] or 0)/1e9:7.2f} GB")

#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats (csv) output directory into the small
summary files committed under profiles/: the per-kernel stats table (top 40) and, for the
library's own GEMM kernels, per-(kernel, grid) averages from the dispatch trace so a
template instantiation that serves several layer shapes can be read per shape.

usage: summarize_rocprof.py <dir with *_kernel_stats.csv, *_kernel_trace.csv> <out prefix>"""
import collections
import csv
import glob
import os
import sys


def main(d, out):
    stats = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    trace = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out + "_kernel_stats_top40.csv", "w") as f:
        f.write("name,calls,total_ms,avg_ms,min_ms,max_ms,pct\n")
        for r in rows[:40]:
            f.write('"%s",%s,%.3f,%.4f,%.4f,%.4f,%.2f\n' % (
                r["Name"][:140].replace('"', "'"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6,
                float(r["Percentage"])))
        f.write('"TOTAL (all kernels)",,%.3f,,,,100\n' % (tot / 1e6))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        n = r["Kernel_Name"]
        if "prh::" not in n:
            continue
        key = (n[:90], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["VGPR_Count"], r["LDS_Block_Size"])
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    with open(out + "_prh_by_grid.csv", "w") as f:
        f.write("kernel,workgroups,vgpr,lds_bytes,calls,avg_ms,min_ms,max_ms,total_ms\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%s,%s,%d,%.4f,%.4f,%.4f,%.3f\n' % (k[0], k[1], k[2], k[3], len(v),
                                                              sum(v) / len(v), min(v), max(v), sum(v)))
    print("wrote", out + "_kernel_stats_top40.csv", out + "_prh_by_grid.csv")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

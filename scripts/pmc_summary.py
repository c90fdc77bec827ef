#!/usr/bin/env python
"""Join the per-pass counter CSVs of scripts/pmc_encoder.sh by dispatch and print one line
per large GEMM dispatch.  usage: pmc_summary.py gpurun_out/<dir> [min_ms] [name substrings, comma separated]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
KERNELS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["prh::gemm"]      # kernel-name substrings to report
# durations from the kernel trace of the first pass, counters keyed by (kernel name, occurrence index)
data = collections.OrderedDict()
for pas in sorted(os.listdir(d)):
    pd = os.path.join(d, pas)
    if not os.path.isdir(pd):
        continue
    cc = glob.glob(os.path.join(pd, "**", "*counter_collection.csv"), recursive=True)
    if not cc:
        continue
    seen = collections.Counter()
    disp_index = {}
    for r in csv.DictReader(open(cc[0])):
        did = r["Dispatch_Id"]
        name = r["Kernel_Name"]
        if did not in disp_index:
            disp_index[did] = (name, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""), seen[(name, r.get("Grid_Size", ""))])
            seen[(name, r.get("Grid_Size", ""))] += 1
        key = disp_index[did]
        data.setdefault(key, {})[r["Counter_Name"]] = data.get(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    kt = glob.glob(os.path.join(pd, "**", "*kernel_trace.csv"), recursive=True)
    if kt and pas == "sq1":
        seen = collections.Counter()
        for r in csv.DictReader(open(kt[0])):
            g = str(int(r["Grid_Size_X"]))
            key = (r["Kernel_Name"], g, seen[(r["Kernel_Name"], g)])
            seen[(r["Kernel_Name"], g)] += 1
            data.setdefault(key, {})["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for key, c in data.items():
    if not any(t in key[0] for t in KERNELS) or c.get("ms", 0) < min_ms or key[2] != (1 if "prh::gemm" in key[0] else 0):
        continue
    ms = c["ms"]
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (ms * 1e-3) / 1e9 if c.get("GRBM_GUI_ACTIVE") else 0
    wc = c.get("SQ_WAVE_CYCLES", 1)
    # matrix-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (MI355X_MICROARCH.md: = 32 x N_mfma for
    # 32x32x16, 16 x N for 16x16x32); chip-wide busy fraction = that sum / (1024 SIMDs x clock x duration)
    simd_cycles = 1024.0 * clk * 1e9 * ms * 1e-3 if clk else 0.0
    mfma_util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles if simd_cycles else 0.0
    print(f"   [matrix pipe busy {mfma_util:5.2f} of the SIMD cycles at the held clock; x clock = {mfma_util * clk:4.2f} GHz-equivalent of 2.4]")
    print(f"{key[0][10:52]:42s} grid={key[1]:>9s} {ms:8.3f} ms clk~{clk:4.2f}GHz "
          f"| wave-cycle shares: wait_any {c.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {c.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} "
          f"active {c.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} valu {c.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} lds {c.get('SQ_ACTIVE_INST_LDS',0)/wc:5.2f} "
          f"| mfma_busy/busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(c.get('SQ_BUSY_CYCLES',1),1):6.3f} "
          f"| lds conflict/active {c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1):5.3f} "
          f"| L2 hit {c.get('TCC_HIT_sum',0)/max(c.get('TCC_HIT_sum',0)+c.get('TCC_MISS_sum',0),1):5.3f} "
          f"| fetch {c.get('FETCH_SIZE',0)*2/1e6:7.2f} GB(x2 corr) write {c.get('WRITE_SIZE',0)/1e6:7.2f} GB "
          f"| insts valu {c.get('SQ_INSTS_VALU',0):.3g} mfma {c.get('SQ_INSTS_MFMA',0):.3g} vmem_rd {c.get('SQ_INSTS_VMEM_RD',0):.3g}")

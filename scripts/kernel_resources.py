"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (registers, spills, scratch)
for the GEMM cores:  hipcc ... -c -Rpass-analysis=kernel-resource-usage prh_lib.hip 2> res.txt;
python scripts/kernel_resources.py res.txt [filter]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else "s3_kernel"
recs = re.split(r"remark: Function Name: ", text)[1:]
for rec in recs:
    name = rec.split(" ")[0]
    if flt not in name:
        continue
    def g(key):
        m = re.search(re.escape(key) + r": (\d+)", rec)
        return m.group(1) if m else "?"
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        dem = name
    print("%-60s VGPR %s AGPR %s SGPRspill %s VGPRspill %s scratch %s" % (
        dem[:60], g("VGPRs"), g("AGPRs"), g("SGPRs Spill"), g("VGPRs Spill"), g("ScratchSize [bytes/lane]")))

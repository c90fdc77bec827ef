#!/usr/bin/env python
"""List VMEM issues, vmcnt waits and barriers inside the MFMA loop of one kernel.
usage: asm_waits.py lib.s <mangled-name-substring>"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l) and sys.argv[2] in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
lines = src[start:end]
for hh in [i for i, l in enumerate(lines) if "Loop Header" in l]:
    lab = lines[hh].split(":")[0]
    e = next((j for j in range(hh, len(lines)) if re.search(r"s_c?branch\w* " + re.escape(lab) + r"\b", lines[j])), None)
    if e and any("v_mfma" in x for x in lines[hh:e]):
        body = lines[hh:e + 1]
        print(lab, "len", len(body), "mfma", sum("v_mfma" in x for x in body), "valu",
              sum(bool(re.match(r"\s+v_(?!mfma)", x)) for x in body))
        for k, x in enumerate(body):
            if re.search(r"s_waitcnt vmcnt|buffer_load|global_load|s_barrier|scratch_", x):
                nxt = body[k + 1].strip()[:50] if "s_waitcnt" in x else ""
                print("   %4d %-62s %s" % (k, x.strip()[:62], ("-> " + nxt) if nxt else ""))
        break

#!/usr/bin/env python
"""Instruction mix of the loops of selected kernels in a hipcc -S dump.
usage: asm_loops.py file.s substring [substring...]"""
import re
import sys

VALU = r"^\s+v_(?!mfma)"
lines = open(sys.argv[1]).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
starts.append((len(lines), "END"))
for (a, name), (b, _) in zip(starts, starts[1:]):
    if not any(s in name for s in sys.argv[2:]):
        continue
    body = lines[a:b]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    print(name[:70], "total: mfma", sum("v_mfma" in x for x in body), "scratch", sum("scratch_" in x for x in body))
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            lb = body[labels[m.group(1)]:i]
            cnt = lambda pat: sum(bool(re.search(pat, x)) for x in lb)
            valu = cnt(VALU)
            print(f"   loop {labels[m.group(1)]}-{i}: mfma={cnt('v_mfma')} scratch={cnt('scratch_')} "
                  f"valu={valu} ds_read={cnt('ds_read')} ds_write={cnt('ds_write')} "
                  f"gload={cnt('global_load')} waitcnt={cnt('s_waitcnt')} barrier={cnt('s_barrier')} nop={cnt('s_nop')}")

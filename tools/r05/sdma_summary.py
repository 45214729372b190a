#!/usr/bin/env python3
"""Counts, per phase of tools/r05/sdma_probe.py, which copy engine the HIP runtime's copy log names for H2D and D2H copies."""
import re, sys
from collections import Counter, defaultdict
phase, cnt, init = "start", defaultdict(Counter), []
for line in open(sys.argv[1], errors="ignore"):
    if line.startswith("#### PHASE"):
        phase = line.split("PHASE", 1)[1].strip()
    elif "Max SDMA" in line:
        init.append(line.strip()[-80:])
    else:
        m = re.search(r"HSA Copy copy_engine=(0x[0-9a-f]+), dst=(0x[0-9a-f]+), src=(0x[0-9a-f]+), size=(\d+), forceSDMA=(\d), engineType=(\d)", line)
        if m:
            eng, size, et = m.group(1), int(m.group(4)), m.group(6)
            kind = "small" if size < 4096 else "packet/words" if size < 3_000_000 else "frame piece"
            cnt[phase][("copy", kind, "engineType=" + et, "engine=" + eng)] += 1
        q = re.search(r"Query copy engine status (\d+), srcAgent (0x[0-9a-f]+), dstAgent (0x[0-9a-f]+), free_engine_mask (0x[0-9a-f]+), rec_engine_mask (0x[0-9a-f]+)", line)
        if q:
            cnt[phase][("query", "src=" + q.group(2)[-5:], "dst=" + q.group(3)[-5:], "free=" + q.group(4), "recommended=" + q.group(5))] += 1
for l in init[:4]:
    print(l)
for ph, c in cnt.items():
    if ph in ("start", "idle"):
        continue
    print(f"== {ph}")
    for k, v in sorted(c.items()):
        print(f"   {v:6d}  {k}")

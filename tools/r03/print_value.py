#!/usr/bin/env python3
"""print the `value` (and the per-kernel us per frame) of a bench.py JSON line:  print_value.py file.json [labels ...]"""
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks = (d.get("roofline") or {}).get("kernels") or {}
    print(*sys.argv[2:], d["value"], {k: v.get("us_per_frame") for k, v in ks.items()})
except Exception as e:  # noqa
    print(*sys.argv[2:], "ERR", e)

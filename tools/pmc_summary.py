#!/usr/bin/env python3
"""Per-kernel averages of the counters in one or more rocprofv3 --pmc output directories:
  python tools/pmc_summary.py DIR [DIR ...]  ->  JSON {kernel: {counter: average per dispatch, "dispatches": n}}"""
import csv
import glob
import json
import os
import sys

res = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                if k.startswith("void at::") or "elementwise" in k or "Memset" in k:
                    continue
                e = res.setdefault(k[:110], {})
                e.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {}
for k, e in res.items():
    out[k] = {c: sum(v) / len(v) for c, v in e.items()}
    out[k]["dispatches"] = max(len(v) for v in e.values())
print(json.dumps(out, indent=1, sort_keys=True))

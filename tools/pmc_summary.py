#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel name: pmc_summary.py <counter_collection.csv> [substring ...]"""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
acc = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if want and not any(w in name for w in want):
        continue
    acc[(name[:120], r["Counter_Name"])].append(float(r["Counter_Value"]))
out = [{"kernel": k, "counter": c, "launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
       for (k, c), v in acc.items()]
print(json.dumps(out, indent=1))

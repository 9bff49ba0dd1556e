#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", "")
    if filt and filt not in name:
        continue
    acc[name[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))

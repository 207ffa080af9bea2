#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name: prints mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        acc[name[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, ctrs in acc.items():
    print(name)
    for c, v in sorted(ctrs.items()):
        print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):.4g}")

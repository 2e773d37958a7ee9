#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (name prefix filter), mean of every counter per dispatch."""
import csv, glob, os, sys
d, pref = sys.argv[1], sys.argv[2:]
f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
acc = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    if pref and not any(k.startswith(p) for p in pref):
        continue
    a = acc.setdefault(k, {})
    c = a.setdefault(r["Counter_Name"], [0.0, 0])
    c[0] += float(r["Counter_Value"]); c[1] += 1
for k, a in acc.items():
    print(k)
    for c, (s, n) in sorted(a.items()):
        print(f"   {c:32s} {s / n:16.0f}  (x{n})")

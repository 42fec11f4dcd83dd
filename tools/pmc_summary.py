#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel (name, grid) mean of every counter."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:52], r["Grid_Size"])
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print("%s grid=%s" % key)
    for c, v in sorted(cs.items()):
        print("    %-32s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))

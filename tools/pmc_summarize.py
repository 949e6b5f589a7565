#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel, sum of each counter over dispatches."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
        for r in csv.DictReader(open(f)):
            import re
            m = re.search(r"(\w+_kernel\w*(<[^>]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])] += 1
        print(f)
        for k, cs in acc.items():
            print("  ", k)
            for c, v in sorted(cs.items()):
                print("      %-28s %16.0f  (%d dispatches)" % (c, v, calls[(k, c)]))

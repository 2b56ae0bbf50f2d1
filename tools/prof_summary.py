#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory into a small text summary (kept under profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-70s calls=%-5s avg_ns=%-12s total_ns=%-14s pct=%s" % (row.get("Name", "")[:70], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"), row.get("Percentage")))
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== PMC (mean per dispatch)")
for k, d in acc.items():
    print(" ", k)
    for c, v in sorted(d.items()):
        print("     %-28s %.4g   (n=%d)" % (c, sum(v) / len(v), len(v)))

#!/usr/bin/env python
"""Fold rocprofv3 --pmc counter_collection CSVs into per-kernel-symbol averages.

usage: pmc_summarize.py OUT.csv DIR [DIR ...]    (each DIR = one --pmc pass, any counters)
Output rows: kernel, counter, dispatches, mean, sum.  Template arguments are kept, parameter lists dropped."""
import csv, glob, os, re, sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: [0, 0.0])
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
                name = re.sub(r"^void ", "", name)
                name = re.sub(r"\(.*$", "", name).replace("gl::", "")
                a = acc[(name, r["Counter_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "dispatches", "mean", "sum"])
    for (k, c), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, f"{s / n:.3f}", f"{s:.1f}"])
print(f"{out}: {len(acc)} rows")

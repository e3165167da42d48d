#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

def main(root, pattern="k_sweep"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if pattern not in name:
                continue
            per_dispatch[(name, row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
        for (name, _), cs in per_dispatch.items():
            for c, v in cs.items():
                acc[name][c].append(v)
    for name, cs in acc.items():
        print(name)
        for c, vs in sorted(cs.items()):
            print(f"  {c:28s} mean/dispatch = {sum(vs)/len(vs):.6g}   (n={len(vs)})")

if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))

#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter CSVs: mean counter value per kernel name.
usage: pmc_summary.py <dir-with-*_counter_collection.csv> [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                short = name.split("(")[0].replace("void sclhip::", "").replace("sclhip::", "")
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"{'kernel':60s} {'counter':14s} {'calls':>6s} {'mean':>16s}")
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"{k[:60]:60s} {c:14s} {len(v):6d} {sum(v) / len(v):16.1f}")

#!/usr/bin/env python3
"""Per-kernel duration over the LAST k launches of a rocprofv3 --kernel-trace run (the timed region of bench.py: the
placement probe and the warm-up launch the same kernels earlier, on other arrangements of the operands).
usage: trace_tail_stats.py <*_kernel_trace.csv> <k> [name-filter ...]"""
import csv
import sys
from collections import defaultdict

path, k = sys.argv[1], int(sys.argv[2])
flt = sys.argv[3:]
runs = defaultdict(list)
for row in csv.DictReader(open(path)):
    name = row["Kernel_Name"].split("(")[0].replace("void sclhip::", "").replace("sclhip::", "")
    if flt and not any(f in name for f in flt):
        continue
    runs[name].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
print(f"{'kernel':50s} {'launches':>8s} {'all: mean ms':>13s} {'last %d: mean ms' % k:>17s} {'min':>8s} {'max':>8s}")
for name, v in sorted(runs.items()):
    v.sort()
    d = [x[1] / 1e6 for x in v]
    tail = d[-k:]
    print(f"{name[:50]:50s} {len(d):8d} {sum(d) / len(d):13.4f} {sum(tail) / len(tail):17.4f} {min(tail):8.4f} {max(tail):8.4f}")

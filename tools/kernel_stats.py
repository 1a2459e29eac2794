#!/usr/bin/env python3
"""Summarise per-kernel resources from the device assembly (make -C .../csrc asm -> capi.s):
VGPRs, SGPRs, LDS bytes, scratch bytes.  Usage: kernel_stats.py capi.s [filter]"""
import re
import subprocess
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
txt = open(path).read()
meta = txt[txt.index("amdhsa.kernels:"):]
rows = []
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
    name = g("name").group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"^void sclhip::", "", dem).split("(")[0]
    rows.append((dem, int(g("vgpr_count").group(1)), int(g("sgpr_count").group(1)),
                 int(g("group_segment_fixed_size").group(1)), int(g("private_segment_fixed_size").group(1))))
print(f"{'kernel':70s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>8s}")
for r in sorted(rows):
    if flt in r[0]:
        print(f"{r[0][:70]:70s} {r[1]:5d} {r[2]:5d} {r[3]:7d} {r[4]:8d}")
print("kernels with scratch:", sum(1 for r in rows if r[4]), "of", len(rows))

#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes over bench.py's headline configuration (FETCH_SIZE and WRITE_SIZE
cannot share a pass on gfx950): mean HBM bytes per launch of the share and the reconstruct kernel, FETCH_SIZE doubled as
MI355X_MICROARCH.md's HBM section prescribes (the counter reports half the bytes of 16-byte-per-lane streaming reads; checked
here against k_copy16, whose byte count is known).  The file is stamped with the kernel symbols and a hash of the kernel
sources; bench.py reports `traffic` only while that hash still matches.

usage (GPU box): make_pmc_traffic.py <fetch-pass dir> <write-pass dir> <copy bytes per k_copy16 launch> > profiles/pmc_traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (only its source-hash helper; importing it touches neither torch nor the GPU)


fetch, write = bench.pmc_means(sys.argv[1], "FETCH_SIZE"), bench.pmc_means(sys.argv[2], "WRITE_SIZE")
copy_bytes = float(sys.argv[3])
out = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py "
                   "--open 0 --cpu-sample 0 --steps 5` (headline = BASELINE configs[1]: n=10, t=3, Mersenne61, 1e8 secrets; "
                   "`configs` = the side configurations of the same run); counters are in KiB; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (HBM section), calibrated on k_copy16 below (16-byte-per-lane streaming loads, which is "
                   "what every kernel listed here issues)",
       "config": {"field": "m61", "n": 10, "t": 3, "secrets_per_gpu": 100000000, "share_mode": "coeffs"},
       "kernel_source_sha256_16": bench.kernel_source_hash(), "kernel_sources": list(bench.KERNEL_SOURCES)}
rep = bench.pmc_report(fetch, write, copy_bytes)   # the parsing bench.py itself uses for its live passes
assert rep.get("calibration_k_copy16") and rep["shamir_share"] and rep["shamir_recover"], "headline kernels / k_copy16 not in the passes"
out.update(rep)
json.dump(out, sys.stdout, indent=2)
print()

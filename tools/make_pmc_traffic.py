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


def means(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    # one kernel may run at several sizes in the bench run (the first-use self-check of the GF(2^128) reconstruct kernel is
    # a 4096-secret launch of the kernel C4 then runs at 1.25e7): the figure is the mean over the launches of the LARGEST
    # size, told apart by their counter value (within a factor of two of the maximum)
    out = {}
    for k, v in acc.items():
        big = [x for x in v if x >= 0.5 * max(v)]
        out[k] = (sum(big) / len(big), len(big))
    return out


def pick(table, needle):
    hits = [(k, v) for k, v in table.items() if needle in k]
    assert len(hits) == 1, (needle, [k for k, _ in hits])
    return hits[0]


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
copy_bytes = float(sys.argv[3])
out = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py "
                   "--open 0 --cpu-sample 0 --steps 5` (headline = BASELINE configs[1]: n=10, t=3, Mersenne61, 1e8 secrets; "
                   "`configs` = the side configurations of the same run); counters are in KiB; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (HBM section), calibrated on k_copy16 below (16-byte-per-lane streaming loads, which is "
                   "what every kernel listed here issues)",
       "config": {"field": "m61", "n": 10, "t": 3, "secrets_per_gpu": 100000000, "share_mode": "coeffs"},
       "kernel_source_sha256_16": bench.kernel_source_hash(), "kernel_sources": list(bench.KERNEL_SOURCES)}
ck, (cf, cn) = pick(fetch, "k_copy16")
_, (cw, _) = pick(write, "k_copy16")
out["calibration_k_copy16"] = {"bytes_read_per_launch": copy_bytes, "fetch_kib_reported": cf, "write_kib": cw,
                               "fetch_correction": copy_bytes / (cf * 1024.0), "launches": cn}
def entry(needle):
    name, (f_kib, nl) = pick(fetch, needle)
    _, (w_kib, _) = pick(write, needle)
    return {"kernel": name.split("(")[0].replace("void ", ""), "launches": nl, "fetch_kib_reported": f_kib, "write_kib": w_kib,
            "bytes": int(round(2 * f_kib * 1024 + w_kib * 1024))}


for key, needle in (("shamir_share", "k_share_small_t<sclhip::M61"), ("shamir_recover", "k_recover_fixed<sclhip::M61")):
    out[key] = entry(needle)
# the side configurations of the same bench run (`configs` in the result line), by the kernel each one launches; a
# configuration whose kernels are not in the passes (bench.py run with --configs 0) is left out
out["configs"] = {}
for cfg, (share_needle, rec_needle, algo) in {
    "C3_mersenne127_10_3": ("k_share_small_t<sclhip::M127", "k_recover_fixed<sclhip::M127", (224 * 10**7, 176 * 10**7)),
    "C3_mont128_10_3": ("k_share_small_t<sclhip::Mont128", "k_recover_table<sclhip::Mont128", (224 * 10**7, 176 * 10**7)),
    "C4_shard_gf2_128_40_13": ("k_share_gf_tiles<13>", "k_recover_gf128_pos<512", (864 * 125 * 10**5, 656 * 125 * 10**5)),
    "C5_shard_mersenne61_128_42": ("k_share_mfma_m61", "k_recover_table<sclhip::M61", (1368 * 125 * 10**6, 1032 * 125 * 10**6)),
}.items():
    try:
        sh, rc = entry(share_needle), entry(rec_needle)
    except AssertionError:
        continue
    sh["algorithmic_bytes"], rc["algorithmic_bytes"] = algo
    sh["traffic_over_algorithmic"], rc["traffic_over_algorithmic"] = sh["bytes"] / algo[0], rc["bytes"] / algo[1]
    out["configs"][cfg] = {"share": sh, "recover": rc}
json.dump(out, sys.stdout, indent=2)
print()

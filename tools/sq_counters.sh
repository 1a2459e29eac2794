#!/bin/bash
# GPU box: SQ counters of every bench kernel HBM does not bound (tools/run_sq_once.py): rocprofv3 --pmc only, the program itself
# after `--`, five passes (the counters do not fit one); the median over the launches at the largest size of every counter per kernel, the
# per-element instruction counts and issue intervals derived from them -> gpurun_out/r6_sq_counters.txt and sq_counters.json
# (copy both to profiles/: bench_legs/compute_roofline.py reads profiles/sq_counters.json).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
rm -f /tmp/sq_rows.txt
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_FLAT" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/sqp
  timeout -k 10 300 rocprofv3 --pmc $pass -d /tmp/sqp --output-format csv -- python3 tools/run_sq_once.py > /dev/null 2>/tmp/sqp.err || { echo "pass failed: $pass"; tail -3 /tmp/sqp.err; exit 2; }
  python3 - <<'PY'
import csv, glob
from collections import defaultdict
names = {"k_ew_inv<sclhip::M61, false": "m61_inv", "k_ew_inv_blocked<sclhip::M127": "m127_inv", "k_ew_inv_rolled<sclhip::Mont128": "mont128_inv",
         "k_ew_inv_rolled<sclhip::Gf128": "gf2_128_inv", "k_ew_gf128_mul": "gf2_128_mul", "k_share_gf_tiles<13>": "c4_share",
         "k_recover_gf128_pos<512": "c4_recover", "k_prg_blocks": "prg_blocks"}
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob("/tmp/sqp/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        for needle, tag in names.items():
            if needle in row["Kernel_Name"]:
                acc[tag][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("/tmp/sq_rows.txt", "a") as fh:
    for tag, cs in acc.items():
        for c, v in sorted(cs.items()):
            big = sorted(x for x in v if x >= 0.25 * max(v)) if max(v) > 0 else sorted(v)   # (the first-use self-check of the reconstruct kernel is a tiny launch)
            med = big[len(big) // 2] if len(big) % 2 else 0.5 * (big[len(big) // 2 - 1] + big[len(big) // 2])   # the median: one cold first launch does not move it
            fh.write(f"{tag}\t{c}\t{med}\t{len(big)}\n")
PY
done
python3 - <<'PY' > gpurun_out/r6_sq_counters.txt
import json
from collections import defaultdict
ELEMENTS = {"m61_inv": 1e8, "m127_inv": 1e7, "mont128_inv": 1e7, "gf2_128_inv": 1e7, "gf2_128_mul": 1e7, "c4_share": 1.25e7,
            "c4_recover": 1.25e7, "prg_blocks": 2e8}
rows = defaultdict(dict)
for ln in open("/tmp/sq_rows.txt"):
    tag, c, v, n = ln.rstrip("\n").split("\t")
    rows[tag][c] = float(v)
print("# tools/sq_counters.sh: rocprofv3 --pmc passes over tools/run_sq_once.py; median over the launches (SQ_* summed over the chip, GRBM_GUI_ACTIVE over the 8 XCDs)")
for tag, r in rows.items():
    for c, v in sorted(r.items()):
        print(f"{tag:14s} {c:24s} {v:18.1f}")
print("\n# derived: cycles = GRBM_GUI_ACTIVE / 8; per SIMD = / 1024; per element and lane = wave instructions x 64 / elements")
out = {}
for tag, r in rows.items():
    if "SQ_INSTS_VALU" not in r or "GRBM_GUI_ACTIVE" not in r:
        continue
    el = ELEMENTS[tag]
    cyc = r["GRBM_GUI_ACTIVE"] / 8
    valu, lds = r["SQ_INSTS_VALU"], r.get("SQ_INSTS_LDS", 0.0)
    waves = max(1.0, r.get("SQ_WAVES", 1.0))
    d = {"elements": el, "cycles": cyc, "valu_per_element": valu * 64 / el, "lds_per_element": lds * 64 / el,
         "valu_issue_cycles_measured": cyc / (valu / 1024), "lds_cycles_per_access_per_cu": (cyc / (lds / 256)) if lds else None,
         "waves": waves, "wait_share_of_wave_cycles": r.get("SQ_WAIT_INST_ANY", 0) / max(1.0, r.get("SQ_WAVE_CYCLES", 1)),
         "waves_per_simd_in_all": waves / 1024,
         "lds_bank_conflict_cycles": r.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active_cycles": r.get("SQ_LDS_IDX_ACTIVE")}
    if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
        # HBM bytes per element: counters in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads
        # (what these kernels issue, scratch round trips of the rolled chains included: they move 16-byte elements)
        d["hbm_bytes_per_element"] = (2 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024 / el
        d["hbm_fetch_kib_reported"], d["hbm_write_kib"] = r["FETCH_SIZE"], r["WRITE_SIZE"]
        if tag.endswith("_inv"):
            E = 8 if tag == "m61_inv" else 16
            d["traffic_over_algorithmic"] = d["hbm_bytes_per_element"] / (2 * E)
    out[tag] = d
    print(f"{tag:14s} {cyc:10.0f} cycles  {d['valu_per_element']:8.1f} vector / {d['lds_per_element']:7.1f} LDS instructions per element and lane  "
          f"one vector instruction per SIMD every {d['valu_issue_cycles_measured']:5.2f} cycles"
          + (f"  one LDS instruction per CU every {d['lds_cycles_per_access_per_cu']:6.2f} cycles" if lds else "")
          + f"  waiting share {d['wait_share_of_wave_cycles']:4.2f}"
          + (f"  HBM {d['hbm_bytes_per_element']:7.1f} B per element" if "hbm_bytes_per_element" in d else "")
          + (f" = {d['traffic_over_algorithmic']:4.2f} x algorithmic" if "traffic_over_algorithmic" in d else ""))
json.dump({"_comment": "tools/sq_counters.sh on an MI355X; read by bench_legs/compute_roofline.py", "kernels": out},
          open("gpurun_out/sq_counters.json", "w"), indent=1)
PY
cat gpurun_out/r6_sq_counters.txt | tail -12

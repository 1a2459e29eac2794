#!/bin/bash
# GPU box: SQ counters of the element-wise kernels that are not stream-bound (tools/run_ew_once.py): rocprofv3 --pmc only, the
# program itself after `--`; mean per launch of every counter per kernel, then the derived busy fractions (profiles/r5_ew_sq.txt).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
rm -f /tmp/ew_sq_rows.txt
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_FLAT"; do
  rm -rf /tmp/sqp
  timeout -k 10 200 rocprofv3 --pmc $pass -d /tmp/sqp --output-format csv -- python3 tools/run_ew_once.py > /dev/null 2>/tmp/sqp.err || { echo "pass failed: $pass"; tail -3 /tmp/sqp.err; continue; }
  python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
names = {"k_ew_inv<sclhip::M61, false": "M61 inv", "k_ew_inv<sclhip::M61, true": "M61 div", "k_ew_inv_rolled<sclhip::M127": "M127 inv/div",
         "k_ew_inv_rolled<sclhip::Gf128": "GF inv/div", "k_ew_gf128_mul": "GF mul"}
for path in glob.glob("/tmp/sqp/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        for needle, tag in names.items():
            if needle in row["Kernel_Name"]:
                acc[tag][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("/tmp/ew_sq_rows.txt", "a") as fh:
    for tag, cs in acc.items():
        for c, v in sorted(cs.items()):
            print(f"{tag:14s} {c:24s} {len(v):3d} {sum(v)/len(v):18.1f}")
            fh.write(f"{tag}\t{c}\t{sum(v)/len(v)}\n")
PY
done
python3 - <<'PY'
from collections import defaultdict
rows = defaultdict(dict)
for ln in open("/tmp/ew_sq_rows.txt"):
    tag, c, v = ln.rstrip("\n").split("\t")
    rows[tag][c] = float(v)
print("\n# derived (SQ_ACTIVE_INST_* count instructions on this chip, like SQ_INSTS_*; GRBM_GUI_ACTIVE is summed over the 8 XCDs):")
print("# cycles = GUI / 8; per SIMD: instructions / 1024; an LDS instruction here is a 64-lane ds_read_b128 (4 cycles of the pipe) or ds_write_b128 (11-14, profiles/r4_ldsbank.txt) = 1 KiB")
for tag, r in rows.items():
    if "SQ_INSTS_VALU" in r and "GRBM_GUI_ACTIVE" in r:
        cyc = r["GRBM_GUI_ACTIVE"] / 8
        valu, lds = r["SQ_INSTS_VALU"] / 1024, r.get("SQ_INSTS_LDS", 0) / 1024
        print(f"{tag:14s} {cyc:9.0f} cycles  {valu:8.0f} VALU instructions per SIMD = one every {cyc / valu:4.2f} cycles"
              + (f"  {lds * 4:8.0f} LDS instructions per CU = {lds * 4 * 1024 / cyc:5.0f} B per cycle and CU" if lds else "")
              + f"  {r.get('SQ_WAVES', 0):8.0f} waves, {r['SQ_INSTS_VALU'] / max(1.0, r.get('SQ_WAVES', 1)):7.0f} VALU instructions each"
              + f"  waiting share of wave cycles {r.get('SQ_WAIT_INST_ANY', 0) / max(1.0, r.get('SQ_WAVE_CYCLES', 1)):4.2f}")
PY
